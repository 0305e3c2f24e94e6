#!/usr/bin/env python
"""Headline benchmark: DPO preference-pair steps/sec, CogVideoX-5B 49f@480x720 (BASELINE.json configs[1]).

One step = one preference pair per rank through the whole hot path: fused noising + v-targets, frozen-reference
forward (win+lose), policy forward (win+lose) with LoRA r=64, Diffusion-DPO loss, backward to the 66 M LoRA
parameters, flat-buffer gradient all-reduce (N>1), fused clip + AdamW.  Synthetic latents / prompt embeddings of the
named shape, random-init weights of the named architecture (no network for checkpoints), all resident in HBM before
the timed region.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5]

cfg5 = BASELINE.json configs[4] (Wan2.2-TI2V-5B, 81f x 704x1280, fp8 feed-forward) is a separate bench line with its own metric name; the
default (and what the driver runs) is cfg2, the configuration the headline metric is quoted on.

`--gpus N` with N > 1 launches N ranks itself (one process per GPU under torch.distributed.run, RCCL); it can also be
started under torchrun directly, in which case WORLD_SIZE must equal --gpus:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
PEAK_FP8_DENSE_TFLOPS = 5000.0    # same guide: ~5 PF dense fp8 (the vendor e4m3 GEMMs of cfg5 are priced against this)
PEAK_HBM_GBS = 8000.0             # same guide: HBM3E ~8 TB/s
TEXT_LEN = 226
# live per-kernel HIP events (ops.KernelTimer): every launch during the first TIMER_FULL_STEPS timed steps, afterwards only the attention kernels
# the roofline objects are computed from -- event packets between ALL ~1500 launches of a step cost 1.3 % of it
TIMER_FULL_STEPS = 3
TIMER_ALWAYS = ("attn_fwd_kernel", "attn_bwd_dkv_kernel", "attn_bwd_dq_kernel", "attn128_fwd", "attn128_fwd_f8", "attn128_bwd")

# BASELINE.json configs that fit a bench line (SURVEY section 8 sizes).  cfg2 is the headline the metric is quoted on.
CONFIGS = {
    "cfg2": dict(model="COGVIDEOX_5B", frames=13, height=60, width=90, checkpoint=False, cond=False, metric="CogVideoX-5B 49f@480x720",
                 label="BASELINE configs[1]: CogVideoX-5B T2V full (42 blocks, D=3072, 48x64 heads), 49f x 480x720"),
    "cfg3": dict(model="COGVIDEOX_5B_I2V", frames=13, height=60, width=90, checkpoint=False, cond=True, metric="CogVideoX-5B-I2V 49f@480x720 + image-cond latent",
                 label="BASELINE configs[2]: CogVideoX-5B-I2V (32 input channels, learned positional table), 49f x 480x720 + image-cond latent"),
    # S = 41 026: 6.6 GB of saved activations per block and pair would need 277 GB + weights; with lean activations (5.1 GB per block) all 42
    # blocks stay resident in 230 GB and nothing is recomputed (9.17 s; every 4th block recomputed without lean: 9.70 s in 227 GB)
    "cfg4": dict(model="COGVIDEOX_1_5_5B", frames=21, height=96, width=170, checkpoint=False, cond=False, lean=True, metric="CogVideoX1.5-5B 81f@768x1360",
                 label="BASELINE configs[3]: CogVideoX1.5-5B T2V (patch_size_t=2), 81f x 768x1360 (21 latent frames, even-cropped to 20)"),
    "cfg5": dict(model="WAN22_TI2V_5B", frames=21, height=44, width=80, checkpoint=False, cond=True,
                 label="BASELINE configs[4]: Wan2.2-TI2V-5B (30 blocks, dim 3072, 24x128 heads, ffn 14336, text 512), 81f x 704x1280 -> latent 48x21x44x80, "
                       "fp8 MFMA path = e4m3 self-attention forward (hand-written kernel) + e4m3 feed-forward GEMMs; bf16 attention backward and LoRA-carrying projections"),
}
WAN_TEXT_LEN = 512


def build_model(cfg_kw, device, seed):
    from videogpa_amd.transformer import CogVideoXTransformer3DModel
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(device):
            model = CogVideoXTransformer3DModel(**cfg_kw)
    finally:
        torch.set_default_dtype(prev)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("norm.weight") or name in ("norm_final.weight",) or ".norm_q.weight" in name or ".norm_k.weight" in name:
                p.fill_(1.0)
            elif name.endswith(".bias"):
                p.zero_()
            else:
                p.normal_(0.0, 0.02, generator=g)
    return model


_RSMI = None


def read_joules(device=0):
    """Accumulated socket energy of GPU `device` in joules (librocm_smi64 rsmi_dev_energy_count_get; None when the library or the counter is not
    there).  Read around the timed region: J per step and mean W next to ms per step -- every MFMA-bound launch of this step runs at the part's
    1.4 kW cap, so time IS joules."""
    global _RSMI
    import ctypes
    if _RSMI is None:
        _RSMI = False
        for name in ("librocm_smi64.so", "/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so.1"):
            try:
                lib = ctypes.CDLL(name)
                if lib.rsmi_init(ctypes.c_uint64(0)) == 0:
                    _RSMI = lib
                    break
            except OSError:
                continue
    if not _RSMI:
        return None
    cnt, res, ts = ctypes.c_uint64(0), ctypes.c_float(0.0), ctypes.c_uint64(0)
    if _RSMI.rsmi_dev_energy_count_get(ctypes.c_uint32(device), ctypes.byref(cnt), ctypes.byref(res), ctypes.byref(ts)) != 0:
        return None
    return cnt.value * float(res.value) * 1e-6


def energy_report(e0, e1, dt, steps, flops_algorithmic, flops_executed=None):
    if e0 is None or e1 is None or e1 <= e0:
        return None
    J = (e1 - e0) / steps
    rep = {"joules_per_step": J, "mean_power_w": (e1 - e0) / dt, "algorithmic_tflop_per_joule": flops_algorithmic / J / 1e12,
           "source": "rsmi_dev_energy_count_get around the timed region (rank 0's GPU)"}
    if flops_executed:
        rep["executed_tflop_per_joule"] = flops_executed / J / 1e12
        rep["note"] = ("executed = algorithmic + the S / dP recomputes of the split attention backward (6 S^2 d per head and layer); hipBLASLt's GEMM and the "
                       "attention backward kernels measure 1.02-1.04 executed TFLOP/J stand-alone (profiles/r04_energy_per_launch.txt)")
    return rep


def flops_per_pair_step(S, D, L, r):
    """BASELINE.md section 4 (algorithmic, no recompute counted)."""
    f_lin, f_attn, f_lora = 24.0 * S * D * D, 4.0 * S * S * D, 16.0 * S * D * r
    fwd = L * (f_lin + f_attn)
    return 2 * fwd + 2 * (fwd + L * f_lora) + 2 * L * (f_lin + 2 * f_attn + 2 * f_lora)


def flops_per_pair_step_wan(L_tok, D, F, T, layers, r):
    """The same accounting for the Wan2.2 block: per token 6 D^2 weights in the attentions (self q,k,v,o + cross q,o) and 2 D F in the feed-forward,
    cross k,v on the T text tokens, self-attention 4 L^2 D and cross-attention 4 L T D per forward; LoRA r on the eight attention linears.  Frozen
    linears cost one forward-equivalent in the backward (dX only), attention two."""
    f_lin = 2.0 * L_tok * (6 * D * D + 2 * D * F) + 2.0 * T * 2 * D * D
    f_attn = 4.0 * L_tok * L_tok * D + 4.0 * L_tok * T * D
    f_lora = 2.0 * (6 * L_tok + 2 * T) * 2 * D * r
    fwd = layers * (f_lin + f_attn)
    return 2 * fwd + 2 * (fwd + layers * f_lora) + 2 * layers * (f_lin + 2 * f_attn + 2 * f_lora)


def flops_fp8_wan(L_tok, D, F, layers, ffn_fp8=True, attn_fp8=True):
    """the part of flops_per_pair_step_wan's count that runs on e4m3 operands (BASELINE configs[4] "fp8 MFMA path"): the two feed-forward GEMMs in all four
    forwards and as dX in both backwards, and both products of the self-attention FORWARD (4 L^2 D) in all four forwards; everything else is bf16"""
    f_ffn = 2.0 * L_tok * 2 * D * F
    f_sa = 4.0 * L_tok * L_tok * D
    return layers * ((6.0 * f_ffn if ffn_fp8 else 0.0) + (4.0 * f_sa if attn_fp8 else 0.0))


def cpu_baseline(F_step, budget_s=200.0):
    """The oracle (kind "port": diffusers / peft are not installed, so the reference's own step cannot run) timed on the
    host cores on BASELINE configs[0] -- the reference's CPU-runnable case: CogVideoX-5B width (D=3072, 48 heads), 2
    transformer blocks, 13f x 64 x 64 paired latents (S = 13 538 tokens), LoRA r=8, fp32: the FULL pair-step
    (4 forwards + backward to the LoRA parameters + clip + AdamW), 1 untimed warm-up + 3 timed steps (median; fewer when
    a step takes so long that three would not fit `budget_s`).  `value` is that time scaled to the bench line's own
    configuration by the algorithmic-FLOP ratio (flagged extrapolated, SURVEY 8d / BASELINE.md section 3)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cfg1_common as c1
    from oracle import cogvideox as ocv
    from oracle import scheduler as osch
    threads, probe = _pick_cpu_threads()
    torch.set_num_threads(threads)
    cfg = c1.config()
    sd = {k: v.float() for k, v in c1.base_state_dict(cfg).items()}
    lora0, r = c1.lora_state_dict(cfg, "r8")
    x_win, x_lose, prompt, t, noise = (v.float() if v.is_floating_point() else v for v in c1.inputs())
    abar = osch.alphas_cumprod()

    def step():
        lora = {k: v.clone().requires_grad_(True) for k, v in lora0.items()}
        opt = torch.optim.AdamW(list(lora.values()), lr=5e-6, weight_decay=0.01)
        t0 = time.perf_counter()
        out = ocv.dpo_pair_step(sd, cfg, lora, abar, x_win, x_lose, prompt, t, noise, beta=1.0)
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(list(lora.values()), 1.0)
        opt.step()
        return time.perf_counter() - t0, float(out["loss"].detach())

    # warm-up = the same code path on a 1-frame version of the inputs (thread pools, allocator, oneDNN primitive caches)
    _x = (x_win[:, :, :1], x_lose[:, :, :1], noise[:, :1])
    ocv.dpo_pair_step(sd, cfg, {k: v.clone().requires_grad_(True) for k, v in lora0.items()}, abar, _x[0], _x[1], prompt, t, _x[2])["loss"].backward()
    times, loss = [], None
    while len(times) < 3 and (not times or sum(times) + times[-1] <= budget_s):      # 3 steps when one takes <= ~65 s, fewer on a slow host
        dt, loss = step()
        times.append(dt)
    med = statistics.median(times)
    S1 = c1.TEXT_LEN + c1.FRAMES * (c1.HEIGHT // 2) * (c1.WIDTH // 2)
    F1 = flops_per_pair_step(S1, cfg.inner_dim, cfg.num_layers, r)
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        model = "unknown"
    ratio = F_step / F1
    return {"value": 1.0 / (med * ratio), "unit": "pair-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"MEASURED: oracle fp32 full pair-step (4 fwd + bwd) of BASELINE configs[0] (2 blocks, D=3072, S={S1}, r={r}): "
                      f"{len(times)} timed step(s) after a 1-frame warm-up, median {med:.2f} s = {1.0 / med:.4f} pair-steps/s (loss {loss:.6f}); "
                      f"{F1 / med / 1e9:.0f} GFLOP/s on {model} with {threads} of {os.cpu_count()} hardware threads (the fastest of {probe}). "
                      f"`value` is that time scaled to this bench line's configuration by the algorithmic-FLOP ratio {ratio:.1f} "
                      "(EXTRAPOLATED, SURVEY 8d: the full configuration would take hours on the host)",
            "measured": {"config": "BASELINE configs[0]", "value": 1.0 / med, "unit": "pair-steps/s", "step_seconds": times,
                         "gflops": F1 / med / 1e9},
            "flop_ratio_to_this_config": ratio, "cpu_model": model}


def _pick_cpu_threads():
    """torch's CPU kernels do not scale to every hardware thread of a big host (256 threads ran the oracle 7x slower than 16
    on the MI355X box): time a small matmul + attention probe at a few thread counts and keep the fastest."""
    import torch.nn.functional as F
    try:
        cores = len(os.sched_getaffinity(0))          # the CPU leg may have been fenced off from the cores of the secondary-config children
    except AttributeError:
        cores = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)
    x, W = torch.randn(2048, 3072, generator=g), torch.randn(3072, 3072, generator=g)
    q = torch.randn(1, 16, 2048, 64, generator=g)
    cands = sorted({c for c in (cores, cores // 2, cores // 4, 64, 32, 16, 8) if 1 <= c <= cores}, reverse=True)
    best, seen = None, {}
    for n in cands:
        torch.set_num_threads(n)
        F.linear(x, W); F.scaled_dot_product_attention(q, q, q)
        t0 = time.perf_counter()
        for _ in range(2):
            F.linear(x, W); F.scaled_dot_product_attention(q, q, q)
        seen[n] = round(time.perf_counter() - t0, 4)
        if best is None or seen[n] < seen[best]:
            best = n
    return best, seen


def add_kernel_report(out, ops, steps, ms, use_pmc=True):
    """per-kernel live timings (HIP events on the launch stream, ops.KernelTimer) -> out["kernels"], out["roofline"], out["roofline_worst"]"""
    if ops.TIMER is None:
        return
    if True:
        summ = ops.TIMER.summary()
        ops.TIMER = None
        kernels = {}
        for name, s in summ.items():
            rate = s["work_per_launch"] / (s["avg_ms"] * 1e-3)
            n_st = s.get("steps") or steps            # kernels outside the roofline set are sampled on the first timed steps only (ops.KernelTimer)
            k = {"launches_per_step": s["launches"] / n_st, "avg_ms": s["avg_ms"], "total_ms_per_step": s["total_ms"] / n_st, "timed_steps": n_st}
            if s["unit"] == "flop":
                peak = PEAK_FP8_DENSE_TFLOPS if ("fp8" in name or name.endswith("_f8")) else PEAK_BF16_DENSE_TFLOPS      # e4m3 kernels are priced against the fp8 peak
                k.update(bound="mfma", algorithmic_flops_per_launch=s["work_per_launch"], achieved_tflops=rate / 1e12, frac=rate / 1e12 / peak, peak_tflops=peak)
            else:
                k.update(bound="hbm", algorithmic_bytes_per_launch=s["work_per_launch"], achieved_gbs=rate / 1e9,
                         frac=rate / 1e9 / PEAK_HBM_GBS)
            kernels[name] = k
        # dominant kernel = largest share of the step among the HAND-WRITTEN kernels.  The dense projections are the vendor's
        # hipBLASLt (north_star: MFMA by hand only for attention and LoRA); they are timed too and reported as one entry
        # ("hipblaslt_gemm (vendor)": all shapes together) so that their share of the step is in the same JSON.
        own = [k for k in kernels if "(vendor)" not in k]
        dom = max(own, key=lambda k: kernels[k]["total_ms_per_step"])
        kd = kernels[dom]
        # traffic: HBM bytes per launch from the PMC pass of the same command (profiles/, collected per the guide's
        # recipe: separate --pmc runs, FETCH_SIZE/WRITE_SIZE with the gfx950 corrections), when a summary is present
        pmc_all = {}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if use_pmc and os.path.exists(pmc):        # the PMC pass ran at the cfg2 / cfg5 kernel shapes: other shapes report no traffic / occupancy
            with open(pmc) as f:
                pmc_all = json.load(f)
        traffic = pmc_all.get(dom, {}).get("hbm_bytes_per_launch")
        # which profile round the constants read from that file (traffic, mfma_busy, clock_mhz) come from: they are NOT measured in this run
        pmc_source = (pmc_all.get("__source__") or {}).get("profile_round") if pmc_all else None

        def roof(name):
            k = kernels[name]
            r = {"kernel": name, "bound": k["bound"], "achieved": k.get("achieved_tflops", k.get("achieved_gbs")),
                 "peak": k.get("peak_tflops", PEAK_BF16_DENSE_TFLOPS) if k["bound"] == "mfma" else PEAK_HBM_GBS, "unit": "TFLOP/s" if k["bound"] == "mfma" else "GB/s",
                 "frac": k["frac"], "avg_launch_ms": k["avg_ms"], "share_of_step": k["total_ms_per_step"] / ms,
                 "traffic": pmc_all.get(name, {}).get("hbm_bytes_per_launch")}
            for key in ("mfma_busy", "clock_mhz"):        # from the round's PMC pass (tools/profile_round.sh), when present
                if key in pmc_all.get(name, {}):
                    r[key] = pmc_all[name][key]
            if pmc_all.get(name):
                r["pmc_source"] = pmc_source
            return r
        if kd["bound"] == "mfma":
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": kd["achieved_tflops"], "peak": kd.get("peak_tflops", PEAK_BF16_DENSE_TFLOPS),
                               "unit": "TFLOP/s", "frac": kd["frac"], "traffic": traffic, "avg_launch_ms": kd["avg_ms"],
                               "note": "largest hand-written kernel of the step; the vendor GEMMs are listed under kernels"}
            for key in ("mfma_busy", "clock_mhz"):
                if key in pmc_all.get(dom, {}):
                    out["roofline"][key] = pmc_all[dom][key]
            if pmc_all.get(dom):
                out["roofline"]["pmc_source"] = pmc_source
        else:
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": kd["achieved_gbs"], "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": kd["frac"], "traffic": traffic, "avg_launch_ms": kd["avg_ms"]}
            if pmc_all.get(dom):
                out["roofline"]["pmc_source"] = pmc_source
        # the weakest hand-written kernel that matters (>= 5 % of the step), so that `roofline` cannot hide it
        big = [k for k in own if kernels[k]["total_ms_per_step"] >= 0.05 * ms]
        if big:
            out["roofline_worst"] = roof(min(big, key=lambda k: kernels[k]["frac"]))
        if "attn_bwd_dkv_kernel" in kernels and "attn_bwd_dq_kernel" in kernels:
            a, b2 = kernels["attn_bwd_dkv_kernel"], kernels["attn_bwd_dq_kernel"]
            fl = a["algorithmic_flops_per_launch"] + b2["algorithmic_flops_per_launch"]       # 8 S^2 d B H: the whole attention backward
            t_ms = a["avg_ms"] + b2["avg_ms"]
            out["attention_bwd_pair"] = {"algorithmic_flops_per_launch": fl, "avg_ms": t_ms, "achieved_tflops": fl / t_ms / 1e9,
                                         "frac": fl / t_ms / 1e9 / PEAK_BF16_DENSE_TFLOPS,
                                         "note": "dK/dV + dQ launches together against the algorithmic 8 S^2 d FLOPs (their S / dP recomputes are overhead)"}
        out["kernels"] = kernels


def dist_report(engine, dt_local, steps, dev, world, force_dist):
    """What a scaling loss would have to be attributed from (every rank calls this; rank 0 prints it): the ranks really in the group, each rank's own time
    per step (min / max: a straggler shows here, not in the max-over-ranks headline), and the gradient exchange -- message size, which collective, its mean
    duration on the communication stream and how long the compute stream actually stalled for it at the optimizer step (events, FlatAdamW.comm_report).
    The reference's only parallelism is DDP (train/CogVideoX-5B/03_train.py:257-266), whose all-reduce this one message replaces."""
    if not (world > 1 or force_dist):
        return {}
    t = torch.tensor([dt_local / steps * 1e3], dtype=torch.float64, device=dev)
    per = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(per, t)
    per = [float(x.item()) for x in per]
    comm = engine.opt.comm_report()
    if comm is not None:
        # rank 0's view plus the slowest rank's exposed wait (the one that bounds the step)
        w = torch.tensor([comm["exposed_wait_ms"], comm["allreduce_ms"]], dtype=torch.float64, device=dev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        comm["exposed_wait_ms_max_over_ranks"], comm["allreduce_ms_max_over_ranks"] = float(w[0].item()), float(w[1].item())
        comm["note"] = ("one exchange per optimizer step, issued on a side stream after the last backward and waited for between the NEXT step's reference "
                        "and policy passes (trainer.DPOEngine); allreduce_ms is measured from `gradients ready` to `collective done`")
    return {"ranks_seen": dist.get_world_size(), "ms_per_step_min": min(per), "ms_per_step_max": max(per), "ms_per_step_by_rank": per, "comm": comm}


def preflight(world, rank, local_rank, dev):
    """N > 1 runs must be attributable from their own output: every rank reports its device, free memory and the collective library it loaded; rank 0 prints the
    table to stderr before the first step and attaches it to the JSON line.  (NCCL_DEBUG=VERSION, set before the process group comes up, makes RCCL print its own
    version line to stderr as well.)"""
    free, total = torch.cuda.mem_get_info(dev)
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:      # noqa: BLE001
        ver = f"unavailable ({e!r})"
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "local_rank": local_rank, "device": props.name, "gcn_arch": getattr(props, "gcnArchName", None), "free_gb": free / 2 ** 30, "total_gb": total / 2 ** 30,
            "rccl_version": ver, "torch": torch.__version__, "hip": torch.version.hip, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
            "VGPA_DP_COLLECTIVE": os.environ.get("VGPA_DP_COLLECTIVE", "all_reduce"), "pid": os.getpid(), "host": os.uname().nodename}
    rows = [None] * dist.get_world_size()
    dist.all_gather_object(rows, mine)
    if rank == 0:
        for r in rows:
            print(f"bench.py preflight: rank {r['rank']} -> GPU {r['local_rank']} {r['device']} ({r['gcn_arch']}), {r['free_gb']:.0f} of {r['total_gb']:.0f} GiB free, RCCL {r['rccl_version']}, "
                  f"torch {r['torch']} / HIP {r['hip']}, HSA_ENABLE_IPC_MODE_LEGACY={r['HSA_ENABLE_IPC_MODE_LEGACY']}", file=sys.stderr, flush=True)
    return rows


def collective_ab(engine, batch, steps, world, force_dist, dev, barrier):
    """the OTHER gradient-exchange form over a second short timed region (same model, same batch), so that ONE `bench.py --gpus N` invocation yields the
    all_reduce / rs_ag A/B (a ring all-reduce is bound by one xGMI link, reduce-scatter + all-gather use all seven).  Runs after the headline's region
    and never touches its numbers.  -> {collective: {"ms_per_step", "comm"}} for the second form (every rank must call this)."""
    opt = engine.opt
    first = opt.collective
    other = "rs_ag" if first == "all_reduce" else "all_reduce"
    opt.collective = other
    try:
        engine.micro_step(batch)               # one untimed step on the new form (buffers, communicator channels)
        engine.flush()
        opt.comm_report()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            engine.micro_step(batch)
        engine.flush()
        barrier()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        comm = opt.comm_report()
        if comm is not None:
            w = torch.tensor([comm["exposed_wait_ms"], comm["allreduce_ms"]], dtype=torch.float64, device=dev)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            comm["exposed_wait_ms_max_over_ranks"], comm["allreduce_ms_max_over_ranks"] = float(w[0].item()), float(w[1].item())
        return {other: {"steps": steps, "ms_per_step": float(tt.item()) / steps * 1e3, "comm": comm}}
    finally:
        opt.collective = first


def attention_forward_summary(base, args):
    """which forward each attention layer ended up on (ops.AttnFwdPolicy: the bound-shifted w1 kernel, or the online-softmax entry once more than 5 % of a layer's
    strips had to be redone) and the largest redone fraction the kernels reported -- so that a run on a trained checkpoint shows whether its time is the fast path's"""
    rep = base.attention_forward_report()
    fr = [r["redo_fraction"] for r in rep if r["redo_fraction"] is not None]
    return {"weights": args.weights + (f" (QK-norm gain {args.qk_gain})" if args.weights == "trained_like" else ""), "layers": len(rep),
            "layers_on_online_softmax": sum(r["mode"] == "online" for r in rep), "max_redo_fraction": max(fr) if fr else None,
            "mean_redo_fraction": sum(fr) / len(fr) if fr else None}


def tuned_gemms_report(ops):
    """which hipBLASLt solutions the vendor GEMMs of this run used: the library's default heuristic, or the per-shape winners of tools/gemm_tune.py
    (videogpa_amd/tuned/, read by PyTorch TunableOp with tuning off; ignored by torch when the file was made on another torch / ROCm / hipBLASLt stack)"""
    st = ops.tuned_gemms_state() or {"enabled": False}
    rep = {"enabled": bool(st.get("enabled")), "entries": st.get("entries")}
    if st.get("file"):
        rep["file"] = os.path.relpath(st["file"], ROOT)
    if st.get("note"):
        rep["note"] = st["note"]
    return rep


def scorer_inputs(dev, pointmap=True):
    """the reference's scorer workload (train/01_preference_pair.py:33-34: NUM_FRAMES = 10 frames of 518 x 518 -> a cloud of 10 x 518 x 518 points re-projected into
    all 10 views).  pointmap=True: the cloud IS a per-pixel point map, as VGGT / DA3 emit it (pipelines/process_video.py:66-98) -- every frame's pixels unprojected
    at a smooth depth and moved into the world frame, so that neighbouring points land on neighbouring pixels; False: an unstructured Gaussian cloud (every
    atomic a different cache line: the worst case)."""
    T, H, W = 10, 518, 518
    N = T * H * W
    g = torch.Generator(device=dev).manual_seed(0)
    K = torch.tensor([[400.0, 0, W / 2], [0, 400.0, H / 2], [0, 0, 1]], device=dev).repeat(T, 1, 1)
    E = torch.eye(4, device=dev).repeat(T, 1, 1)
    for t in range(T):
        E[t, 0, 3] = 0.05 * t
    if pointmap:
        v, u = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
        pts = []
        for t in range(T):
            z = 3.0 + 0.5 * torch.sin(u / 60.0 + 0.3 * t) * torch.cos(v / 45.0) + 0.01 * torch.randn(H, W, generator=g, device=dev)
            cam = torch.stack([(u - W / 2) / 400.0 * z, (v - H / 2) / 400.0 * z, z], dim=-1)        # pixel -> camera frame of view t
            pts.append(cam - E[t, :3, 3])                                                              # -> world (R = I)
        pc = torch.stack(pts).reshape(N, 3).contiguous()
    else:
        pc = torch.randn(N, 3, generator=g, device=dev) * torch.tensor([1.5, 1.5, 0.5], device=dev) + torch.tensor([0, 0, 3.0], device=dev)
    colors = torch.rand(N, 3, generator=g, device=dev) * 255
    gt = (torch.rand(T, H, W, 3, generator=g, device=dev) * 255).to(torch.uint8)
    return T, H, W, N, pc, colors, K, E, gt


def scorer_report(dev, with_cpu=True):
    """the scorer kernels K14 / K15 "reported as GB/s vs the 8 TB/s peak" (utils/projection_utils.py:12-101, metrics/mse.py:14-54,
    metrics/consistency_score.py:8-40) at the reference's scale, next to the reference's own formulation (z-descending argsort + scatter per view, as
    project_points writes it) run through torch on this GPU, and -- in the CPU leg -- the oracle on the host cores.
    Algorithmic bytes per video: every point-view reads xyz + rgb (24 B) and does one 8-byte atomicMin; the resolve pass reads the z-buffer (8 B per pixel and
    view), gathers the winner's colour (12 B) and writes the fp32 frame (12 B)."""
    from videogpa_amd import scorer
    out = {}
    for kind, pm in (("pointmap", True), ("random_cloud", False)):
        T, H, W, N, pc, colors, K, E, gt = scorer_inputs(dev, pm)

        def timeit(f, n=20):
            f()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                r = f()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n, r
        ms, rep = timeit(lambda: scorer.batch_reproject(pc, colors, K, E, H, W))
        alg = T * N * (24 + 8) + T * H * W * (8 + 12 + 12)
        covered = float((rep.reshape(T, 3, -1) > -1.0).any(dim=1).float().mean())
        rec = {"ms_per_video": ms, "gpoint_views_per_s": T * N / ms / 1e6, "algorithmic_bytes": alg, "algorithmic_gbs": alg / ms / 1e6,
               "frac_of_hbm_peak": alg / ms / 1e6 / PEAK_HBM_GBS, "atomics_per_s": T * N / (ms * 1e-3), "pixels_covered": covered}
        if pm:
            m = scorer.MSEMetric()
            ms2, _ = timeit(lambda: m.compute_device(gt=gt, rep=rep))
            mse_bytes = 2 * (T * H * W * 3) * (1 + 4)               # range scan + squared-error pass over the u8 frames and the fp32 re-projections
            rec["mse_ms"] = ms2
            rec["mse_gbs"] = mse_bytes / ms2 / 1e6
            rec["mse_frac_of_hbm_peak"] = mse_bytes / ms2 / 1e6 / PEAK_HBM_GBS
            ms3, _ = timeit(lambda: scorer.compute_motion_score_vectorized(E))
            rec["motion_score_us"] = ms3 * 1e3

            def ref_style():                                         # project_points as the reference writes it (utils/projection_utils.py:12-51), one view
                R, tr = E[0, :3, :3], E[0, :3, 3]
                pp = (pc @ R.T + tr) @ K[0].T
                z = pp[:, 2]
                u = (pp[:, 0] / (z + 1e-8)).round().long()
                v = (pp[:, 1] / (z + 1e-8)).round().long()
                mk = (u >= 0) & (u < W) & (v >= 0) & (v < H) & (z > 0)
                u, v, z, c = u[mk], v[mk], z[mk], colors[mk]
                si = torch.argsort(z, descending=True)
                canvas = torch.zeros(H, W, 3, dtype=torch.uint8, device=dev)
                canvas[v[si], u[si]] = c[si].clamp(0, 255).to(torch.uint8)
                return canvas
            ms5, _ = timeit(ref_style, n=5)
            rec["torch_argsort_ms_same_gpu"] = T * ms5
            rec["torch_argsort_note"] = f"the reference's sort + scatter formulation through torch on this GPU: {ms5:.2f} ms per view x {T} views"
            if with_cpu:
                rec["cpu_oracle_ms"] = cpu_baseline_scorer(pc, colors, K, E, H, W, T)
        out[kind] = rec
        del pc, colors, gt, rep
    out["workload"] = "10 frames x 518 x 518 (train/01_preference_pair.py:33-34): 2 683 240 points re-projected into 10 views"
    out["bound"] = ("project_zbuf: one 64-bit atomicMin per point-view at device scope (executed beyond the XCD's L2); tools/zbuf_atomic_probe.hip measures the same "
                    "pattern with the loads and the arithmetic taken away (profiles/r06_scorer_*)")
    return out


def cpu_baseline_scorer(pc, colors, K, E, H, W, T):
    """CPU leg of the scorer block: oracle/scorer.py::project_points (numpy) for ONE view on the host, scaled to the T views of a video"""
    from oracle import scorer as osc
    a = [x.cpu().numpy() for x in (pc, colors, K[0], E[0])]
    t0 = time.perf_counter()
    osc.project_points(a[0], a[1], a[2], a[3], H, W)
    return (time.perf_counter() - t0) * 1e3 * T


OTHER_CONFIGS = ("cfg3", "cfg3_batch2", "cfg4", "cfg5")


CHILD_CPUS = 16      # host threads fenced off for the secondary-config child processes while the CPU leg is being timed


def split_host_cpus():
    """(cpus for the CPU leg, cpus for the child benches): the children get the CHILD_CPUS highest-numbered hardware threads this process may run on, the
    CPU leg everything else -- the two overlap in TIME (so that the default run still ends within minutes) but never share a core.  On a host too small
    to split (<= 2 x CHILD_CPUS threads) both get everything and the report says so."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None, None
    if len(cpus) <= 2 * CHILD_CPUS:
        return None, None
    return cpus[:-CHILD_CPUS], cpus[-CHILD_CPUS:]


def pin_all_threads(cpus):
    """sched_setaffinity(0, ...) moves only the calling thread on Linux: torch's OpenMP / intra-op workers created earlier keep their old mask and could
    still run on the children's cores.  Apply the mask to every thread of this process (/proc/self/task); threads created later inherit it from their creator."""
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
        except (OSError, ProcessLookupError):
            pass                       # a thread that exited meanwhile


def run_other_configs(steps=3, warmup=1, timeout_s=420, cpus=None):
    """The secondary BASELINE configurations as driver-witnessed numbers: each runs as its own `python bench.py --config cfgN` process (so the
    memory of one is gone before the next starts -- cfg4 needs 230 GB -- and a failure in one cannot touch the headline line), `steps` timed steps
    after `warmup`, no CPU leg.  cpus: the hardware threads the children are pinned to (and their OMP / MKL pools sized for), so that they cannot take
    cores from the CPU leg timed meanwhile.  Returns {name: compact record}; a config that fails or times out is recorded as {"error": ...}."""
    res = {}
    env = dict(os.environ)
    if cpus:        # the child pins ITSELF first thing in main() (no preexec_fn: this is called from a thread while the CPU leg's OpenMP pool is busy)
        env.update(OMP_NUM_THREADS=str(len(cpus)), MKL_NUM_THREADS=str(len(cpus)), VGPA_BENCH_CPUS=",".join(str(c) for c in cpus))
    for name in OTHER_CONFIGS:
        extra = []
        if name == "cfg3_batch2":       # the reference's own I2V setting: two pairs per step, no accumulation (train/CogVideoX-I2V-5B/03_train.py:59-60)
            name_arg, extra = "cfg3", ["--pairs", "2"]
        else:
            name_arg = name
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name_arg, "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline",
               "--no-other-configs", *extra]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
            line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
            if r.returncode != 0 or line is None:
                res[name] = {"error": f"exit {r.returncode}", "stderr_tail": r.stderr[-400:]}
                continue
            j = json.loads(line)
            keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "loss", "max_memory_gb", "step_flops_algorithmic",
                    "step_mfma_frac", "step_mfma_frac_note", "roofline", "roofline_worst", "attention_bwd_pair", "energy", "memory_policy", "tuned_gemms")
            rec = {k: j[k] for k in keep if k in j}
            rec["workload"] = j["config"]["workload"]
            rec["wall_s_incl_model_build"] = time.perf_counter() - t0
            res[name] = rec
        except subprocess.TimeoutExpired:
            res[name] = {"error": f"timeout after {timeout_s} s"}
        except Exception as e:                                   # noqa: BLE001 -- the headline line must still be printed
            res[name] = {"error": repr(e)}
    return res


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    have = torch.cuda.device_count()
    if have < n:
        sys.exit(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible; refusing to run fewer ranks than asked")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = env.get("MASTER_PORT", "29511")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.exit(subprocess.call(cmd, env=env))


def main_wan(args, C, world, rank, dev, force_dist):
    """cfg5: one Diffusion-DPO pair step of Wan2.2-TI2V-5B per rank (train/Wan2.2-TI2V-5B/03_train.py:189-242): shifted-sigma noising, clean first
    latent frame, frozen-reference pass and policy pass over win + lose as one batch of two samples, DPO loss, backward to the LoRA r=64 adapters
    on q/k/v/o of both attentions of every block, flat all-reduce (N > 1), clip + AdamW.  All activations resident (no block recompute)."""
    from videogpa_amd import ops
    from videogpa_amd.trainer import DPOEngine
    from videogpa_amd.wan import WanDPOTrainer
    from videogpa_amd.wan_model import WanModel
    F_ = C["frames"] if args.frames is None else args.frames
    H_ = C["height"] if args.height is None else args.height
    W_ = C["width"] if args.width is None else args.width
    layers = 30 if args.layers is None else args.layers
    ckpt = bool(args.checkpoint)
    stride = args.checkpoint_stride or 1
    torch.manual_seed(0)                                 # identical base weights and adapter init on every rank
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            model = WanModel(num_layers=layers)          # TI2V-5B defaults: dim 3072, ffn 14336, 24 heads, in / out 48, text 512
    finally:
        torch.set_default_dtype(prev)
    with torch.no_grad():
        torch.nn.init.normal_(model.head.head.weight, std=0.02)      # upstream zero-inits the output layer: give the loss a signal
    if args.weights == "trained_like":      # as for cfg2: QK-norm gains of --qk-gain (+- 20 %), three 3 x outlier channels per head (whole-row RMS-norm: weight [dim])
        gq = torch.Generator(device=dev).manual_seed(7)
        with torch.no_grad():
            for blk in model.blocks:
                for att in (blk.self_attn, blk.cross_attn):
                    for nrm in (att.norm_q, att.norm_k):
                        w = args.qk_gain * (1 + 0.2 * torch.randn(model.dim, generator=gq, device=dev))
                        w.view(model.num_heads, -1)[:, :3] *= 3.0
                        nrm.weight.copy_(w.to(nrm.weight.dtype))
    model.enable_fp8(not args.no_fp8, attention="auto" if (args.fp8_attn_auto and not args.no_fp8) else not (args.no_fp8 or args.no_fp8_attn))
    trainer = WanDPOTrainer({"lora_rank": args.rank_r, "lora_alpha": 2.0 * args.rank_r, "accumulate_grad_batches": 1, "seed": 1234,
                             "enable_gradient_checkpointing": ckpt, "gradient_checkpointing_stride": stride, "tuned_gemms": not args.no_tuned_gemms}, model)
    gB = torch.Generator(device=dev).manual_seed(1)
    with torch.no_grad():
        for n, p in trainer.transformer.named_parameters():
            if ".lora_B." in n:
                p.normal_(0.0, 1e-3, generator=gB)
    trainer.train()
    engine = DPOEngine(trainer)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    lat = lambda f: torch.randn(1, 48, f, H_, W_, generator=g, device=dev).to(torch.bfloat16)
    batch = {"x_win": lat(F_), "x_lose": lat(F_), "prompt_emb": torch.randn(1, 300, 4096, generator=g, device=dev).to(torch.bfloat16), "image_latent": lat(1)}
    L_tok = F_ * (H_ // 2) * (W_ // 2)
    F_step = flops_per_pair_step_wan(L_tok, model.dim, model.ffn_dim, WAN_TEXT_LEN, layers, args.rank_r)
    F_fp8 = flops_fp8_wan(L_tok, model.dim, model.ffn_dim, layers, ffn_fp8=not args.no_fp8, attn_fp8=not (args.no_fp8 or args.no_fp8_attn))

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        engine.micro_step(batch)
    engine.flush()
    engine.opt.comm_report()          # drop the warm-up's exchange events: `comm` covers the timed steps only
    if rank == 0 and not args.no_kernel_timer:
        ops.TIMER = ops.KernelTimer(full_steps=TIMER_FULL_STEPS, always=TIMER_ALWAYS)
    barrier()
    e0 = read_joules(dev.index or 0) if rank == 0 else None
    t0 = time.perf_counter()
    logs = None
    for _ in range(args.steps):
        logs = engine.micro_step(batch)
        if ops.TIMER is not None:
            ops.TIMER.next_step()
    logs.update(engine.flush())
    barrier()
    dt_local = time.perf_counter() - t0
    e1 = read_joules(dev.index or 0) if rank == 0 else None
    dt = dt_local
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1 or force_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    drep = dist_report(engine, dt_local, args.steps, dev, world, force_dist)
    if getattr(args, "preflight", None):
        drep["preflight"] = args.preflight
    if rank == 0:
        named = (layers, F_, H_, W_, args.rank_r, ckpt, args.no_fp8, args.no_fp8_attn) == (30, C["frames"], C["height"], C["width"], 64, False, False, False) \
            and args.weights == "bench" and not args.fp8_attn_auto
        ms = dt / args.steps * 1e3
        out = {
            "metric": "DPO preference-pair steps/sec, Wan2.2-TI2V-5B 81f@704x1280", "value": world * args.steps / dt, "unit": "pair-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.no_fp8 else ("bf16 (attention, LoRA-carrying projections) + fp8 e4m3 (feed-forward GEMMs)" if args.no_fp8_attn else
                                                 "fp8 e4m3 (self-attention forward: hand-written MFMA kernel; feed-forward GEMMs) + bf16 (attention backward, cross-attention, "
                                                 "LoRA-carrying projections)"),
            "data": "synthetic",
            "config": {"workload": (C["label"] + " -> " if named else "NOT a BASELINE config (debug flags): Wan2.2-shaped denoiser, ")
                                   + f"paired latents 2 x [1,48,{F_},{H_},{W_}], {L_tok} tokens, {layers} blocks, LoRA r={args.rank_r} on q/k/v/o of self- and "
                                   "cross-attention, 1 pair/GPU/step, optimizer step every step; random-init weights"
                                   + (f"; activation recompute of every {stride}. block" if ckpt else "; all activations resident (no recompute)"),
                       "name": "cfg5", "layers": layers, "tokens": L_tok, "pairs_per_gpu": 1, "parallelism": f"dp{world}"},
            "note": "NOT the headline line: BASELINE.json's metric is quoted on cfg2 (python bench.py without --config)",
            "loss": float(logs["train/loss"]), "loss_rank_mean": logs["sync"].tolist()[0],
            "step_flops_algorithmic": F_step,
            "step_flops_algorithmic_fp8": F_fp8,
            # priced against the FLOP-weighted BLENDED peak of the step: the e4m3 share at 5 PF, the rest at 2.5 PF (round 5 priced the whole step at the bf16 peak)
            "step_mfma_frac": (F_fp8 / (PEAK_FP8_DENSE_TFLOPS * 1e12) + (F_step - F_fp8) / (PEAK_BF16_DENSE_TFLOPS * 1e12)) / (dt / args.steps),
            "step_mfma_frac_note": f"time at peak / time taken with {F_fp8 / F_step:.3f} of the algorithmic FLOPs priced at the dense fp8 peak (5 PF) and the rest at the dense "
                                   f"bf16 peak (2.5 PF); against the bf16 peak alone the same step is {F_step / (dt / args.steps) / (PEAK_BF16_DENSE_TFLOPS * 1e12):.3f}",
            "max_memory_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
            "tuned_gemms": tuned_gemms_report(ops),
        }
        frep = model.fp8_attention_report()
        est = [r["estimated_score_error_log2"] for r in frep if r["estimated_score_error_log2"] is not None]
        out["fp8_attention"] = {"weights": args.weights + (f" (QK-norm gain {args.qk_gain})" if args.weights == "trained_like" else ""),
                                "mode": "auto" if args.fp8_attn_auto and not args.no_fp8 else "pinned", "layers_on_e4m3": sum(r["fp8_attn"] for r in frep), "layers": len(frep),
                                "estimated_score_error_log2_max": max(est) if est else None}
        out.update(drep)
        add_kernel_report(out, ops, args.steps, ms, use_pmc=named)
        if not ckpt:
            out["energy"] = energy_report(e0, e1, dt_local, args.steps, F_step)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2", help="BASELINE.json configuration (cfg2 = the headline)")
    ap.add_argument("--layers", type=int, default=None, help="debug only: anything but the model's own depth (42; cfg5: 30) is NOT a BASELINE config")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--rank-r", type=int, default=64)
    ap.add_argument("--weights", choices=["bench", "trained_like"], default="bench", help="trained_like: QK-norm gains of --qk-gain with a few 3 x outlier channels and biases (low-entropy "
                    "attention rows: what the bound-shifted forward has to cope with on a trained checkpoint) instead of the gains of 1 a random init gives; NOT a BASELINE config")
    ap.add_argument("--qk-gain", type=float, default=2.5)
    ap.add_argument("--pairs", type=int, default=1, help="preference pairs per GPU and step (the reference's I2V trainer runs 2: train/CogVideoX-I2V-5B/03_train.py:59-60); "
                    "more than one lets the trainer's memory policy choose lean activations")
    ap.add_argument("--checkpoint", action="store_true", default=None, help="per-block activation recompute (needed beyond ~22k tokens per sequence)")
    ap.add_argument("--no-checkpoint", dest="checkpoint", action="store_false", help="keep every block's activations (overrides a config's default recompute)")
    ap.add_argument("--checkpoint-stride", type=int, default=None, help="with --checkpoint: recompute only every k-th block (1 = all, like the reference)")
    ap.add_argument("--lean", action="store_true", default=None, help="lean activations: the LN output and the normalised q / k are made again in the backward (23 %% fewer saved bytes per block)")
    ap.add_argument("--no-fp8", action="store_true", help="cfg5 only: bf16 feed-forward GEMMs and bf16 attention instead of the e4m3 paths")
    ap.add_argument("--no-fp8-attn", action="store_true", help="cfg5 only: keep the e4m3 feed-forward but run the self-attention forward in bf16")
    ap.add_argument("--fp8-attn-auto", action="store_true", help="cfg5 only: enable_fp8(attention='auto') -- every self-attention layer decides from its first call's score range "
                    "whether e4m3 scores are accurate enough (ops.F8AttnPolicy); NOT the BASELINE line (which pins the e4m3 kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="run the vendor GEMMs on hipBLASLt's default heuristic instead of the solutions of videogpa_amd/tuned/ (the A/B of tools/gemm_tune.py)")
    ap.add_argument("--no-collective-ab", action="store_true", help="N > 1: skip the second short timed region that runs the other gradient-exchange form (all_reduce <-> rs_ag)")
    ap.add_argument("--no-scorer", action="store_true", help="skip the geometry-scorer block (a few milliseconds of GPU work after the headline's timed region)")
    ap.add_argument("--no-other-configs", action="store_true", help="default run (1 GPU, cfg2, no debug flags) also measures cfg3 / cfg4 / cfg5 for 3 steps each "
                    "and attaches them as `other_configs`; this switches that off")
    args = ap.parse_args()
    if os.environ.get("VGPA_BENCH_CPUS"):        # a secondary-config child of the default run: stay off the cores the parent's CPU leg is being timed on
        try:
            pin_all_threads({int(c) for c in os.environ["VGPA_BENCH_CPUS"].split(",")})
        except (AttributeError, OSError, ValueError):
            pass

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} are visible")
    force_dist = os.environ.get("VGPA_FORCE_DIST") == "1"       # exercise the RCCL path on a single GPU (debug)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "VERSION")      # RCCL prints its version line to stderr when the communicator comes up
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pre = None
    if world > 1 or force_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
        if rank == 0:
            print(f"bench.py: RCCL process group up, {dist.get_world_size()} ranks", file=sys.stderr, flush=True)
        pre = preflight(world, rank, local_rank, dev)
    args.preflight = pre

    from videogpa_amd import ops, transformer as vtr
    from videogpa_amd.trainer import CogVideoXDPOTrainer, DPOEngine

    C = CONFIGS[args.config]
    if args.config == "cfg5":
        return main_wan(args, C, world, rank, dev, force_dist)
    if args.layers is None:
        args.layers = 42
    F_ = C["frames"] if args.frames is None else args.frames
    H_ = C["height"] if args.height is None else args.height
    W_ = C["width"] if args.width is None else args.width
    ckpt = C["checkpoint"] if args.checkpoint is None else args.checkpoint
    ckpt_stride = (C.get("checkpoint_stride", 1) if args.checkpoint_stride is None else args.checkpoint_stride) if ckpt else 1
    cfg_kw = dict(getattr(vtr, C["model"]), num_layers=args.layers)
    torch.manual_seed(0)                           # identical adapter init (PEFT kaiming-uniform A) on every rank
    model = build_model(cfg_kw, dev, seed=0)       # identical base weights on every rank
    if args.weights == "trained_like":
        gq = torch.Generator(device=dev).manual_seed(7)
        with torch.no_grad():
            for blk in model.transformer_blocks:
                for nrm in (blk.attn1.norm_q, blk.attn1.norm_k):
                    w = args.qk_gain * (1 + 0.2 * torch.randn(64, generator=gq, device=dev))
                    w[:3] *= 3.0
                    nrm.weight.copy_(w.to(nrm.weight.dtype))
                    nrm.bias.copy_((0.1 * args.qk_gain * torch.randn(64, generator=gq, device=dev)).to(nrm.bias.dtype))
    lean = bool(C.get("lean", False) if args.lean is None else args.lean)
    P_ = max(1, args.pairs)
    lean_cfg = "auto" if (P_ > 1 and args.lean is None) else lean        # more than one pair per step: the trainer's own memory policy decides (and is reported)
    trainer = CogVideoXDPOTrainer({"lora_rank": args.rank_r, "lora_alpha": 2 * args.rank_r, "beta": 1.0, "accumulate_grad_batches": 1,
                                   "enable_gradient_checkpointing": ckpt, "gradient_checkpointing_stride": ckpt_stride, "lean_activations": lean_cfg,
                                   "seed": 1234, "tuned_gemms": not args.no_tuned_gemms}, transformer=model)
    # LoRA B ~ N(0, 1e-3) so the step is beyond the trivial B=0 point
    gB = torch.Generator(device=dev).manual_seed(1)
    with torch.no_grad():
        for n, p in trainer.transformer.named_parameters():
            if ".lora_B." in n:
                p.normal_(0.0, 1e-3, generator=gB)
    trainer.train()
    engine = DPOEngine(trainer)

    # synthetic preference pair, resident in HBM (seed 1234 + rank): every rank has its own pair and its own (t, eps) stream
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x_pair = (0.7 * torch.randn(P_, 2, F_, 16, H_, W_, generator=g, device=dev)).to(torch.bfloat16)
    prompt = (0.2 * torch.randn(P_, TEXT_LEN, 4096, generator=g, device=dev)).to(torch.bfloat16)
    batch = {"x_pair": x_pair, "prompt_emb": prompt}
    if C["cond"]:
        batch["image_latent"] = (0.7 * torch.randn(P_, 1, 16, H_, W_, generator=g, device=dev)).to(torch.bfloat16)

    pt = cfg_kw.get("patch_size_t") or 1
    Fe, He, We = (F_ - F_ % 2, H_ - H_ % 2, W_ - W_ % 2) if pt > 1 else (F_, H_, W_)     # the 1.5 step even-crops
    S = TEXT_LEN + (Fe // pt) * (He // 2) * (We // 2)
    D = cfg_kw["num_attention_heads"] * 64
    F_step = P_ * flops_per_pair_step(S, D, args.layers, args.rank_r)

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        engine.micro_step(batch)
    engine.flush()
    engine.opt.comm_report()          # drop the warm-up's exchange events: `comm` covers the timed steps only
    if rank == 0 and not args.no_kernel_timer:
        ops.TIMER = ops.KernelTimer(full_steps=TIMER_FULL_STEPS, always=TIMER_ALWAYS)
    barrier()
    e0 = read_joules(dev.index or 0) if rank == 0 else None
    t0 = time.perf_counter()
    logs = None
    for _ in range(args.steps):
        logs = engine.micro_step(batch)
        if ops.TIMER is not None:
            ops.TIMER.next_step()
    logs.update(engine.flush())       # the last optimizer step (applied one micro-step late when the all-reduce is overlapped) is inside the timed region
    barrier()
    dt_local = time.perf_counter() - t0
    e1 = read_joules(dev.index or 0) if rank == 0 else None
    dt = dt_local
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1 or force_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    drep = dist_report(engine, dt_local, args.steps, dev, world, force_dist)
    if (world > 1 or force_dist) and not args.no_collective_ab:
        ab = collective_ab(engine, batch, min(args.steps, 3), world, force_dist, dev, barrier)
        if drep.get("comm"):
            drep["collective_ab"] = dict({drep["comm"]["collective"]: {"steps": args.steps, "ms_per_step": dt / args.steps * 1e3, "comm": drep["comm"]}}, **ab)
    if args.preflight:
        drep["preflight"] = args.preflight

    if rank == 0:
        named = (args.layers, F_, H_, W_, args.rank_r, ckpt) == (42, C["frames"], C["height"], C["width"], 64, C["checkpoint"]) and \
            (lean == bool(C.get("lean", False)) or P_ > 1) and args.weights == "bench"
        lean = bool(trainer.transformer.get_base_model().lean_activations)     # what actually ran (the memory policy may have chosen)
        ms = dt / args.steps * 1e3
        value = world * args.steps * P_ / dt
        sync = logs["sync"].tolist()
        out = {
            "metric": "DPO preference-pair steps/sec, " + C["metric"], "value": value, "unit": "pair-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (C["label"] + " -> " if named else "NOT a BASELINE config (debug flags): CogVideoX-5B-shaped transformer, ")
                                   + f"paired latents [{P_},2,{F_},16,{H_},{W_}], S={S} tokens, {args.layers} blocks, LoRA r={args.rank_r} on "
                                   f"to_q/to_k/to_v/to_out.0, {P_} pair{'s' if P_ > 1 else ''}/GPU/step, optimizer step every step; random-init weights"
                                   + (f"; activation recompute of every {ckpt_stride}. block" if ckpt and ckpt_stride > 1 else
                                      "; per-block activation recompute" if ckpt else "")
                                   + ("; lean activations (LN output and normalised q / k made again in the backward)" if lean else ""),
                       "name": args.config, "layers": args.layers, "tokens": S, "pairs_per_gpu": P_, "parallelism": f"dp{world}"},
            "loss": float(logs["train/loss"]), "loss_rank_mean": sync[0], "memory_policy": trainer.memory_policy_log,
            "attention_forward": attention_forward_summary(trainer.transformer.get_base_model(), args),
            "step_flops_algorithmic": F_step,
            "step_mfma_frac": F_step / (dt / args.steps) / (PEAK_BF16_DENSE_TFLOPS * 1e12),
            "max_memory_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
            "tuned_gemms": tuned_gemms_report(ops),
        }
        out.update(drep)
        add_kernel_report(out, ops, args.steps, ms, use_pmc=named and args.config in ("cfg2", "cfg3"))
        if not ckpt:      # with block recompute the executed count would also carry the recomputed forwards
            out["energy"] = energy_report(e0, e1, dt_local, args.steps, F_step, F_step + P_ * 6.0 * args.layers * 2 * cfg_kw["num_attention_heads"] * 64.0 * S * S)
        if world == 1 and not force_dist and named and args.config == "cfg2" and not args.no_scorer:
            out["scorer"] = scorer_report(dev, with_cpu=not args.no_cpu_baseline)
        others = None
        if world == 1 and not force_dist and named and args.config == "cfg2" and not args.no_other_configs:
            # the secondary configurations run on the (now idle) GPU in child processes WHILE the host cores time the CPU leg: the timed region of the
            # headline is over, this process's device memory is released first, and the two legs are fenced onto DISJOINT host cores (split_host_cpus):
            # the children (each builds a 5 B-parameter model on the host before its timed steps) cannot slow the CPU baseline down
            import gc
            import threading
            del engine, trainer, model, batch, x_pair, prompt, logs
            gc.collect()
            torch.cuda.empty_cache()
            box = {}
            leg_cpus, child_cpus = split_host_cpus()
            others = threading.Thread(target=lambda: box.update(run_other_configs(cpus=child_cpus)), daemon=True)
            others.start()
            if leg_cpus and not args.no_cpu_baseline:
                pin_all_threads(leg_cpus)                  # this process is done with the GPU work that needed its launch threads
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(F_step)
            if others is not None:
                out["cpu_baseline"]["legs_overlapped"] = True
                out["cpu_baseline"]["host_cpu_fence"] = ({"cpu_leg_threads_available": len(leg_cpus), "child_bench_threads": len(child_cpus)} if leg_cpus else
                                                         "host too small to split: the CPU leg shared its cores with the secondary-config children")
        if others is not None:
            others.join()
            out["other_configs"] = box
            out["other_configs_note"] = ("cfg3 / cfg4 / cfg5 measured by this same invocation after the headline's timed region (3 timed steps each, "
                                         "1 warm-up, own process per config, concurrently with the CPU leg but pinned to their own host cores); the "
                                         "headline fields above are cfg2 only")
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
