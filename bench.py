#!/usr/bin/env python
"""Headline benchmark: DPO preference-pair steps/sec, CogVideoX-5B 49f@480x720 (BASELINE.json configs[1]).

One step = one preference pair per rank through the whole hot path: fused noising + v-targets, frozen-reference
forward (win+lose), policy forward (win+lose) with LoRA r=64, Diffusion-DPO loss, backward to the 66 M LoRA
parameters, flat-buffer gradient all-reduce (N>1), fused clip + AdamW.  Synthetic latents / prompt embeddings of the
named shape, random-init weights of the named architecture (no network for checkpoints), all resident in HBM before
the timed region.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
TEXT_LEN = 226


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


def flops_per_pair_step(S, D, L, r):
    """BASELINE.md section 4 (algorithmic, no recompute counted)."""
    f_lin, f_attn, f_lora = 24.0 * S * D * D, 4.0 * S * S * D, 16.0 * S * D * r
    fwd = L * (f_lin + f_attn)
    return 2 * fwd + 2 * (fwd + L * f_lora) + 2 * L * (f_lin + 2 * f_attn + 2 * f_lora)


def cpu_baseline(F_step):
    """Bounded CPU sample of the same path with the oracle (kind 'port'): one CogVideoX-5B-geometry transformer block
    forward (fp32, D=3072, 48 heads, text 226 + 4096 video tokens) on the host cores, scaled to a full pair-step by the
    algorithmic-FLOP ratio."""
    from oracle import cogvideox as ocv
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = ocv.CogVideoXConfig(num_layers=1)
    D, Sv, Lt = cfg.inner_dim, 4096, TEXT_LEN
    sd = ocv.init_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(0)
    hid = torch.randn(1, Sv, D, generator=g)
    enc = torch.randn(1, Lt, D, generator=g)
    temb = torch.randn(1, cfg.time_embed_dim, generator=g)
    S = Sv + Lt
    f_block = 24.0 * S * D * D + 4.0 * S * S * D
    with torch.no_grad():
        t0 = time.time()
        ocv.block_forward(sd, cfg, 0, hid, enc, temb)
        dt = time.time() - t0
    est_step_s = dt * (F_step / f_block)
    return {"value": 1.0 / est_step_s, "unit": "pair-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fp32 forward of 1 CogVideoX-5B block at S={S} tokens took {dt:.2f} s ({f_block / dt / 1e9:.0f} GFLOP/s); "
                      f"extrapolated to the {F_step:.3g}-FLOP pair-step by algorithmic-FLOP ratio"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=42, help="debug only: anything but 42 is NOT the headline config")
    ap.add_argument("--frames", type=int, default=13)
    ap.add_argument("--height", type=int, default=60)
    ap.add_argument("--width", type=int, default=90)
    ap.add_argument("--rank-r", type=int, default=64)
    ap.add_argument("--checkpoint", action="store_true", help="per-block activation recompute (needed beyond ~22k tokens per sequence)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("VGPA_FORCE_DIST") == "1"       # exercise the RCCL path on a single GPU (debug)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or force_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm

    from videogpa_amd import ops
    from videogpa_amd.trainer import CogVideoXDPOTrainer, DPOEngine
    from videogpa_amd.transformer import COGVIDEOX_5B

    cfg_kw = dict(COGVIDEOX_5B, num_layers=args.layers)
    torch.manual_seed(0)                           # identical adapter init (PEFT kaiming-uniform A) on every rank
    model = build_model(cfg_kw, dev, seed=0)       # identical base weights on every rank
    trainer = CogVideoXDPOTrainer({"lora_rank": args.rank_r, "lora_alpha": 2 * args.rank_r, "beta": 1.0, "accumulate_grad_batches": 1,
                                   "enable_gradient_checkpointing": args.checkpoint}, transformer=model)
    # LoRA B ~ N(0, 1e-3) so the step is beyond the trivial B=0 point
    gB = torch.Generator(device=dev).manual_seed(1)
    with torch.no_grad():
        for n, p in trainer.transformer.named_parameters():
            if ".lora_B." in n:
                p.normal_(0.0, 1e-3, generator=gB)
    trainer.train()
    engine = DPOEngine(trainer)

    # synthetic preference pair, resident in HBM (seed 1234 + rank)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    F_, H_, W_ = args.frames, args.height, args.width
    x_pair = (0.7 * torch.randn(1, 2, F_, 16, H_, W_, generator=g, device=dev)).to(torch.bfloat16)
    prompt = (0.2 * torch.randn(1, TEXT_LEN, 4096, generator=g, device=dev)).to(torch.bfloat16)
    batch = {"x_pair": x_pair, "prompt_emb": prompt}

    S = TEXT_LEN + F_ * (H_ // 2) * (W_ // 2)
    D = cfg_kw["num_attention_heads"] * 64
    F_step = flops_per_pair_step(S, D, args.layers, args.rank_r)

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        engine.micro_step(batch)
    if rank == 0 and not args.no_kernel_timer:
        ops.TIMER = ops.KernelTimer()
    barrier()
    t0 = time.perf_counter()
    logs = None
    for _ in range(args.steps):
        logs = engine.micro_step(batch)
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1 or force_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    if rank == 0:
        headline = (args.layers, args.frames, args.height, args.width, args.rank_r, args.checkpoint) == (42, 13, 60, 90, 64, False)
        ms = dt / args.steps * 1e3
        value = world * args.steps / dt
        out = {
            "metric": "DPO preference-pair steps/sec, CogVideoX-5B 49f@480x720", "value": value, "unit": "pair-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: CogVideoX-5B T2V full (42 blocks, D=3072, 48x64 heads), 49f x 480x720 -> "
                                    if headline else "NOT the headline config (debug flags): CogVideoX-5B-shaped transformer, ")
                                   + f"paired latents [1,2,{F_},16,{H_},{W_}], S={S} tokens, {args.layers} blocks, LoRA r={args.rank_r} on "
                                   "to_q/to_k/to_v/to_out.0, 1 pair/GPU/step, optimizer step every step; random-init weights"
                                   + ("; per-block activation recompute" if args.checkpoint else ""),
                       "layers": args.layers, "tokens": S, "pairs_per_gpu": 1, "parallelism": f"dp{world}"},
            "loss": float(logs["train/loss"]),
            "step_flops_algorithmic": F_step,
            "step_mfma_frac": F_step / (dt / args.steps) / (PEAK_BF16_DENSE_TFLOPS * 1e12),
            "max_memory_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
        }
        if ops.TIMER is not None:
            summ = ops.TIMER.summary()
            ops.TIMER = None
            kernels = {}
            for name, s in summ.items():
                tf = s["work_per_launch"] / (s["avg_ms"] * 1e-3) / 1e12
                kernels[name] = {"launches": s["launches"], "avg_ms": s["avg_ms"], "total_ms_per_step": s["total_ms"] / args.steps,
                                 "algorithmic_flops_per_launch": s["work_per_launch"], "achieved_tflops": tf}
            dom = max(kernels, key=lambda k: kernels[k]["total_ms_per_step"])
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": kernels[dom]["achieved_tflops"], "peak": PEAK_BF16_DENSE_TFLOPS,
                               "unit": "TFLOP/s", "frac": kernels[dom]["achieved_tflops"] / PEAK_BF16_DENSE_TFLOPS, "traffic": None,
                               "avg_launch_ms": kernels[dom]["avg_ms"]}
            out["kernels"] = kernels
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(F_step)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
