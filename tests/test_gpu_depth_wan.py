"""FULL-DEPTH parity of BASELINE.json configs[4] as `bench.py --config cfg5` runs it: Wan2.2-TI2V-5B (30 blocks, dim 3072, 24 x 128 heads, ffn 14 336, text 512),
paired latents 2 x [1,48,21,44,80] -> 18 480 tokens per sample, the first latent frame clean, LoRA r = 64 on q / k / v / o of self- and cross-attention
(480 tensors), lora_B ~ N(0, 1e-3) -- the HIP pair-step (train/Wan2.2-TI2V-5B/03_train.py:189-242 through WanDPOTrainer._shared_step) against
oracle/steps.py::wan_pair_step over oracle/wan.py run ON THE GPU in fp32 at the same depth (per-block torch.utils.checkpoint as the reference trains, :151-160, and
the head-chunked exact attention; pinned to the plain oracle in tests/test_oracle_kat.py).

Two device modes, each against the oracle that makes the same roundings at the same places AND against plain fp32:
    bf16   enable_fp8(False)                                            vs  round_activations, exact_delta
    fp8    enable_fp8(True): e4m3 feed-forward GEMMs + e4m3 self-attention forward, consistent backward   vs  + fp8_ffn, f8_attn
Compared: loss (1e-3, north_star) and rewards, the four predictions, EVERY LoRA gradient tensor -- relative error and cosine -- as a function of block index
(gpurun_out/cfg5_depth_parity_<mode>.json -> profiles/r06_cfg5_depth_parity_*.json).  Bounds next to the asserts; everything is written before the first assert.

What the first full-depth run showed (round 6) and how the gradient bounds are therefore stated.  With bench.py's weights (upstream init: modulation tables ~ dim^-1/2,
attention logits nearly flat) the loss agrees to 1e-5 and the predictions to 0.3-0.5 % of their range, but some adapter gradients are SMALL BY CANCELLATION -- the
self-attention q / k adapters have norms of 1e-7, a hundred times below the others, the cross-attention k adapters see 300 real + 212 zero-padded text keys -- and for
those the activation-rounded ORACLE (fp32 arithmetic that merely rounds the tensors the device stores in bf16 / e4m3) is itself 10-33 % (bf16) and 50-94 % (fp8) from
plain fp32: that is what the arithmetic type costs on an ill-conditioned quantity, whoever computes it.  So:
  * tensors the rounded oracle keeps within 5 % of fp32 (cross o / q, self o / v, ...: the well-conditioned ones) are held to the cfg1 bounds: 10 % / cosine 0.995
    against the matching oracle, 12 % (bf16) / 20 % (fp8) against fp32;
  * EVERY tensor: the HIP path is no further from fp32 than 1.5 x the matching oracle is (+ 0.02; measured <= 1.41 x), and no further from that oracle than two
    independent realisations of the same rounding noise are (1.6 x the larger of the two distances from fp32, + 0.02).

-m gpu only; the HIP step takes 181 GB, the oracles ~80 GB afterwards; ~10 minutes per mode."""
import gc
import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LAYERS = int(os.environ.get("VGPA_DEPTH_LAYERS", "30"))         # debug knob; anything but 30 is labelled in the report
GRID = (21, 44, 80)
RANK, ALPHA = 64, 128.0
TIMESTEP = 613
CFG = dict(model_type="ti2v", patch_size=(1, 2, 2), text_len=512, in_dim=48, dim=3072, ffn_dim=14336, freq_dim=256, text_dim=4096, out_dim=48,
           num_heads=24, num_layers=LAYERS, cross_attn_norm=True, eps=1e-6)

LOSS_TOL = 1e-3
PRED_ERR_OVER_RANGE = 0.04
ROUNDED_REL, ROUNDED_COS = 0.10, 0.995             # the cfg1 bound (tests/test_gpu_wan_cfg1.py), unchanged at 15 x the depth
FP32_REL_CAP = {"bf16": 0.12, "fp8": 0.20}         # against plain fp32: the cfg1 cap for bf16; e4m3 operands cost 4.75 % at 2 blocks and are not held to the bf16 cap
WELL_CONDITIONED, VS_ORACLE_DISTANCE, TWO_REALISATIONS = 0.05, 1.5, 1.6       # see the header


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _clean():
    gc.collect()
    torch.cuda.empty_cache()


def _inputs():
    g = torch.Generator(device="cuda").manual_seed(1234)         # bench.py::main_wan
    Fr, H, W = GRID
    lat = lambda f: torch.randn(1, 48, f, H, W, generator=g, device="cuda").to(torch.bfloat16)
    batch = {"x_win": lat(Fr), "x_lose": lat(Fr), "prompt_emb": torch.randn(1, 300, 4096, generator=g, device="cuda").to(torch.bfloat16), "image_latent": lat(1)}
    noise = torch.randn(1, 48, Fr, H, W, generator=g, device="cuda").to(torch.bfloat16)
    return batch, torch.tensor([TIMESTEP], device="cuda"), noise


def _hip_step(mode):
    from videogpa_amd.wan import WanDPOTrainer
    from videogpa_amd.wan_model import WanModel
    torch.manual_seed(0)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("cuda"):
            model = WanModel(num_layers=LAYERS)
    finally:
        torch.set_default_dtype(prev)
    with torch.no_grad():
        torch.nn.init.normal_(model.head.head.weight, std=0.02)          # bench.py: upstream zero-inits the output layer
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.enable_fp8(mode == "fp8", attention=(mode == "fp8"))
    tr = WanDPOTrainer({"lora_rank": RANK, "lora_alpha": ALPHA, "accumulate_grad_batches": 1, "seed": 1234, "enable_gradient_checkpointing": False}, model)
    gB = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        for n, p in tr.transformer.named_parameters():
            if ".lora_B." in n:
                p.normal_(0.0, 1e-3, generator=gB)
            if ".lora_" in n:
                p.copy_(p.bfloat16().float())                             # bf16-representable adapters on both sides
    mods = {n: m for n, m in tr.transformer.get_base_model().named_modules() if type(m).__name__ == "LoraLinear"}
    assert len(mods) == LAYERS * 8
    lora = {n: (m.lora_A["default"].weight.detach().clone(), m.lora_B["default"].weight.detach().clone()) for n, m in mods.items()}
    tr.train()
    batch, t, noise = _inputs()
    captured = []
    orig = tr.transformer.forward

    def spy(*a, **k):
        out = orig(*a, **k)
        captured.append([o.detach().float() for o in out])
        return out
    tr.transformer.forward = spy
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    out = tr._shared_step(batch, timesteps=t, noise=noise)
    tr.transformer.forward = orig
    out.loss.backward()
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    v_ref, v_pol = captured                                               # reference first (:227-229); batch = (win, lose)
    preds = {"v_win": v_pol[0], "v_lose": v_pol[1], "v_win_ref": v_ref[0], "v_lose_ref": v_ref[1]}
    grads = {n: (m.lora_A["default"].weight.grad.detach().float().clone(), m.lora_B["default"].weight.grad.detach().float().clone()) for n, m in mods.items()}
    scal = {"loss": out.loss.item(), "reward_margin": out.reward_margin.item(), "winner_reward": out.winner_reward.item(), "loser_reward": out.loser_reward.item()}
    del out, tr, model, mods, captured
    _clean()
    return scal, preds, grads, state, lora, peak, secs


def _oracle(state32, lora, **kw):
    from oracle import steps as ost
    from oracle import wan as ow
    leaves = {n: (A.clone().requires_grad_(True), Bm.clone().requires_grad_(True)) for n, (A, Bm) in lora.items()}
    P = ow.Params(state32, {n: (A, Bm, ALPHA / RANK) for n, (A, Bm) in leaves.items()}, dtype=torch.float32, checkpoint_blocks=True, chunked_attention=True, **kw)
    P0 = ow.Params(state32, {}, dtype=torch.float32, chunked_attention=True, **kw)
    batch, t, noise = _inputs()
    pol = lambda xs, t, context, seq_len: ow.forward(P, CFG, xs, t, [c.float() for c in context], seq_len)
    ref = lambda xs, t, context, seq_len: ow.forward(P0, CFG, xs, t, [c.float() for c in context], seq_len)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = ost.wan_pair_step(pol, ref, batch["x_win"].float(), batch["x_lose"].float(), batch["prompt_emb"].float(), t, noise.float(),
                            image_latent=batch["image_latent"].float(), beta=1.0)
    out["loss"].backward()
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    scal = {k: float(out[k]) for k in ("loss", "reward_margin", "winner_reward", "loser_reward")}
    preds = {k: out[k].detach()[0].float() for k in ("v_win", "v_lose", "v_win_ref", "v_lose_ref")}
    grads = {n: (A.grad.detach().float(), Bm.grad.detach().float()) for n, (A, Bm) in leaves.items()}
    del out, P, P0, leaves
    _clean()
    return scal, preds, grads, secs


def _rel(a, r):
    return float((a.double() - r.double()).norm() / r.double().norm().clamp_min(1e-300))


def _cos(a, r):
    a, r = a.double().flatten(), r.double().flatten()
    return float((a * r).sum() / (a.norm() * r.norm()).clamp_min(1e-300))


ORACLE_FOR = {"bf16": dict(round_activations=True, exact_delta=True), "fp8": dict(round_activations=True, exact_delta=True, fp8_ffn=True, f8_attn=True)}
_PLAIN = {}


def _run(mode):
    report = {"config": f"BASELINE configs[4] as bench.py --config cfg5 runs it: {LAYERS} blocks, {GRID[0] * GRID[1] * GRID[2] // 4} tokens per sample, r = {RANK} on q/k/v/o of both "
                        f"attentions, lora_B ~ N(0, 1e-3), t = {TIMESTEP}, first latent frame clean; upstream init + head N(0, 0.02)",
              "is_baseline_depth": LAYERS == 30, "mode": mode,
              "bounds": {"loss": LOSS_TOL, "pred_err_over_range": PRED_ERR_OVER_RANGE, "well_conditioned_if_oracle_within": WELL_CONDITIONED, "rounded_rel": ROUNDED_REL,
                         "rounded_cos": ROUNDED_COS, "fp32_rel_cap": FP32_REL_CAP[mode], "vs_oracle_distance": VS_ORACLE_DISTANCE, "two_realisations": TWO_REALISATIONS}}
    fails = []

    def check(ok, what):
        if not ok:
            fails.append(what)

    hip, hip_preds, hip_grads, state, lora, peak, secs = _hip_step(mode)
    report.update(hip=hip, hip_peak_gb=peak, hip_step_seconds_incl_first_call_setup=secs)
    state32 = {k: v.float() for k, v in state.items()}
    del state
    if "fp32" not in _PLAIN:                      # the plain fp32 oracle does not depend on the device mode (same weights, same adapters, same inputs)
        _PLAIN["fp32"] = _oracle(state32, lora)
    oracles = {"fp32": _PLAIN["fp32"], "rounded": _oracle(state32, lora, **ORACLE_FOR[mode])}
    del state32
    _clean()
    for name in oracles:
        report[name] = dict(oracles[name][0], seconds=oracles[name][3])
        d = abs(hip["loss"] - oracles[name][0]["loss"])
        report[f"loss_abs_err_vs_{name}"] = d
        check(d < LOSS_TOL, ("loss", name, hip["loss"], oracles[name][0]["loss"]))
        check(abs(hip["reward_margin"] - oracles[name][0]["reward_margin"]) < 1e-3, ("reward_margin", name, hip["reward_margin"], oracles[name][0]["reward_margin"]))
    report["predictions"] = {}
    for k, v in hip_preds.items():
        row = {}
        for name in oracles:
            ref = oracles[name][1][k]
            row[f"max_err_over_range_vs_{name}"] = float((v - ref).abs().max()) / float(ref.abs().max())
            row[f"rel_norm_err_vs_{name}"] = _rel(v, ref)
        report["predictions"][k] = row
        check(row["max_err_over_range_vs_rounded"] < PRED_ERR_OVER_RANGE, (k, "prediction vs rounded", row["max_err_over_range_vs_rounded"]))
        check(row["max_err_over_range_vs_fp32"] < PRED_ERR_OVER_RANGE * (2 if mode == "fp8" else 1), (k, "prediction vs fp32", row["max_err_over_range_vs_fp32"]))
    g32, gro = oracles["fp32"][2], oracles["rounded"][2]
    per_tensor, by_block = {}, []
    for i in range(LAYERS):
        rows = {}
        for n in sorted(n for n in hip_grads if n.startswith(f"blocks.{i}.")):
            for j, which in ((0, "A"), (1, "B")):
                g = hip_grads[n][j]
                e32, ero, cro = _rel(g, g32[n][j]), _rel(g, gro[n][j]), _cos(g, gro[n][j])
                rows[f"{n}.lora_{which}"] = {"rel_vs_fp32": round(e32, 5), "cos_vs_fp32": round(_cos(g, g32[n][j]), 6), "rel_vs_rounded": round(ero, 5),
                                             "cos_vs_rounded": round(cro, 6), "rounded_oracle_rel_vs_fp32": round(_rel(gro[n][j], g32[n][j]), 5),
                                             "norm_fp32": float(g32[n][j].double().norm())}
                e_ro32 = rows[f"{n}.lora_{which}"]["rounded_oracle_rel_vs_fp32"]
                well = e_ro32 <= WELL_CONDITIONED
                rows[f"{n}.lora_{which}"]["well_conditioned"] = bool(well)
                if well:
                    check(ero <= ROUNDED_REL and cro >= ROUNDED_COS, (n, which, "well-conditioned tensor vs matching rounded oracle", ero, cro))
                    check(e32 <= FP32_REL_CAP[mode], (n, which, "well-conditioned tensor vs fp32", e32))
                check(e32 <= VS_ORACLE_DISTANCE * e_ro32 + 0.02, (n, which, "vs fp32, against the matching oracle's own distance", e32, e_ro32))
                check(ero <= TWO_REALISATIONS * max(e_ro32, e32) + 0.02, (n, which, "vs matching oracle, against the noise floor", ero, e_ro32, e32))
        per_tensor.update(rows)
        vals = list(rows.values())
        by_block.append({"block": i, "max_rel_vs_rounded": max(v["rel_vs_rounded"] for v in vals), "min_cos_vs_rounded": min(v["cos_vs_rounded"] for v in vals),
                         "max_rel_vs_fp32": max(v["rel_vs_fp32"] for v in vals), "min_cos_vs_fp32": min(v["cos_vs_fp32"] for v in vals),
                         "max_rounded_oracle_rel_vs_fp32": max(v["rounded_oracle_rel_vs_fp32"] for v in vals),
                         "worst_tensor_vs_fp32": max(rows, key=lambda k: rows[k]["rel_vs_fp32"])})
    report["error_vs_depth"] = by_block
    wc = [v for v in per_tensor.values() if v["well_conditioned"]]
    report["worst"] = {"rel_vs_rounded": max(b["max_rel_vs_rounded"] for b in by_block), "cos_vs_rounded": min(b["min_cos_vs_rounded"] for b in by_block),
                       "rel_vs_fp32": max(b["max_rel_vs_fp32"] for b in by_block), "cos_vs_fp32": min(b["min_cos_vs_fp32"] for b in by_block),
                       "well_conditioned_tensors": len(wc), "tensors": len(per_tensor),
                       "well_conditioned_rel_vs_rounded": max((v["rel_vs_rounded"] for v in wc), default=None),
                       "well_conditioned_rel_vs_fp32": max((v["rel_vs_fp32"] for v in wc), default=None),
                       "hip_over_oracle_distance_from_fp32": max(v["rel_vs_fp32"] / max(v["rounded_oracle_rel_vs_fp32"], 1e-9) for v in per_tensor.values())}
    report["per_tensor"] = per_tensor
    report["failures"] = [repr(f) for f in fails]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = mode + ("" if LAYERS == 30 else f"_L{LAYERS}")
    with open(os.path.join(ROOT, "gpurun_out", f"cfg5_depth_parity_{tag}.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps({k: report[k] for k in ("hip", "fp32", "rounded", "worst", "hip_peak_gb")}))
    assert not fails, fails[:12]


# default: the BASELINE configs[4] arithmetic (fp8); the bf16 arm (79 s more of a whole GPU; profiles/r06_cfg5_depth_parity_bf16.json) with VGPA_GPU_FULL=1
@pytest.mark.parametrize("mode", ["bf16", "fp8"] if os.environ.get("VGPA_GPU_FULL", "0") == "1" else ["fp8"])
def test_cfg5_full_depth_pair_step_matches_the_oracle_block_by_block(mode):
    _run(mode)
