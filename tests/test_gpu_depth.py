"""FULL-DEPTH parity of the headline configuration: BASELINE.json configs[1] exactly as bench.py runs it (CogVideoX-5B T2V, 42 blocks, D = 3072, 48 heads,
paired latents [1,2,13,16,60,90] -> S = 17 776 tokens, LoRA r = 64 on to_q / to_k / to_v / to_out.0, bench.py's weight init and lora_B ~ N(0, 1e-3), (t, eps)
fixed) -- the HIP pair-step (train/CogVideoX-5B/03_train.py:116-157 through CogVideoXDPOTrainer._shared_step, every kernel through the C-ABI) against
oracle/cogvideox.py run ON THE GPU in fp32 at the same depth: per-block torch.utils.checkpoint (what the reference itself trains with, :107-108) and the
head-chunked exact attention keep the fp32 step inside 288 GB (tests/test_oracle_kat.py pins that form to the plain oracle).

VERDICT r5 "missing 2": until round 6 every oracle comparison stopped at 2 blocks; the number the bench reports had only the ln 2 identity at B = 0
behind it.  What is compared here, for all 42 blocks:
  loss / rewards       |d loss| <= 1e-3 (north_star) against the fp32 AND the activation-rounded oracle
  predictions          the four v-predictions (policy / reference x win / lose): max |d| over the prediction range, relative norm error
  LoRA gradients       EVERY one of the 336 tensors: relative error and cosine against (a) the activation-rounded oracle (round_activations=True,
                       exact_delta=True: fp32 arithmetic with the bf16-stored tensors and their gradients rounded where the HIP path rounds them) and
                       (b) the plain fp32 oracle, next to (c) the error of the oracle's own code in plain torch bf16 (what the reference's bf16-mixed
                       run computes) -- AS A FUNCTION OF BLOCK INDEX: profiles/r06_cfg2_depth_parity*.json holds the table (copied from gpurun_out/).
Bounds: stated next to the asserts below; every comparison is made and the report written before the first assert fires.

A second arm ("rich") repeats the comparison with AdaLN modulation weights of std 0.3 (gates / scales O(0.3), as tests/cfg1_common.py) so that all 42 blocks
contribute at order one to the residual stream: bench.py's N(0, 0.02) init gives gates of ~0.05 and lets early-block gradients travel mostly through the
identity path, which is the friendliest case for error growth with depth.  MEASURED (round 6, profiles/r06_cfg2_depth_parity_rich.json): such a random 42-block
net amplifies rounding noise by itself -- the activation-rounded ORACLE, fp32 arithmetic that merely rounds the same tensors, is 25-54 % from plain fp32 on the
gradients, torch bf16 30-380 % -- so absolute bounds against fp32 say nothing about kernels there.  What the arm asserts instead, on EVERY tensor: the HIP path is
no further from fp32 than 1.35 x the rounded oracle is (measured <= 1.21 x) and than 1.1 x torch bf16 is (measured <= 1.01 x; median 0.37 vs 0.57), and within
25 % / cosine 0.97 of the rounded oracle (two realisations of the same rounding noise through the same amplifier: measured 3-19 %).

-m gpu only; takes the whole GPU (HIP step 145 GB, then the oracle ~60 GB) and ~10 minutes per arm.
"""
import gc
import json
import math
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LAYERS = int(os.environ.get("VGPA_DEPTH_LAYERS", "42"))          # debug knob; anything but 42 is not the BASELINE config and is labelled so in the report
FRAMES, HEIGHT, WIDTH, TEXT_LEN, RANK = 13, 60, 90, 226, 64
TIMESTEP = 417

# bounds (EVERY tensor of every block)
LOSS_TOL = 1e-3                 # north_star
PRED_ERR_OVER_RANGE = 0.04      # bf16 activations through 42 blocks (cfg1, 2 blocks: 0.03)
ROUNDED_REL, ROUNDED_COS = 0.10, 0.995      # against the activation-rounded oracle: the cfg1 bounds, unchanged at 21 x the depth
FP32_REL_CAP, FLOOR_FACTOR, FP32_REL_FIXED = 0.12, 1.25, 0.08          # against fp32: tests/test_gpu_cfg1.py's rule, unchanged
RICH_ROUNDED_REL, RICH_ROUNDED_COS, RICH_VS_ROUNDED_ORACLE, RICH_VS_TORCH_BF16, RICH_LOSS_TOL_FP32 = 0.25, 0.97, 1.35, 1.10, 2e-3      # the "rich" arm (see the header)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _clean():
    gc.collect()
    torch.cuda.empty_cache()


def _build_model(arm):
    """bench.py::build_model (norm weights 1, biases 0, everything else N(0, 0.02), seed 0); arm "rich": the AdaLN modulation linears N(0, 0.3) + biases"""
    from videogpa_amd.transformer import COGVIDEOX_5B, CogVideoXTransformer3DModel
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("cuda"):
            model = CogVideoXTransformer3DModel(**dict(COGVIDEOX_5B, num_layers=LAYERS))
    finally:
        torch.set_default_dtype(prev)
    g = torch.Generator(device="cuda").manual_seed(0)
    with torch.no_grad():
        for name, p in model.named_parameters():
            mod_lin = (".norm1.linear." in name or ".norm2.linear." in name or name.startswith("norm_out.linear."))
            if name.endswith("norm.weight") or name in ("norm_final.weight",) or ".norm_q.weight" in name or ".norm_k.weight" in name:
                p.fill_(1.0)
            elif arm == "rich" and mod_lin:
                p.normal_(0.0, 0.3, generator=g)
            elif name.endswith(".bias"):
                p.zero_()
            else:
                p.normal_(0.0, 0.02, generator=g)
    return model


def _inputs():
    g = torch.Generator(device="cuda").manual_seed(1234)          # bench.py: seed 1234 + rank
    x_pair = (0.7 * torch.randn(1, 2, FRAMES, 16, HEIGHT, WIDTH, generator=g, device="cuda")).to(torch.bfloat16)
    prompt = (0.2 * torch.randn(1, TEXT_LEN, 4096, generator=g, device="cuda")).to(torch.bfloat16)
    noise = torch.randn(1, FRAMES, 16, HEIGHT, WIDTH, generator=g, device="cuda").to(torch.bfloat16)
    t = torch.tensor([TIMESTEP], device="cuda")
    return x_pair, prompt, t, noise


def _hip_step(arm):
    """-> (scalars, preds {name: bf16 tensor on the GPU}, grads {PEFT key: fp32 on the GPU}, base state dict bf16, adapter dict fp32, peak GB, seconds)"""
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    model = _build_model(arm)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    tr = CogVideoXDPOTrainer({"lora_rank": RANK, "lora_alpha": 2 * RANK, "beta": 1.0, "accumulate_grad_batches": 1, "lean_activations": False, "seed": 1234},
                             transformer=model)
    gB = torch.Generator(device="cuda").manual_seed(1)
    lora = {}
    with torch.no_grad():
        for n, p in tr.transformer.named_parameters():
            if ".lora_B." in n:
                p.normal_(0.0, 1e-3, generator=gB)              # bench.py
            if ".lora_" in n:
                p.copy_(p.bfloat16().float())                   # bf16-representable adapters: both sides multiply the same numbers (as tests/cfg1_common.py)
                lora[n.replace(".default.weight", ".weight")] = p.detach().clone()
    tr.train()
    x_pair, prompt, t, noise = _inputs()
    captured = []
    orig = tr.transformer.forward

    def spy(*a, **k):
        out = orig(*a, **k)
        captured.append(out.sample.detach())
        return out
    tr.transformer.forward = spy
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    out = tr._shared_step({"x_pair": x_pair, "prompt_emb": prompt}, timesteps=t, noise=noise)
    tr.transformer.forward = orig
    out.loss.backward()
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    v_ref, v_pol = captured                  # reference pass first (adapter off), then the policy pass; batch = (win, lose)
    preds = {"v_win": v_pol[0:1].clone(), "v_lose": v_pol[1:2].clone(), "v_win_ref": v_ref[0:1].clone(), "v_lose_ref": v_ref[1:2].clone()}
    grads = {n.replace(".default.weight", ".weight"): p.grad.detach().float().clone() for n, p in tr.transformer.named_parameters() if ".lora_" in n}
    scal = {"loss": out.loss.item(), "reward_margin": out.reward_margin.item(), "winner_reward": out.winner_reward.item(),
            "loser_reward": out.loser_reward.item()}
    del out, tr, model, captured, v_ref, v_pol
    _clean()
    return scal, preds, grads, sd, lora, peak, secs


def _oracle(sd_bf16, lora_f32, dtype, **kw):
    """oracle/cogvideox.py::dpo_pair_step on the GPU at full depth -> (scalars, preds, grads fp32, seconds)"""
    from oracle import cogvideox as ocv
    from oracle import scheduler as osch
    cfg = ocv.CogVideoXConfig(num_layers=LAYERS)
    sd = {k: v.to(dtype) for k, v in sd_bf16.items()}
    lora = {k: v.clone().requires_grad_(True) for k, v in lora_f32.items()}
    x_pair, prompt, t, noise = _inputs()
    xw, xl = (x_pair[:, i].permute(0, 2, 1, 3, 4).to(dtype) for i in (0, 1))       # the dataset's [B,C,F,H,W] (train/dataset.py:228-229)
    abar = osch.alphas_cumprod().cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = ocv.dpo_pair_step(sd, cfg, lora, abar, xw, xl, prompt.to(dtype), t, noise.to(dtype), beta=1.0, checkpoint_blocks=True,
                            chunked_attention=(dtype == torch.float32), **kw)
    out["loss"].backward()
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    scal = {"loss": float(out["loss"].detach()), "reward_margin": float(out["reward_margin"]), "winner_reward": float(out["winner_reward"]),
            "loser_reward": float(out["loser_reward"])}
    preds = {k: out[k].detach().float() for k in ("v_win", "v_lose", "v_win_ref", "v_lose_ref")}
    grads = {k: p.grad.detach().float() for k, p in lora.items()}
    del out, sd, lora
    _clean()
    return scal, preds, grads, secs


def _rel(a, r):
    return float((a.double() - r.double()).norm() / r.double().norm().clamp_min(1e-300))


def _cos(a, r):
    a, r = a.double().flatten(), r.double().flatten()
    return float((a * r).sum() / (a.norm() * r.norm()).clamp_min(1e-300))


def _short(k):
    return k.replace("base_model.model.transformer_blocks.", "")


def _run(arm):
    report = {"config": f"BASELINE configs[1] as bench.py runs it: {LAYERS} blocks, S = {TEXT_LEN + FRAMES * (HEIGHT // 2) * (WIDTH // 2)}, r = {RANK}, "
                        f"lora_B ~ N(0, 1e-3), t = {TIMESTEP}; weights: " + ("bench.py init (N(0, 0.02), biases 0)" if arm == "bench" else
                                                                             "bench.py init with AdaLN modulation linears N(0, 0.3) (gates / scales of order 0.3)"),
              "is_baseline_depth": LAYERS == 42, "arm": arm,
              "bounds": ({"loss": LOSS_TOL, "pred_err_over_range": PRED_ERR_OVER_RANGE, "rounded_rel": ROUNDED_REL, "rounded_cos": ROUNDED_COS,
                          "fp32_rule": f"min(max({FP32_REL_FIXED}, {FLOOR_FACTOR} x torch-bf16 error), {FP32_REL_CAP})"} if arm == "bench" else
                         {"loss_vs_rounded": LOSS_TOL, "loss_vs_fp32": RICH_LOSS_TOL_FP32, "pred_err_over_range": PRED_ERR_OVER_RANGE, "rounded_rel": RICH_ROUNDED_REL,
                          "rounded_cos": RICH_ROUNDED_COS, "fp32_rule": f"<= {RICH_VS_ROUNDED_ORACLE} x (rounded oracle vs fp32) + 0.02 and <= {RICH_VS_TORCH_BF16} x (torch bf16 vs fp32) + 0.02"})}
    fails = []

    def check(ok, what):
        if not ok:
            fails.append(what)

    hip, hip_preds, hip_grads, sd, lora, peak, secs = _hip_step(arm)
    report.update(hip=hip, hip_peak_gb=peak, hip_step_seconds_incl_first_call_setup=secs)
    assert len(hip_grads) == LAYERS * 8

    oracles = {}
    for name, dtype, kw in (("fp32", torch.float32, {}), ("rounded", torch.float32, {"round_activations": True, "exact_delta": True}),
                            ("torch_bf16", torch.bfloat16, {})):
        oracles[name] = _oracle(sd, lora, dtype, **kw)
        report[name] = dict(oracles[name][0], seconds=oracles[name][3])
    del sd

    for name in ("fp32", "rounded"):
        d = abs(hip["loss"] - oracles[name][0]["loss"])
        report[f"loss_abs_err_vs_{name}"] = d
        check(d < (RICH_LOSS_TOL_FP32 if (arm == "rich" and name == "fp32") else LOSS_TOL), ("loss", name, hip["loss"], oracles[name][0]["loss"]))
        for k in ("winner_reward", "loser_reward"):
            ref = oracles[name][0][k]
            check(abs(hip[k] - ref) < 2e-4 + 0.01 * abs(ref), (k, name, hip[k], ref))
        check(abs(hip["reward_margin"] - oracles[name][0]["reward_margin"]) < 1e-3, ("reward_margin", name, hip["reward_margin"], oracles[name][0]["reward_margin"]))
    report["loss_abs_err_torch_bf16_vs_fp32"] = abs(oracles["torch_bf16"][0]["loss"] - oracles["fp32"][0]["loss"])

    report["predictions"] = {}
    for k, v in hip_preds.items():
        row = {}
        for name in ("fp32", "rounded", "torch_bf16"):
            ref = oracles[name][1][k]
            rng_ = float(ref.abs().max())
            row[f"max_err_over_range_vs_{name}"] = float((v.float() - ref).abs().max()) / rng_
            row[f"rel_norm_err_vs_{name}"] = _rel(v.float(), ref)
        row["torch_bf16_max_err_over_range_vs_fp32"] = float((oracles["torch_bf16"][1][k] - oracles["fp32"][1][k]).abs().max()) / float(oracles["fp32"][1][k].abs().max())
        report["predictions"][k] = row
        check(row["max_err_over_range_vs_fp32"] < PRED_ERR_OVER_RANGE, (k, "prediction vs fp32", row["max_err_over_range_vs_fp32"]))
        check(row["max_err_over_range_vs_rounded"] < PRED_ERR_OVER_RANGE, (k, "prediction vs rounded", row["max_err_over_range_vs_rounded"]))

    g32, gro, gbf = oracles["fp32"][2], oracles["rounded"][2], oracles["torch_bf16"][2]
    assert set(hip_grads) == set(g32) == set(gro)
    per_tensor, by_block = {}, []
    for i in range(LAYERS):
        rows = {}
        for k in sorted(k for k in hip_grads if f".transformer_blocks.{i}.attn1." in k):
            g = hip_grads[k]
            e32, c32 = _rel(g, g32[k]), _cos(g, g32[k])
            ero, cro = _rel(g, gro[k]), _cos(g, gro[k])
            efl = _rel(gbf[k], g32[k])
            e_ro32 = _rel(gro[k], g32[k])
            bound = min(max(FP32_REL_FIXED, FLOOR_FACTOR * efl), FP32_REL_CAP)
            rows[k] = {"rel_vs_fp32": round(e32, 5), "cos_vs_fp32": round(c32, 6), "rel_vs_rounded": round(ero, 5), "cos_vs_rounded": round(cro, 6),
                       "torch_bf16_rel_vs_fp32": round(efl, 5), "rounded_oracle_rel_vs_fp32": round(e_ro32, 5), "bound_vs_fp32": round(bound, 5),
                       "norm_fp32": float(g32[k].double().norm())}
            if arm == "bench":
                check(ero <= ROUNDED_REL and cro >= ROUNDED_COS, (_short(k), "vs activation-rounded oracle", ero, cro))
                check(e32 <= bound, (_short(k), "vs fp32", e32, bound))
            else:       # the noise-amplifying arm: relative to what fp32 arithmetic with the same roundings, and torch bf16, achieve on the same tensor
                check(ero <= RICH_ROUNDED_REL and cro >= RICH_ROUNDED_COS, (_short(k), "vs activation-rounded oracle", ero, cro))
                check(e32 <= RICH_VS_ROUNDED_ORACLE * e_ro32 + 0.02, (_short(k), "vs fp32, against the rounded oracle's own distance", e32, e_ro32))
                check(e32 <= RICH_VS_TORCH_BF16 * efl + 0.02, (_short(k), "vs fp32, against torch bf16's distance", e32, efl))
        per_tensor.update({_short(k): v for k, v in rows.items()})
        vals = list(rows.values())
        by_block.append({"block": i,
                         "max_rel_vs_rounded": max(v["rel_vs_rounded"] for v in vals), "min_cos_vs_rounded": min(v["cos_vs_rounded"] for v in vals),
                         "max_rel_vs_fp32": max(v["rel_vs_fp32"] for v in vals), "min_cos_vs_fp32": min(v["cos_vs_fp32"] for v in vals),
                         "max_torch_bf16_rel_vs_fp32": max(v["torch_bf16_rel_vs_fp32"] for v in vals),
                         "max_rounded_oracle_rel_vs_fp32": max(v["rounded_oracle_rel_vs_fp32"] for v in vals),
                         "worst_tensor_vs_fp32": _short(max(rows, key=lambda k: rows[k]["rel_vs_fp32"]))})
    report["error_vs_depth"] = by_block
    report["worst"] = {"rel_vs_rounded": max(b["max_rel_vs_rounded"] for b in by_block), "cos_vs_rounded": min(b["min_cos_vs_rounded"] for b in by_block),
                       "rel_vs_fp32": max(b["max_rel_vs_fp32"] for b in by_block), "cos_vs_fp32": min(b["min_cos_vs_fp32"] for b in by_block),
                       "torch_bf16_rel_vs_fp32": max(b["max_torch_bf16_rel_vs_fp32"] for b in by_block)}
    report["per_tensor"] = per_tensor
    report["failures"] = [repr(f) for f in fails]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = ("" if arm == "bench" else "_" + arm) + ("" if LAYERS == 42 else f"_L{LAYERS}")
    with open(os.path.join(ROOT, "gpurun_out", f"cfg2_depth_parity{tag}.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps({k: report[k] for k in ("hip", "fp32", "rounded", "torch_bf16", "worst", "hip_peak_gb")}))
    assert not fails, fails[:12]


def test_cfg2_full_depth_pair_step_matches_the_fp32_oracle_block_by_block():
    _run("bench")


@pytest.mark.skipif(os.environ.get("VGPA_DEPTH_RICH", "0") != "1", reason="opt-in (VGPA_DEPTH_RICH=1): 2.2 more minutes of a whole GPU; measured in round 6, "
                    "profiles/r06_cfg2_depth_parity_rich.json")
def test_cfg2_full_depth_with_order_one_gates_matches_the_fp32_oracle_block_by_block():
    _run("rich")
