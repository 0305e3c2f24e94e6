"""Full-width parity at BASELINE.json configs[0] geometry: D = 3072, 48 heads, 2 transformer blocks, 13f x 64 x 64 paired
latents (S = 13 538 tokens), LoRA r = 8 (and r = 64: the rp = 64 / 192 kernels of the headline config).  The HIP
pair-step (train/CogVideoX-5B/03_train.py:116-157 through CogVideoXDPOTrainer._shared_step, every kernel through the
C-ABI) runs on the seeded inputs of tests/cfg1_common.py and is compared with the fp32 oracle: the committed CPU results
tests/golden/cfg1_<variant>.pt (tests/golden/make_cfg1_golden.py) for the loss, the rewards and the predictions, and -- because the
golden holds only 256 samples and the norm of each gradient -- the SAME oracle code run in fp32 in this process for the full gradient
tensors (first checked against the golden's samples: cosine >= 0.99999, norm within 1e-4).

Tolerances (the HIP path computes in bf16, the oracle in fp32 on the same bf16-rounded weights / inputs):
  loss                  |d| <= 1e-3 (north_star) -- measured ~3e-4
  rewards (= -mean err) |d| <= 2e-4 + 1 % relative; reward_margin (their difference, 1e-3 of the rewards) |d| <= 1e-3
  v_pred samples        |d| <= 3 % of the prediction range (bf16 activations through 2 blocks)
  LoRA grads, EVERY tensor, both variants, against TWO references:
    (a) the ACTIVATION-ROUNDED oracle (oracle/cogvideox.py round_activations=True): fp32 arithmetic with every tensor the HIP path stores in bf16
        rounded to bf16 in the forward and its gradient rounded in the backward -- the same arithmetic TYPE at the same places, so what is left is
        arithmetic ORDER and the independent realisation of the rounding noise.  Bound: relative error <= 10 % AND cosine >= 0.995 on every tensor,
        the last block's to_q / to_k included (measured 1.2-6 %; two runs of the rounded oracle itself that differ by ONE bf16 ulp in ONE input
        element drift 0.3-2.8 % apart: tools/cfg1_round_diag.py, profiles/r04_cfg1_round_diag_*.json).
    (b) the fp32 oracle: e_hip <= min(max(0.08, 1.25 x e_torch_bf16), 0.12), where e_torch_bf16 is the error of the oracle's own code run in plain
        torch bf16 on the same GPU (what the reference's bf16-mixed training computes; 0.04-0.86 per tensor).  The 12 % cap holds for EVERY tensor
        since round 4 (measured 1.7-10.4 %; the activation-rounded oracle itself is 1.8-9.7 % from fp32 on the same tensors).
  What round 4 found with (a): the 37-87 % the last block's to_q / to_k adapters were off (in this path AND in torch bf16) is not diffuse
  "noise amplified by cancellation" but ONE mechanism: the flash-attention backward forms delta = rowsum(dO o O) from the STORED bf16 output, while
  the identity delta = rowsum(P o dP) holds for the unrounded O = P V only; each row's dS then stops summing to zero and dQ_i picks up
  -d(delta_i) sum_j P_ij K_j, a coherent term.  The rounded oracle reproduces the HIP gradients to 4-9 % once it does the same
  (test_cfg1_plain_delta_path_matches_the_oracle_that_rounds_the_output_in_delta), and is within 3-4 % of fp32 when delta comes from the
  unrounded output.  The HIP path therefore stores what the output's bf16 rounding dropped next to it (ops.py "Precise delta": eight further mantissa
  bits per element since round 5, csrc/common.h res8, + S D bytes per layer, kept under lean activations too; round 4's bf16 residual tensor stays
  selectable) and forms delta from the completed output: reference (a) is the oracle with exact_delta=True, and the fp32 errors of that family drop from
  0.37-0.87 to the level of every other tensor.  test_cfg1_pair_step_matches_oracle_golden runs all three modes (MODES) against the same bounds.
  attention backward, kernel level (test_cfg1_attention_backward_matches_rounding_injected_recompute): every attention-backward
                        launch of the step is recorded and dQ / dK / dV of six (batch, head) slices each are recomputed in fp32 from
                        the SAME inputs with only the two roundings every bf16 flash attention makes (P -> bf16 for dV, dS -> bf16 for
                        dQ / dK): cosine >= 0.999, norm within 1 % -- the tight statement about the kernels themselves.
"""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import cfg1_common as c1
from oracle import cogvideox as ocv

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


REL_FIXED, FLOOR_FACTOR, ABS_CAP = 0.08, 1.25, 0.12    # against fp32: never further than 1.25 x torch bf16, within 8 % where that is achievable, within 12 % EVERYWHERE
ROUNDED_REL, ROUNDED_COS = 0.10, 0.995          # against the activation-rounded oracle


_ORACLE_CACHE = {}


def _oracle_on_gpu(variant, dtype, round_p_ds=False, **kw):
    """cached per (variant, mode): the parametrised tests below share the four oracle runs of a variant (results live on the CPU)"""
    key = (variant, dtype, round_p_ds, tuple(sorted(kw.items())))
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = _oracle_on_gpu_uncached(variant, dtype, round_p_ds, **kw)
    return _ORACLE_CACHE[key]


def _oracle_on_gpu_uncached(variant, dtype, round_p_ds=False, qk_gain=None, **kw):
    """The oracle's own code on the GPU: fp32 (the reference for whole gradient tensors), fp32 with the P / dS roundings injected, fp32 with EVERY
    bf16-stored tensor rounded (round_activations=True; exact_delta=True forms the attention backward's delta from the unrounded output), or
    bf16 weights / activations (plain torch bf16: the floor)  -> (loss, {name: grad fp32 on the CPU})."""
    from oracle import scheduler as osch
    cfg = c1.config()
    sd = {k: v.to(dtype).cuda() for k, v in (c1.base_state_dict(cfg) if qk_gain is None else c1.trained_like_state_dict(cfg, qk_gain)).items()}
    lora, _ = c1.lora_state_dict(cfg, variant)
    lora = {k: v.cuda().requires_grad_(True) for k, v in lora.items()}
    xw, xl, prompt, t, noise = (v.to(dtype) if v.is_floating_point() else v for v in c1.inputs())
    abar = osch.alphas_cumprod().cuda()
    out = ocv.dpo_pair_step(sd, cfg, lora, abar, xw.cuda(), xl.cuda(), prompt.cuda(), t.cuda(), noise.cuda(), beta=1.0, round_p_ds=round_p_ds, **kw)
    out["loss"].backward()
    loss = float(out["loss"].detach())
    grads = {k: p.grad.float().cpu() for k, p in lora.items()}
    del out, sd, lora
    torch.cuda.empty_cache()
    return loss, grads


def _hip_step(variant, precise_delta="int8", lean=False, qk_gain=None):
    from videogpa_amd.lora import LoraConfig, get_peft_model
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    from videogpa_amd.transformer import COGVIDEOX_5B, CogVideoXTransformer3DModel
    cfg = c1.config()
    sd = c1.base_state_dict(cfg) if qk_gain is None else c1.trained_like_state_dict(cfg, qk_gain)
    model = CogVideoXTransformer3DModel(**dict(COGVIDEOX_5B, num_layers=cfg.num_layers, sample_height=c1.HEIGHT, sample_width=c1.WIDTH))
    model.load_state_dict(sd, strict=True)
    del sd
    model = model.to(device="cuda", dtype=torch.bfloat16)
    model.set_precise_delta(precise_delta)         # per-model setting (ops.py "Precise delta"): "int8" (the default), "bf16" or None
    model.enable_lean_activations(lean)
    lora, r = c1.lora_state_dict(cfg, variant)
    pm = get_peft_model(model, LoraConfig(r=r, lora_alpha=2 * r, target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    own = pm.state_dict()
    for k, v in lora.items():
        own[k[:-len(".weight")] + ".default.weight"].copy_(v)
    tr = CogVideoXDPOTrainer({"beta": 1.0, "lean_activations": bool(lean)}, transformer=pm)
    tr.train()
    x_win, x_lose, prompt, t, noise = c1.inputs()
    captured = {}
    orig = tr.transformer.forward

    def spy(*a, **k):
        out = orig(*a, **k)
        captured.setdefault("preds", []).append(out.sample.detach())
        return out
    tr.transformer.forward = spy
    out = tr._shared_step({"x_win": x_win.cuda(), "x_lose": x_lose.cuda(), "prompt_emb": prompt.cuda()}, timesteps=t.cuda(), noise=noise.cuda())
    tr.transformer.forward = orig
    out.loss.backward()
    torch.cuda.synchronize()
    v_ref, v_pol = captured["preds"]            # reference pass first (adapter off), then the policy pass; batch = (win, lose)
    grads = {}
    named = dict(pm.named_parameters())
    for k in lora:
        grads[k] = named[k[:-len(".weight")] + ".default.weight"].grad.detach().float().cpu()
    return out, {"v_win": v_pol[0:1], "v_lose": v_pol[1:2], "v_win_ref": v_ref[0:1], "v_lose_ref": v_ref[1:2]}, grads


# (precise_delta, lean_activations): the default; the default under lean activations (cfg4's memory policy, and cfg3 at the reference's batch 2); the round-4
# bf16 residual.  All three must meet the SAME bounds -- gradient accuracy does not depend on a memory switch (VERDICT r4 weak 1 / ADVICE r4).
MODES = {"int8": ("int8", False), "int8_lean": ("int8", True), "bf16res": ("bf16", False)}


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("variant", ["r8", "r64"])
def test_cfg1_pair_step_matches_oracle_golden(variant, mode):
    gold = torch.load(os.path.join(HERE, "golden", f"cfg1_{variant}.pt"), weights_only=False)
    out, preds, grads = _hip_step(variant, *MODES[mode])
    report = {"variant": variant, "mode": mode, "precise_delta": MODES[mode][0], "lean_activations": MODES[mode][1],
              "loss_hip": out.loss.item(), "loss_oracle": float(gold["loss"])}
    fails = []      # every comparison is made and written to gpurun_out/ before the first assert fires

    def check(ok, what):
        if not ok:
            fails.append(what)

    check(abs(out.loss.item() - float(gold["loss"])) < 1e-3, ("loss", out.loss.item(), float(gold["loss"])))
    for name, got in (("reward_margin", out.reward_margin), ("winner_reward", out.winner_reward), ("loser_reward", out.loser_reward)):
        ref = float(gold[name])
        report[name] = (got.item(), ref)
        tol = 1e-3 if name == "reward_margin" else 2e-4 + 0.01 * abs(ref)
        check(abs(got.item() - ref) < tol, (name, got.item(), ref))

    for k, v in preds.items():
        v = v.float().cpu()
        ref = gold[k + "_samples"].float()
        idx = c1.sample_index(v.numel(), k)
        got = v.flatten()[idx]
        rng_ = ref.abs().max().item()
        err = (got - ref).abs().max().item()
        report[k + "_err_over_range"] = err / rng_
        report[k + "_norm_rel"] = abs(v.double().norm().item() / float(gold[k + "_norm"]) - 1)
        check(err < 0.03 * rng_, (k, err, rng_))
        check(report[k + "_norm_rel"] < 0.01, (k, report[k + "_norm_rel"]))

    del preds
    torch.cuda.empty_cache()
    ref_loss, ref_grads = _oracle_on_gpu(variant, torch.float32)
    floor_loss, floor_grads = _oracle_on_gpu(variant, torch.bfloat16)
    ro_loss, ro_grads = _oracle_on_gpu(variant, torch.float32, round_activations=True, exact_delta=True)
    report.update(loss_oracle_fp32_here=ref_loss, loss_torch_bf16=floor_loss, loss_activation_rounded_oracle=ro_loss)
    check(abs(ref_loss - float(gold["loss"])) < 1e-5, ("in-process fp32 oracle vs golden loss", ref_loss, float(gold["loss"])))
    check(abs(ro_loss - float(gold["loss"])) < 5e-4, ("activation-rounded oracle drifted from the golden", ro_loss, float(gold["loss"])))
    check(abs(out.loss.item() - ro_loss) < 5e-4, ("loss vs the activation-rounded oracle", out.loss.item(), ro_loss))
    worst = {"rel_err_hip": 0.0, "rel_err_over_bound": 0.0}
    per_tensor = {}
    assert set(grads) == set(gold["lora_grads"]) == set(ref_grads)

    def rel(a, r):
        return float((a.double() - r.double()).norm() / r.double().norm())

    def cosine(a, r):
        a, r = a.double().flatten(), r.double().flatten()
        return float((a * r).sum() / (a.norm() * r.norm()).clamp_min(1e-300))

    for k, g in grads.items():
        r = ref_grads[k]
        # the in-process fp32 oracle IS the golden: same samples, same norm
        gs = gold["lora_grads"][k]
        idx = c1.sample_index(g.numel(), k)
        check(cosine(r.flatten()[idx], gs["samples"]) > 0.99999 and abs(r.double().norm().item() / float(gs["norm"]) - 1) < 1e-4,
              (k, "in-process fp32 oracle vs golden samples", cosine(r.flatten()[idx], gs["samples"])))
        e_hip, e_floor = rel(g, r), rel(floor_grads[k], r)
        bound = min(max(REL_FIXED, FLOOR_FACTOR * e_floor), ABS_CAP)
        e_ro, c_ro = rel(g, ro_grads[k]), cosine(g, ro_grads[k])
        per_tensor[k.replace("base_model.model.transformer_blocks.", "")] = {
            "rel_err_hip": round(e_hip, 5), "rel_err_torch_bf16": round(e_floor, 5), "bound": round(bound, 5), "cos_hip": round(cosine(g, r), 6),
            "cos_torch_bf16": round(cosine(floor_grads[k], r), 6), "rel_err_hip_vs_activation_rounded_oracle": round(e_ro, 5),
            "cos_hip_vs_activation_rounded_oracle": round(c_ro, 6), "rel_err_activation_rounded_oracle_vs_fp32": round(rel(ro_grads[k], r), 5)}
        worst["rel_err_hip"] = max(worst["rel_err_hip"], e_hip)
        worst["rel_err_over_bound"] = max(worst["rel_err_over_bound"], e_hip / bound)
        worst["rel_err_vs_rounded_oracle"] = max(worst.get("rel_err_vs_rounded_oracle", 0.0), e_ro)
        worst["cos_vs_rounded_oracle"] = min(worst.get("cos_vs_rounded_oracle", 1.0), c_ro)
        check(math.isfinite(e_hip) and e_hip <= bound, (k, "relative error vs fp32 oracle", e_hip, "bound", bound, "torch bf16", e_floor))
        check(e_ro <= ROUNDED_REL and c_ro >= ROUNDED_COS, (k, "vs the activation-rounded oracle: relative error", e_ro, "cosine", c_ro))
    report["lora_grads_worst"] = worst
    report["lora_grads_per_tensor"] = per_tensor
    report["failed_checks"] = [str(f) for f in fails]
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(HERE), "gpurun_out", f"cfg1_parity_{variant}" + ("" if mode == "int8" else "_" + mode) + ".json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
    assert not fails, fails


def test_cfg1_pair_step_on_trained_like_qk_norm_gains_matches_the_oracles():
    """Round 6: the same full-width pair step (D = 3072, 48 heads, 2 blocks, S = 13 538, r = 64) with the QK-norm affines of a TRAINED model's shape
    (c1.trained_like_state_dict: gains 2.5 +- 20 %, three 3 x outlier channels, biases) -- sharp attention rows, scores spread over +-100 log2 units: the data on
    which the forward's shift follows the sampled maximum instead of the bound, weights far above 1 pass through the matrix-pipe row sums, and strips can be flagged.
    Every other full-width parity test runs unit gains (nearly flat rows).  Against the fp32 oracle, the activation-rounded oracle and plain torch bf16, all run here on
    the GPU.  On such rows bf16 ARITHMETIC is an order noisier than on flat ones -- a score of +-100 log2 units carries the 2^-9 roundings of q and k as ~0.03 units,
    2 % of a weight -- so the references themselves sit 7-12 % (activation-rounded oracle) and 8-13 % (torch bf16) from fp32 on EVERY LoRA tensor and 1.1e-3 / 1.3e-2
    in the loss (measured: profiles/r06_cfg1_parity_trained_like.json).  Asserted in that relative form: every HIP tensor no further from fp32 than 1.2 x the worse of
    the two references (measured 0.91-1.07 x) and 15 % absolutely; from the rounded oracle no further than 1.5 x that oracle's own distance from fp32 (two realisations
    of one noise: measured <= 1.17 x), cosine >= 0.99; loss within max(1e-3, 4 x the rounded oracle's own error) of fp32.  The same step on the all-online forward
    (fp32 row sums, running maximum) is 8-12 % from fp32 too and 3-5 % from this one (tools/cfg1_trained_like_diag.py): the noise is not the shift's or the row sums'."""
    variant, gain = "r64", 2.5
    out, preds, grads = _hip_step(variant, qk_gain=gain)
    ref_loss, ref_grads = _oracle_on_gpu(variant, torch.float32, qk_gain=gain)
    floor_loss, floor_grads = _oracle_on_gpu(variant, torch.bfloat16, qk_gain=gain)
    ro_loss, ro_grads = _oracle_on_gpu(variant, torch.float32, qk_gain=gain, round_activations=True, exact_delta=True)

    def rel(a, r):
        return float((a.double() - r.double()).norm() / r.double().norm())

    def cosine(a, r):
        a, r = a.double().flatten(), r.double().flatten()
        return float((a * r).sum() / (a.norm() * r.norm()).clamp_min(1e-300))
    report = {"qk_gain": gain, "variant": variant, "loss_hip": out.loss.item(), "loss_fp32": ref_loss, "loss_rounded": ro_loss, "loss_torch_bf16": floor_loss, "per_tensor": {}}
    fails = []
    if not abs(out.loss.item() - ref_loss) <= max(1e-3, 4.0 * abs(ro_loss - ref_loss)):
        fails.append(("loss vs fp32", out.loss.item(), ref_loss, "rounded oracle", ro_loss, "torch bf16", floor_loss))
    worst = {"rel_vs_fp32": 0.0, "rel_vs_rounded": 0.0, "cos_vs_rounded": 1.0, "torch_bf16_rel_vs_fp32": 0.0, "rounded_rel_vs_fp32": 0.0}
    for k, g in grads.items():
        e_hip, e_floor, e_ro, c_ro, e_rr = rel(g, ref_grads[k]), rel(floor_grads[k], ref_grads[k]), rel(g, ro_grads[k]), cosine(g, ro_grads[k]), rel(ro_grads[k], ref_grads[k])
        bound = min(max(REL_FIXED, 1.2 * max(e_floor, e_rr)), 0.15)
        report["per_tensor"][k.replace("base_model.model.transformer_blocks.", "")] = {"rel_vs_fp32": round(e_hip, 5), "torch_bf16_rel_vs_fp32": round(e_floor, 5),
                                                                                         "rel_vs_rounded": round(e_ro, 5), "cos_vs_rounded": round(c_ro, 6), "rounded_rel_vs_fp32": round(e_rr, 5)}
        worst["rel_vs_fp32"] = max(worst["rel_vs_fp32"], e_hip)
        worst["rel_vs_rounded"] = max(worst["rel_vs_rounded"], e_ro)
        worst["cos_vs_rounded"] = min(worst["cos_vs_rounded"], c_ro)
        worst["torch_bf16_rel_vs_fp32"] = max(worst["torch_bf16_rel_vs_fp32"], e_floor)
        worst["rounded_rel_vs_fp32"] = max(worst["rounded_rel_vs_fp32"], e_rr)
        if not (math.isfinite(e_hip) and e_hip <= bound):
            fails.append((k, "vs fp32", e_hip, "bound", bound, "torch bf16", e_floor))
        if not (e_ro <= max(ROUNDED_REL, 1.5 * e_rr) and c_ro >= 0.99):
            fails.append((k, "vs rounded oracle", e_ro, c_ro, "that oracle vs fp32", e_rr))
    report["worst"] = worst
    report["failed_checks"] = [str(f) for f in fails]
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(HERE), "gpurun_out", "cfg1_parity_trained_like.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps({k: v for k, v in report.items() if k != "per_tensor"}))
    assert not fails, fails


@pytest.mark.parametrize("variant", ["r8", "r64"])
def test_cfg1_plain_delta_path_matches_the_oracle_that_rounds_the_output_in_delta(variant):
    """precise_delta None = the textbook flash-attention backward (delta from the stored bf16 output; what torch's bf16 kernels do): the HIP gradients
    must then follow the activation-rounded oracle that ALSO forms delta from the rounded output -- every tensor within 12 % / cosine 0.993 (measured
    1.3-9.2 %, cosine >= 0.9958; the last block's to_q / to_k are 35-83 % away from fp32 in BOTH, which is the point: the mechanism is modelled, not
    tolerated) -- and the default path must be at least 3x closer to fp32 than this one on the tensors it was introduced for."""
    _, _, g_plain = _hip_step(variant, precise_delta=None)
    _, _, g_prec = _hip_step(variant, precise_delta="int8")
    _, ro = _oracle_on_gpu(variant, torch.float32, round_activations=True, exact_delta=False)
    _, ref = _oracle_on_gpu(variant, torch.float32)

    def rel(a, r):
        return float((a.double() - r.double()).norm() / r.double().norm())

    worst, gains = 0.0, {}
    for k, g in g_plain.items():
        a, r = g.double().flatten(), ro[k].double().flatten()
        cos = float((a * r).sum() / (a.norm() * r.norm()))
        worst = max(worst, rel(g, ro[k]))
        assert rel(g, ro[k]) <= 0.12 and cos >= 0.993, (k, rel(g, ro[k]), cos)
        if ".transformer_blocks.1.attn1.to_q." in k or ".transformer_blocks.1.attn1.to_k." in k:
            gains[k] = (rel(g, ref[k]), rel(g_prec[k], ref[k]))
            assert gains[k][1] < gains[k][0] / 3 and gains[k][1] < 0.12, (k, gains[k])
    print(json.dumps({"variant": variant, "plain_delta_worst_rel_err_vs_matching_oracle": worst,
                      "last_block_qk_rel_err_vs_fp32_plain_then_precise": {k.split("blocks.")[1]: v for k, v in gains.items()}}))


def test_cfg1_attention_backward_matches_rounding_injected_recompute():
    """Kernel-level, at full width: every attention-backward launch of the cfg1 HIP step (r64) is recorded and, for six (batch, head)
    slices each, dQ / dK / dV are recomputed in fp32 torch from the SAME bf16 inputs with only the kernels' two roundings (P, dS to
    bf16) injected.  What is left is summation order and the exp2 / lse path: cosine >= 0.999, norm within 1 %."""
    from videogpa_amd import ops
    rec = []
    orig = ops.attention_bwd_raw

    def spy(q, k, v, o, do, lse, dq, dk, dv, **kw):
        orig(q, k, v, o, do, lse, dq, dk, dv, **kw)
        r = kw.get("o_res")      # what the kernels' delta is formed from: the output completed by the forward's residual tensor (ops.py "Precise delta")
        o_full = o.float() if r is None else (o.float() + r.float() if r.dtype == torch.bfloat16 else ops.res8_decode(o, r))
        rec.append(tuple(t.clone() for t in (q, k, v, o_full, do, dq, dk, dv)))
    ops.attention_bwd_raw = spy
    try:
        _hip_step("r64")
    finally:
        ops.attention_bwd_raw = orig
    assert len(rec) == 2                      # one launch per block (win and lose batched)
    log2e, scale = 1.4426950408889634, 0.125
    worst = {"cos": 1.0, "norm": 0.0}
    for li, (q, k, v, o, do, dq, dk, dv) in enumerate(rec):
        for b in range(q.shape[0]):
            for h in (0, 17, 47):
                qf = q[b, h].float() / (scale * log2e)          # the kernels take q pre-multiplied by scale * log2(e)
                kf, vf, of, dof = k[b, h].float(), v[b, h].float(), o[b, h].float(), do[b, h].float()
                p = torch.softmax((qf @ kf.t()) * scale, dim=-1)
                delta = (dof * of).sum(-1, keepdim=True)
                ds = (p * (dof @ vf.t() - delta)).bfloat16().float()
                ref = {"dq": (ds @ kf) * scale, "dk": (ds.t() @ qf) * scale, "dv": p.bfloat16().float().t() @ dof}
                for name, got in (("dq", dq[b, h]), ("dk", dk[b, h]), ("dv", dv[b, h])):
                    a, r = got.double().flatten(), ref[name].double().flatten()
                    cos = float((a * r).sum() / (a.norm() * r.norm()))
                    nrm = abs(float(a.norm() / r.norm()) - 1)
                    worst["cos"], worst["norm"] = min(worst["cos"], cos), max(worst["norm"], nrm)
                    assert cos >= 0.999 and nrm <= 0.01, (li, b, h, name, cos, nrm)
    print(json.dumps({"attention_backward_vs_rounding_injected_recompute": worst}))
