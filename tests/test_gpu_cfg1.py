"""Full-width parity at BASELINE.json configs[0] geometry: D = 3072, 48 heads, 2 transformer blocks, 13f x 64 x 64 paired
latents (S = 13 538 tokens), LoRA r = 8 (and r = 64: the rp = 64 / 192 kernels of the headline config).  The HIP
pair-step (train/CogVideoX-5B/03_train.py:116-157 through CogVideoXDPOTrainer._shared_step, every kernel through the
C-ABI) runs on the seeded inputs of tests/cfg1_common.py and is compared with the fp32 CPU-oracle results committed as
tests/golden/cfg1_<variant>.pt (made by tests/golden/make_cfg1_golden.py in the build container).

Tolerances (the HIP path computes in bf16, the oracle in fp32 on the same bf16-rounded weights / inputs):
  loss                  |d| <= 1e-3 (north_star) -- measured ~1e-6
  rewards (= -mean err) |d| <= 2e-4 + 1 % relative; reward_margin (their difference, 1e-3 of the rewards) |d| <= 1e-3
  v_pred samples        |d| <= 3 % of the prediction range (bf16 activations through 2 blocks)
  LoRA grads, per tensor: norm within 5 %, sampled entries within 5 % of the tensor's max |grad|, cosine >= 0.99 -- OR
                        within 3 x the error of the bf16 FLOOR of the tensor's family (to_q / to_k x A / B of one block; the other
                        adapters of a block), whichever is larger.  Measured: every tensor but the last block's q / k family
                        passes the fixed 5 % / 0.99 bounds with room (cos >= 0.995), where plain torch bf16 does not (cos 0.95-0.99).
                        In the r8 variant (B ~ N(0, 1e-3)) that family's gradient norm is 6e-5: below the bf16 noise of ANY
                        implementation (torch bf16: norm off by 48 %, cos 0.88) -- the factor 3 covers the run-to-run spread of
                        a noise-dominated quantity; it is not a statement about signal.  A family whose torch-bf16 floor is itself
                        below cos 0.95 is checked for direction (cos > 0.5) and magnitude (norm within a factor 2) only.

The bf16 floor: the same step run by the ORACLE'S OWN CODE (plain torch ops) on the GPU with bf16 weights and activations,
compared with the same fp32 golden.  It is needed for one family of tensors: the to_q / to_k adapters of the LAST block.
Their gradient is a heavily cancelling sum (|dQ| is 20x smaller than |dK| there, profiles/r02_cfg1_attention_bwd_diag.txt),
and any bf16 flash-attention backward rounds P and dS to bf16 before the dQ / dK products: re-computing the HIP kernel's
dQ in fp32 torch from the same inputs with ONLY those two roundings added reproduces its error to 5 digits
(cos 0.945049 vs 0.945045), i.e. the deviation is the arithmetic type, not the kernel.
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


def _bf16_floor(variant, gold):
    """Plain-torch bf16 run of the step on the GPU (oracle code) -> per-tensor (norm_rel, sample_err/max, cos) vs the golden."""
    from oracle import scheduler as osch
    cfg = c1.config()
    sd = {k: v.cuda() for k, v in c1.base_state_dict(cfg).items()}
    lora, _ = c1.lora_state_dict(cfg, variant)
    lora = {k: v.cuda().requires_grad_(True) for k, v in lora.items()}
    xw, xl, prompt, t, noise = c1.inputs()
    out = ocv.dpo_pair_step(sd, cfg, lora, osch.alphas_cumprod().cuda(), xw.cuda(), xl.cuda(), prompt.cuda(), t.cuda(), noise.cuda(), beta=1.0)
    out["loss"].backward()
    floor = {}
    for k, p in lora.items():
        ref = gold["lora_grads"][k]
        g = p.grad.float().cpu()
        got, rs = g.flatten()[c1.sample_index(g.numel(), k)].double(), ref["samples"].double()
        floor[k] = (abs(g.double().norm().item() / float(ref["norm"]) - 1), (got - rs).abs().max().item() / float(ref["absmax"]),
                    float((got * rs).sum() / (got.norm() * rs.norm()).clamp_min(1e-300)))
    return floor, float(out["loss"])


def _hip_step(variant):
    from videogpa_amd.lora import LoraConfig, get_peft_model
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    from videogpa_amd.transformer import COGVIDEOX_5B, CogVideoXTransformer3DModel
    cfg = c1.config()
    sd = c1.base_state_dict(cfg)
    model = CogVideoXTransformer3DModel(**dict(COGVIDEOX_5B, num_layers=cfg.num_layers, sample_height=c1.HEIGHT, sample_width=c1.WIDTH))
    model.load_state_dict(sd, strict=True)
    del sd
    model = model.to(device="cuda", dtype=torch.bfloat16)
    lora, r = c1.lora_state_dict(cfg, variant)
    pm = get_peft_model(model, LoraConfig(r=r, lora_alpha=2 * r, target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    own = pm.state_dict()
    for k, v in lora.items():
        own[k[:-len(".weight")] + ".default.weight"].copy_(v)
    tr = CogVideoXDPOTrainer({"beta": 1.0}, transformer=pm)
    tr.train()
    x_win, x_lose, prompt, t, noise = c1.inputs()
    captured = {}
    orig = tr.transformer.forward

    def spy(*a, **k):
        out = orig(*a, **k)
        captured.setdefault("preds", []).append(out.sample.detach())
        return out
    tr.transformer.forward = spy
    out = tr._shared_step({"x_win": x_win.cuda(), "x_lose": x_lose.cuda(), "prompt_emb": prompt.cuda()},
                          timesteps=t.cuda(), noise=noise.cuda())
    tr.transformer.forward = orig
    out.loss.backward()
    torch.cuda.synchronize()
    v_ref, v_pol = captured["preds"]            # reference pass first (adapter off), then the policy pass; batch = (win, lose)
    grads = {}
    named = dict(pm.named_parameters())
    for k in lora:
        grads[k] = named[k[:-len(".weight")] + ".default.weight"].grad.detach().float().cpu()
    return out, {"v_win": v_pol[0:1], "v_lose": v_pol[1:2], "v_win_ref": v_ref[0:1], "v_lose_ref": v_ref[1:2]}, grads


@pytest.mark.parametrize("variant", ["r8", "r64"])
def test_cfg1_pair_step_matches_oracle_golden(variant):
    gold = torch.load(os.path.join(HERE, "golden", f"cfg1_{variant}.pt"), weights_only=False)
    out, preds, grads = _hip_step(variant)
    report = {"variant": variant, "loss_hip": out.loss.item(), "loss_oracle": float(gold["loss"])}
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
    floor, floor_loss = _bf16_floor(variant, gold)
    report["loss_torch_bf16"] = floor_loss
    worst = {"norm_rel": 0.0, "sample_err_over_max": 0.0, "cos_min": 1.0}
    per_tensor = {}
    assert set(grads) == set(gold["lora_grads"])
    for k, g in grads.items():
        ref = gold["lora_grads"][k]
        idx = c1.sample_index(g.numel(), k)
        got = g.flatten()[idx].double()
        rs = ref["samples"].double()
        amax = float(ref["absmax"])
        nrel = abs(g.double().norm().item() / float(ref["norm"]) - 1)
        serr = (got - rs).abs().max().item() / amax
        cos = float((got * rs).sum() / (got.norm() * rs.norm()).clamp_min(1e-300))
        # the floor of a tensor's family: rounding noise is random, so q / k x A / B of one block share the worst of their four floors
        fam = [f for kk, f in floor.items() if kk.split(".attn1.")[0] == k.split(".attn1.")[0]
               and (("to_q" in kk or "to_k" in kk) == ("to_q" in k or "to_k" in k))]
        fn, fs, fc = max(f[0] for f in fam), max(f[1] for f in fam), min(f[2] for f in fam)
        per_tensor[k.replace("base_model.model.transformer_blocks.", "")] = {"hip": (round(nrel, 5), round(serr, 5), round(cos, 6)),
                                                                              "torch_bf16_floor": (round(fn, 5), round(fs, 5), round(fc, 6))}
        worst["norm_rel"] = max(worst["norm_rel"], nrel)
        worst["sample_err_over_max"] = max(worst["sample_err_over_max"], serr)
        worst["cos_min"] = min(worst["cos_min"], cos)
        if fc < 0.95:
            # noise-dominated family (plain torch bf16 itself is below cos 0.95 against fp32: the r8 variant's last-block q / k adapters,
            # gradient norm 6e-5): only direction and order of magnitude are meaningful, for either implementation
            check(math.isfinite(nrel) and nrel < 1.0 and cos > 0.5, (k, "noise-dominated family", nrel, cos, fn, fc))
            continue
        check(math.isfinite(nrel) and nrel < max(0.05, 3 * fn), (k, "norm", nrel, fn))
        check(serr < max(0.05, 3 * fs), (k, "sample", serr, fs))
        check(1 - cos < max(0.01, 3 * (1 - fc)), (k, "cos", cos, fc))
    report["lora_grads_worst"] = worst
    report["lora_grads_per_tensor (norm_rel, sample_err/max, cos)"] = per_tensor
    report["failed_checks"] = [str(f) for f in fails]
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(HERE), "gpurun_out", f"cfg1_parity_{variant}.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
    assert not fails, fails
