"""Loss CURVE at model width (north_star: "loss curve matching reference within 1e-3"; VERDICT r5 missing 3): 20 optimizer steps x accumulate 2 (the reference's
train/CogVideoX-5B/03_train.py:61,265) = 40 preference-pair micro-steps at the geometry of BASELINE configs[0] -- D = 3072, 48 heads, 2 blocks, 13f x 64 x 64
latents (S = 13 538), LoRA r = 64 / alpha 128 on to_q / to_k / to_v / to_out.0, beta = 1 (:56) -- on the HIP engine (CogVideoXDPOTrainer._shared_step, FlatAdamW:
fused clip + AdamW, cosine warm-up) against the SAME loop run by the oracle in fp32 on the same GPU (oracle/cogvideox.py::dpo_pair_step with the adapters'
bf16 copies inside the forward as PEFT's autocast makes them, torch.optim.AdamW + clip_grad_norm_ + the cosine-warm-up multiplier), identical pairs, timesteps and
noise.  Four fixed pairs are cycled (10 passes over them), with a learning rate (6e-5, 12 x the reference's 5e-6) at which the loss on those pairs falls by more than
1e-2 within the 20 steps: a curve that really moves.  (The error grows with the distance the loss has moved: at 1.5e-4 one pair's loss falls from 0.693 to 0.449 and the device is 1.6e-3 off there, 0.9e-3 or less everywhere else
(profiles/r06_loss_curve_width_lr1.5e-4.json); at 1e-3 the same loop is unstable in fp32 already -- the loss on one pair jumps from 0.26 to 1.18 between two
passes -- and two runs of ANY arithmetic type separate: measured in round 6, profiles/r06_loss_curve_width_lr1e-3.json.)  The toy-sized test this supersedes as evidence (tests/test_gpu_model.py::test_loss_curve_matches_oracle_training_loop: 54 tokens, 2 heads) stays as a fast check.

Three device modes (model-level settings, ops.py "Precise delta" / transformer.enable_lean_activations):
    int8        the default: the attention backward's delta from the completed output (8 further mantissa bits per element)
    int8_lean   + lean activations (LN output and normalised q / k made again in the backward) -- must be BIT-identical to int8, micro-step by micro-step
    plain       precise_delta None: the textbook flash-attention backward (delta from the stored bf16 output) -- its trajectory cost goes on record
Asserted: |loss_hip - loss_oracle| <= 1e-3 at EVERY micro-step (all modes), the loss on the four pairs falls by >= 1e-2 between the first and the last pass, and the
adapters stay together: ||theta_hip - theta_oracle|| <= 35 % of the distance ||theta_oracle - theta_0|| the training moved them (AdamW moves every element
by about lr per step whatever its gradient's size, so elements whose gradient is below the bf16 noise floor walk in unrelated directions on the two sides: the drift is
reported, the loss is what is held to 1e-3).
Report: gpurun_out/loss_curve_width.json -> profiles/r06_loss_curve_width.json."""
import gc
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import cfg1_common as c1
from oracle import cogvideox as ocv
from oracle import scheduler as osch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OPT_STEPS, ACCUM, N_PAIRS = 20, 2, 4
CONF = {"beta": 1.0, "learning_rate": 6e-5, "weight_decay": 0.01, "warmup_steps": 2, "max_steps": OPT_STEPS, "accumulate_grad_batches": ACCUM, "gradient_clip_val": 1.0}
TIMESTEPS = (417, 83, 901, 640)
LOSS_TOL, MOVE_MIN, DRIFT_MAX = 1e-3, 1e-2, 0.35
# VGPA_GPU_FULL=1: also the lean-activation arm (bit-identical to the default: asserted) and the textbook-delta arm (what the precise delta buys: profiles/r06_loss_curve_width.json
# holds all three); the default suite runs the product's default mode only -- the four longest tests of the suite took 6 of its 10 minutes
FULL = os.environ.get("VGPA_GPU_FULL", "0") == "1"
MODES = {"int8": ("int8", False), "int8_lean": ("int8", True), "plain": (None, False)} if FULL else {"int8": ("int8", False)}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _pairs():
    out = []
    for i in range(N_PAIRS):
        g = torch.Generator().manual_seed(5000 + i)
        xw = (0.7 * torch.randn(1, 16, c1.FRAMES, c1.HEIGHT, c1.WIDTH, generator=g)).to(torch.bfloat16)
        xl = (0.7 * torch.randn(1, 16, c1.FRAMES, c1.HEIGHT, c1.WIDTH, generator=g)).to(torch.bfloat16)
        txt = (0.2 * torch.randn(1, c1.TEXT_LEN, 4096, generator=g)).to(torch.bfloat16)
        eps = torch.randn(1, c1.FRAMES, 16, c1.HEIGHT, c1.WIDTH, generator=g).to(torch.bfloat16)
        out.append(tuple(v.cuda() for v in (xw, xl, txt, torch.tensor([TIMESTEPS[i]]), eps)))
    return out


def _oracle_curve():
    """the reference loop in fp32: -> (losses per micro-step, {key: final adapter}, {key: initial adapter})"""
    from videogpa_amd.optim import cosine_schedule_with_warmup
    cfg = c1.config()
    sd = {k: v.float().cuda() for k, v in c1.base_state_dict(cfg).items()}
    lora0, r = c1.lora_state_dict(cfg, "r64")
    params = {k: v.clone().cuda().requires_grad_(True) for k, v in lora0.items()}
    opt = torch.optim.AdamW(list(params.values()), lr=CONF["learning_rate"], weight_decay=CONF["weight_decay"])
    abar = osch.alphas_cumprod().cuda()
    pairs = _pairs()
    losses, micro = [], 0
    for step in range(OPT_STEPS):
        for grp in opt.param_groups:
            grp["lr"] = CONF["learning_rate"] * cosine_schedule_with_warmup(step, CONF["warmup_steps"], CONF["max_steps"])
        opt.zero_grad()
        for _ in range(ACCUM):
            xw, xl, txt, t, eps = pairs[micro % N_PAIRS]
            # the adapters' bf16 copies inside the forward (PEFT casts the fp32 adapter weights under bf16 autocast), gradient straight through to the fp32 master
            lb = {k: (v.detach().bfloat16().float() - v.detach()) + v for k, v in params.items()}
            out = ocv.dpo_pair_step(sd, cfg, lb, abar, xw.float(), xl.float(), txt.float(), t, eps.float(), beta=CONF["beta"], chunked_attention=True)
            (out["loss"] / ACCUM).backward()
            losses.append(float(out["loss"].detach()))
            del out, lb
            micro += 1
        torch.nn.utils.clip_grad_norm_(list(params.values()), CONF["gradient_clip_val"])
        opt.step()
    final = {k: v.detach().clone() for k, v in params.items()}
    init = {k: v.cuda() for k, v in lora0.items()}
    del sd, params, opt
    gc.collect()
    torch.cuda.empty_cache()
    return losses, final, init


def _hip_curve(mode):
    from videogpa_amd.lora import LoraConfig, get_peft_model
    from videogpa_amd.trainer import CogVideoXDPOTrainer, DPOEngine
    from videogpa_amd.transformer import COGVIDEOX_5B, CogVideoXTransformer3DModel
    precise, lean = MODES[mode]
    cfg = c1.config()
    model = CogVideoXTransformer3DModel(**dict(COGVIDEOX_5B, num_layers=cfg.num_layers, sample_height=c1.HEIGHT, sample_width=c1.WIDTH))
    model.load_state_dict(c1.base_state_dict(cfg), strict=True)
    model = model.to(device="cuda", dtype=torch.bfloat16)
    model.set_precise_delta(precise)
    model.enable_lean_activations(lean)
    lora0, r = c1.lora_state_dict(cfg, "r64")
    pm = get_peft_model(model, LoraConfig(r=r, lora_alpha=2 * r, target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    own = pm.state_dict()
    for k, v in lora0.items():
        own[k[:-len(".weight")] + ".default.weight"].copy_(v)
    tr = CogVideoXDPOTrainer(dict(CONF, lean_activations=bool(lean)), transformer=pm)
    tr.train()
    eng = DPOEngine(tr, overlap=False)
    pairs = _pairs()
    losses, micro = [], 0
    for step in range(OPT_STEPS):
        for _ in range(ACCUM):
            xw, xl, txt, t, eps = pairs[micro % N_PAIRS]
            out = tr._shared_step({"x_win": xw, "x_lose": xl, "prompt_emb": txt}, timesteps=t, noise=eps)
            (out.loss / ACCUM).backward()
            losses.append(out.loss.item())
            micro += 1
        eng.opt.step(eng.opt.all_reduce_grads())
        eng.opt.zero_grad()
    named = dict(pm.named_parameters())
    final = {k: named[k[:-len(".weight")] + ".default.weight"].detach().float().clone() for k in lora0}
    del eng, tr, pm, model, named
    gc.collect()
    torch.cuda.empty_cache()
    return losses, final


def test_loss_curve_at_model_width_tracks_the_fp32_oracle_loop():
    ref_losses, ref_final, init = _oracle_curve()
    report = {"config": f"D=3072, 48 heads, 2 blocks, S={c1.TEXT_LEN + c1.FRAMES * (c1.HEIGHT // 2) * (c1.WIDTH // 2)}, r=64, beta=1, {OPT_STEPS} optimizer steps x accumulate {ACCUM}, "
                        f"{N_PAIRS} fixed pairs cycled, AdamW lr {CONF['learning_rate']} (warm-up {CONF['warmup_steps']}, cosine), wd {CONF['weight_decay']}, clip {CONF['gradient_clip_val']}",
              "bounds": {"loss_abs": LOSS_TOL, "loss_must_move_by": MOVE_MIN, "adapter_drift_over_distance_travelled": DRIFT_MAX},
              "oracle_fp32_losses": ref_losses}
    first, last = sum(ref_losses[:N_PAIRS]) / N_PAIRS, sum(ref_losses[-N_PAIRS:]) / N_PAIRS
    report["oracle_mean_loss_first_pass"], report["oracle_mean_loss_last_pass"] = first, last
    travelled = float(torch.sqrt(sum(((ref_final[k].double() - init[k].double()) ** 2).sum() for k in init)))
    report["oracle_adapter_distance_travelled"] = travelled
    fails = []
    curves = {}
    for mode in MODES:
        losses, final = _hip_curve(mode)
        curves[mode] = losses
        diffs = [abs(a - b) for a, b in zip(losses, ref_losses)]
        drift = float(torch.sqrt(sum(((final[k].double() - ref_final[k].double()) ** 2).sum() for k in init)))
        worst_el = max(float((final[k] - ref_final[k]).abs().max()) for k in init)
        report[mode] = {"losses": losses, "abs_err": diffs, "max_abs_err": max(diffs), "max_abs_err_at_micro_step": diffs.index(max(diffs)),
                        "mean_loss_first_pass": sum(losses[:N_PAIRS]) / N_PAIRS, "mean_loss_last_pass": sum(losses[-N_PAIRS:]) / N_PAIRS,
                        "adapter_drift_l2": drift, "adapter_drift_over_distance_travelled": drift / travelled, "adapter_max_abs_element_drift": worst_el}
        if max(diffs) > LOSS_TOL:
            fails.append((mode, "loss", max(diffs), diffs.index(max(diffs))))
        if drift / travelled > DRIFT_MAX:
            fails.append((mode, "adapter drift", drift / travelled))
    if first - last < MOVE_MIN:
        fails.append(("the curve does not move", first, last))
    if "int8_lean" in curves:
        report["int8_lean_bit_identical_to_int8"] = curves["int8"] == curves["int8_lean"]
        if not report["int8_lean_bit_identical_to_int8"]:
            fails.append(("lean activations changed the losses", [i for i, (a, b) in enumerate(zip(curves["int8"], curves["int8_lean"])) if a != b][:5]))
    report["failures"] = [repr(f) for f in fails]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "loss_curve_width.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps({"first_pass": first, "last_pass": last, "travelled": travelled,
                      **{m: {k: report[m][k] for k in ("max_abs_err", "adapter_drift_over_distance_travelled", "mean_loss_last_pass")} for m in MODES}}))
    assert not fails, fails
