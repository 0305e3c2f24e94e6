"""Full-width golden of BASELINE.json configs[0]: one DPO preference-pair step (train/CogVideoX-5B/03_train.py:116-157)
at D = 3072 / 48 heads / 2 blocks / 13f x 64 x 64 latents (S = 13 538 tokens), fp32, run with the CPU oracle in the
build container.  Takes ~10 minutes per variant on 8 cores and ~20 GB; writes tests/golden/cfg1_<variant>.pt
(a few hundred KB: scalars, per-tensor LoRA-gradient norms, sampled gradient / prediction entries -- no weights, the
inputs are regenerated from seeds by tests/cfg1_common.py).

    python tests/golden/make_cfg1_golden.py [r8] [r64]

diffusers / peft are not installed, so this is the ORACLE's output (parity unpinned at the diffusers boundary,
oracle/cogvideox.py header), not the reference's; what it pins is the assembled HIP path at full width.
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cfg1_common as c1  # noqa: E402
from oracle import cogvideox as ocv  # noqa: E402
from oracle import scheduler as osch  # noqa: E402


def run(variant):
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = c1.config()
    sd = {k: v.float() for k, v in c1.base_state_dict(cfg).items()}
    lora, r = c1.lora_state_dict(cfg, variant)
    lora = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
    x_win, x_lose, prompt, t, noise = c1.inputs()
    t0 = time.time()
    out = ocv.dpo_pair_step(sd, cfg, lora, osch.alphas_cumprod(), x_win.float(), x_lose.float(), prompt.float(), t, noise.float(),
                            beta=1.0, lora_scale=2.0)
    t1 = time.time()
    out["loss"].backward()
    t2 = time.time()
    gold = {"variant": variant, "rank": r, "seconds_fwd": t1 - t0, "seconds_bwd": t2 - t1, "threads": torch.get_num_threads(),
            "torch": torch.__version__}
    for k in ("loss", "reward_margin", "winner_reward", "loser_reward", "accuracy", "logits"):
        gold[k] = out[k].detach().double()
    for k in ("v_win", "v_lose", "v_win_ref", "v_lose_ref"):
        v = out[k].detach()
        idx = c1.sample_index(v.numel(), k)
        gold[k + "_norm"] = v.double().norm()
        gold[k + "_samples"] = v.flatten()[idx].clone()
    # policy - reference prediction difference: what the DPO logits are made of
    for a, b in (("v_win", "v_win_ref"), ("v_lose", "v_lose_ref")):
        d = (out[a] - out[b]).detach()
        gold[a + "_minus_ref_norm"] = d.double().norm()
        gold[a + "_minus_ref_samples"] = d.flatten()[c1.sample_index(d.numel(), a)].clone()
    grads = {}
    for k, p in lora.items():
        gsum = p.grad.detach()
        idx = c1.sample_index(gsum.numel(), k)
        grads[k] = {"norm": gsum.double().norm(), "samples": gsum.flatten()[idx].clone(),
                    "absmax": gsum.abs().max()}
    gold["lora_grads"] = grads
    path = os.path.join(HERE, f"cfg1_{variant}.pt")
    torch.save(gold, path)
    print(f"{path}: loss {float(gold['loss']):.7f} margin {float(gold['reward_margin']):.3e} "
          f"fwd {t1 - t0:.0f}s bwd {t2 - t1:.0f}s", flush=True)


if __name__ == "__main__":
    for v in (sys.argv[1:] or list(c1.VARIANTS)):
        run(v)
