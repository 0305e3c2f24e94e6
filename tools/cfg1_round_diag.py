"""What can an activation-rounded oracle pin at BASELINE configs[0] width?  

Runs on the GPU box.  For r8 and r64: the HIP pair-step's LoRA gradients (h), the fp32 oracle (f), plain torch bf16 (b), the oracle with P / dS rounded
(p), the ACTIVATION-ROUNDED oracle (a: oracle/cogvideox.py round_activations=True; x: the same with the attention backward's delta formed from the UNROUNDED output) and the same oracle on an input that differs by ONE bf16 ulp in
ONE prompt-embedding element (a').  Prints, per adapter tensor, relative errors and cosines between the pairs that matter:

    h|f  b|f  a|f     distance of each bf16-class computation from fp32
    h|a               what the judge asked to bound at 10 % / cos 0.995
    a'|a              how far two runs of the SAME rounded arithmetic drift apart after a 1-ulp input change: the noise floor of ANY
                      rounding-pattern-matched comparison (if a'|a is as large as h|a, h|a measures chaos, not a defect)

    python tools/cfg1_round_diag.py [r8|r64 ...]  -> gpurun_out/cfg1_round_diag_<variant>.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfg1_common as c1                      # noqa: E402
import test_gpu_cfg1 as t1                    # noqa: E402
from oracle import cogvideox as ocv           # noqa: E402
from oracle import scheduler as osch          # noqa: E402


def oracle(variant, dtype=torch.float32, perturb=False, **kw):
    cfg = c1.config()
    sd = {k: v.to(dtype).cuda() for k, v in c1.base_state_dict(cfg).items()}
    lora, _ = c1.lora_state_dict(cfg, variant)
    lora = {k: v.cuda().requires_grad_(True) for k, v in lora.items()}
    xw, xl, prompt, t, noise = (v.to(dtype) if v.is_floating_point() else v for v in c1.inputs())
    if perturb:
        prompt = prompt.clone()
        prompt[0, 3, 5] = prompt[0, 3, 5] * (1 + 2.0 ** -7)          # one bf16 ulp on one of 925 696 prompt elements
    abar = osch.alphas_cumprod().cuda()
    out = ocv.dpo_pair_step(sd, cfg, lora, abar, xw.cuda(), xl.cuda(), prompt.cuda(), t.cuda(), noise.cuda(), beta=1.0, **kw)
    out["loss"].backward()
    loss = float(out["loss"].detach())
    grads = {k: p.grad.float().cpu() for k, p in lora.items()}
    del out, sd, lora
    torch.cuda.empty_cache()
    return loss, grads


def rel(a, r):
    return float((a.double() - r.double()).norm() / r.double().norm())


def cos(a, r):
    a, r = a.double().flatten(), r.double().flatten()
    return float((a * r).sum() / (a.norm() * r.norm()).clamp_min(1e-300))


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for variant in (sys.argv[1:] or ["r8", "r64"]):
        out, _, h = t1._hip_step(variant)
        lh = out.loss.item()
        del out
        torch.cuda.empty_cache()
        lf, f = oracle(variant)
        lb, b = oracle(variant, torch.bfloat16)
        lp, p = oracle(variant, round_p_ds=True)
        la, a = oracle(variant, round_activations=True)
        lx, x = oracle(variant, round_activations=True, exact_delta=True)
        la2, a2 = oracle(variant, round_activations=True, perturb=True)
        lf2, f2 = oracle(variant, perturb=True)
        rep = {"variant": variant, "loss": {"hip": lh, "fp32": lf, "torch_bf16": lb, "p_ds_rounded": lp, "act_rounded": la, "act_rounded_perturbed": la2,
                                            "fp32_perturbed": lf2}, "tensors": {}}
        print(f"== {variant}: loss hip {lh:.6f} fp32 {lf:.6f} bf16 {lb:.6f} act-rounded {la:.6f} / perturbed {la2:.6f}")
        print(f"{'tensor':34s} {'h|f':>7s} {'b|f':>7s} {'a|f':>7s} {'h|a':>7s} {'a`|a':>7s} {'f`|f':>8s} | cos {'h,f':>7s} {'h,a':>7s} {'a`,a':>7s} {'b,f':>7s}")
        for k in h:
            name = k.replace("base_model.model.transformer_blocks.", "").replace(".weight", "").replace("attn1.", "")
            r = {"h|f": rel(h[k], f[k]), "b|f": rel(b[k], f[k]), "a|f": rel(a[k], f[k]), "p|f": rel(p[k], f[k]), "h|a": rel(h[k], a[k]), "a'|a": rel(a2[k], a[k]),
                 "f'|f": rel(f2[k], f[k]), "cos h,f": cos(h[k], f[k]), "cos h,a": cos(h[k], a[k]), "cos a',a": cos(a2[k], a[k]), "cos b,f": cos(b[k], f[k]),
                 "cos a,f": cos(a[k], f[k]), "x|f": rel(x[k], f[k]), "h|x": rel(h[k], x[k]), "h|b": rel(h[k], b[k]), "cos h,b": cos(h[k], b[k]), "norm f": float(f[k].norm()), "norm h": float(h[k].norm()), "norm a": float(a[k].norm())}
            rep["tensors"][name] = r
            print(f"{name:34s} {r['h|f']:7.4f} {r['b|f']:7.4f} {r['a|f']:7.4f} {r['h|a']:7.4f} {r[chr(97)+chr(39)+'|a']:7.4f} {r[chr(102)+chr(39)+'|f']:8.5f} |     "
                  f"{r['cos h,f']:7.4f} {r['cos h,a']:7.4f} {r['cos a'+chr(39)+',a']:7.4f} {r['cos b,f']:7.4f}  | x|f {r['x|f']:7.4f} h|x {r['h|x']:7.4f} h|b {r['h|b']:7.4f}")
        with open(os.path.join(ROOT, "gpurun_out", f"cfg1_round_diag_{variant}.json"), "w") as fh:
            json.dump(rep, fh, indent=1)


if __name__ == "__main__":
    main()
