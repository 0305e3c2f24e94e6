"""Where does `bench.py --weights trained_like` lose its loss to NaN?  Builds the cfg2 model exactly as bench.py does (random N(0, 0.02) weights, QK-norm gains of
--qk-gain with 3 x outlier channels), runs micro-steps through the DPO engine and reports, per transformer block and per pass (reference, policy), the magnitude
and finiteness of the residual stream leaving the block -- a bf16 overflow of a random 42-block network (sharp attention no longer averages V away) and a kernel
fault look the same in the loss but not here: an overflow grows block by block in BOTH passes, a kernel fault appears out of finite inputs.
    python tools/trained_like_diag.py [--qk-gain 2.5] [--layers 42] [--steps 2]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qk-gain", type=float, default=2.5)
    ap.add_argument("--layers", type=int, default=42)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from videogpa_amd import transformer as vtr
    from videogpa_amd.trainer import CogVideoXDPOTrainer, DPOEngine
    dev = torch.device("cuda", 0)
    C = bench.CONFIGS["cfg2"]
    cfg_kw = dict(getattr(vtr, C["model"]), num_layers=args.layers)
    torch.manual_seed(0)
    model = bench.build_model(cfg_kw, dev, seed=0)
    gq = torch.Generator(device=dev).manual_seed(7)
    with torch.no_grad():
        for blk in model.transformer_blocks:
            for nrm in (blk.attn1.norm_q, blk.attn1.norm_k):
                w = args.qk_gain * (1 + 0.2 * torch.randn(64, generator=gq, device=dev))
                w[:3] *= 3.0
                nrm.weight.copy_(w.to(nrm.weight.dtype))
                nrm.bias.copy_((0.1 * args.qk_gain * torch.randn(64, generator=gq, device=dev)).to(nrm.bias.dtype))
    trainer = CogVideoXDPOTrainer({"lora_rank": 64, "lora_alpha": 128, "beta": 1.0, "accumulate_grad_batches": 1, "enable_gradient_checkpointing": False,
                                   "lean_activations": False, "seed": 1234}, transformer=model)
    gB = torch.Generator(device=dev).manual_seed(1)
    with torch.no_grad():
        for n, p in trainer.transformer.named_parameters():
            if ".lora_B." in n:
                p.normal_(0.0, 1e-3, generator=gB)
    trainer.train()
    engine = DPOEngine(trainer)
    g = torch.Generator(device=dev).manual_seed(1234)
    x_pair = (0.7 * torch.randn(1, 2, C["frames"], 16, C["height"], C["width"], generator=g, device=dev)).to(torch.bfloat16)
    prompt = (0.2 * torch.randn(1, bench.TEXT_LEN, 4096, generator=g, device=dev)).to(torch.bfloat16)
    batch = {"x_pair": x_pair, "prompt_emb": prompt}

    rec = []          # (call index of the block, block index, absmax per sequence, finite)

    def hook(i):
        def f(mod, inp, out):
            x = out[0] if isinstance(out, (tuple, list)) else out
            xf = x.detach().float()
            per_seq = xf.abs().flatten(1).max(dim=1).values
            rec.append({"block": i, "absmax_per_sequence": [float(v) for v in per_seq.cpu()], "finite": bool(torch.isfinite(xf).all()),
                        "rms": float(xf[torch.isfinite(xf)].pow(2).mean().sqrt()) if torch.isfinite(xf).any() else None})
        return f
    rec_a = []        # the attention module's output (before the gated residual add): finite inputs -> non-finite output would be a kernel fault

    def hook_attn(i):
        def f(mod, inp, out):
            a = (out[0] if isinstance(out, (tuple, list)) else out).detach().float()
            n_in = inp[0].detach().float()
            rec_a.append({"block": i, "in_finite": bool(torch.isfinite(n_in).all()), "out_finite": bool(torch.isfinite(a).all()),
                          "out_absmax": float(a[torch.isfinite(a)].abs().max()) if torch.isfinite(a).any() else None})
        return f
    base = trainer.transformer.get_base_model()
    for i, blk in enumerate(base.transformer_blocks):
        blk.register_forward_hook(hook(i))
        blk.attn1.register_forward_hook(hook_attn(i))
    out = {"qk_gain": args.qk_gain, "layers": args.layers, "steps": []}
    for s in range(args.steps):
        rec.clear()
        rec_a.clear()
        logs = engine.micro_step(batch)
        logs.update(engine.flush())
        torch.cuda.synchronize()
        passes = [rec[k:k + args.layers] for k in range(0, len(rec), args.layers)]
        st = {"loss": float(logs["train/loss"]), "grad_norm": float(engine.opt.total_norm), "passes": []}
        for pi, p in enumerate(passes):
            first_bad = next((r["block"] for r in p if not r["finite"]), None)
            st["passes"].append({"pass": pi, "first_non_finite_block": first_bad, "rms_by_block": [r["rms"] for r in p],
                                 "absmax_by_block": [max(r["absmax_per_sequence"]) for r in p]})
        faults = [r for r in rec_a if r["in_finite"] and not r["out_finite"]]
        st["attention_finite_in_nonfinite_out"] = faults
        st["attention_out_absmax_by_call"] = [r["out_absmax"] for r in rec_a]
        print(f"  attention modules with finite input and non-finite output: {[(k, r['block']) for k, r in enumerate(rec_a) if r['in_finite'] and not r['out_finite']]}")
        st["attention_forward"] = [dict(r) for r in base.attention_forward_report()]
        out["steps"].append(st)
        print(f"step {s}: loss {st['loss']:.6f}  grad_norm {st['grad_norm']}")
        for p in st["passes"]:
            am = p["absmax_by_block"]
            print(f"  pass {p['pass']}: first non-finite block {p['first_non_finite_block']}; absmax block 0 / {args.layers // 2} / last: "
                  f"{am[0]:.3g} / {am[len(am) // 2]:.3g} / {am[-1]:.3g}; rms last {p['rms_by_block'][-1]}")
        fr = [r["redo_fraction"] for r in st["attention_forward"] if r["redo_fraction"] is not None]
        print(f"  attention: redo fraction max {max(fr) if fr else None}, layers online {sum(r['mode'] == 'online' for r in st['attention_forward'])}")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
