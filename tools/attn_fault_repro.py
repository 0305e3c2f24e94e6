"""Catch the forward-attention call that turns finite operands into a non-finite output inside the trained_like cfg2 model (tools/trained_like_diag.py found it at
block 38, QK-norm gain 2.5), and take it apart: which (batch, head, strip) rows, what the strip flags say, what the all-online entry and the fp32 reference give on
the same operands, and the row statistics the shift is built from (bound, sampled maximum, true maximum).  Saves the operands of one faulty head.
    python tools/attn_fault_repro.py [--qk-gain 2.5] [--out gpurun_out/attn_fault]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class Found(Exception):
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qk-gain", type=float, default=2.5)
    ap.add_argument("--layers", type=int, default=42)
    ap.add_argument("--out", default="gpurun_out/attn_fault")
    args = ap.parse_args()
    from videogpa_amd import ops, _lib, transformer as vtr
    from videogpa_amd.trainer import CogVideoXDPOTrainer
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
    trainer.train()
    g = torch.Generator(device=dev).manual_seed(1234)
    x_pair = (0.7 * torch.randn(1, 2, C["frames"], 16, C["height"], C["width"], generator=g, device=dev)).to(torch.bfloat16)
    prompt = (0.2 * torch.randn(1, bench.TEXT_LEN, 4096, generator=g, device=dev)).to(torch.bfloat16)
    batch = {"x_pair": x_pair, "prompt_emb": prompt}

    orig = ops.attention_fwd_raw
    box = {"calls": 0}

    def wrapped(q, k, v, *a, **kw):
        o, lse = orig(q, k, v, *a, **kw)
        fin_in = bool(torch.isfinite(q.float()).all() and torch.isfinite(k.float()).all() and torch.isfinite(v.float()).all())
        fin_out = bool(torch.isfinite(o.float()).all() and torch.isfinite(lse).all())
        if fin_in and not fin_out:
            box.update(q=q.detach(), k=k.detach(), v=v.detach(), o=o.detach(), lse=lse.detach(), call=box["calls"], kw={k_: v_ for k_, v_ in kw.items() if k_ != "policy" and k_ != "o_res"})
            raise Found()
        box["calls"] += 1
        return o, lse
    ops.attention_fwd_raw = wrapped
    try:
        with torch.no_grad():
            trainer.training_step(batch, 0)
        print("no faulty call found")
        return
    except Found:
        pass
    finally:
        ops.attention_fwd_raw = orig
    q, k, v, o, lse = box["q"], box["k"], box["v"], box["o"], box["lse"]
    B, H, S, Dh = q.shape
    rep = {"call_index": box["call"], "shape": [B, H, S, Dh], "kw": {k_: str(v_) for k_, v_ in box["kw"].items()}}
    ov = o.unflatten(-1, (H, Dh)).permute(0, 2, 1, 3) if o.dim() == 3 else o
    bad_o = ~torch.isfinite(ov.float()).all(-1)             # [B,H,S]
    bad_l = ~torch.isfinite(lse)
    rep["rows_bad_o"] = int(bad_o.sum())
    rep["rows_bad_lse"] = int(bad_l.sum())
    per_bh = bad_o.sum(-1)
    rep["bad_rows_per_bh"] = {f"{b},{h}": int(per_bh[b, h]) for b in range(B) for h in range(H) if per_bh[b, h] > 0}
    bh = [(b, h) for b in range(B) for h in range(H) if per_bh[b, h] > 0]
    b0, h0 = bh[0]
    rows = torch.nonzero(bad_o[b0, h0]).flatten()
    rep["first_bad_head"] = [b0, h0]
    rep["bad_rows_first_head"] = {"count": int(rows.numel()), "min": int(rows.min()), "max": int(rows.max()), "strips": sorted({int(r) // 256 for r in rows.tolist()})[:40]}
    # the same call again with an own workspace, to read the strip flags; then the all-online entry
    kwq = dict(box["kw"])
    for name, pol in (("bound", ops.AttnFwdPolicy(mode="bound", fixed=True)), ("online", ops.AttnFwdPolicy(mode="online", fixed=True))):
        o2, l2 = orig(q, k, v, policy=pol, **kwq)
        o2v = o2.unflatten(-1, (H, Dh)).permute(0, 2, 1, 3)
        rep[f"rerun_{name}"] = {"rows_bad_o": int((~torch.isfinite(o2v.float()).all(-1)).sum()), "rows_bad_lse": int((~torch.isfinite(l2)).sum())}
    ws_bytes = _lib.query("vgpa_attn_fwd_w1_workspace_bytes", B, H, S)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    o3 = torch.empty(B, S, H * Dh, dtype=torch.bfloat16, device=dev)
    l3 = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    o3v = o3.unflatten(-1, (H, Dh)).permute(0, 2, 1, 3)
    _lib.call("vgpa_attn_fwd_w1_res", q, k, v, o3, None, ops._res_kind(None), l3, ops._bhs_strides(q), ops._bhs_strides(k), ops._bhs_strides(v), ops._bhs_strides(o3v), None,
              B, H, S, Dh, float(Dh ** -0.5), int(ops.ATTN_SPLIT_MODE), ws, ws_bytes, ops._stream())
    torch.cuda.synchronize()
    tasks = B * H * ((S + 255) // 256)
    flags = ws[:4 * (B * H + tasks)].view(torch.int32)[B * H:].view(B, H, -1)
    kmax2 = ws[:4 * B * H].view(torch.float32).view(B, H)
    rep["flags_set_total"] = int((flags != 0).sum())
    rep["flags_first_head"] = [int(i) for i in torch.nonzero(flags[b0, h0]).flatten().tolist()][:80]
    rep["flag_values_first_head"] = sorted({int(x) for x in flags[b0, h0].tolist()})[:10]
    rep["kmax_first_head"] = float(kmax2[b0, h0].sqrt())
    bad3 = ~torch.isfinite(o3v.float()).all(-1)
    rep["direct_call_rows_bad_o"] = int(bad3.sum())
    # row statistics of the first faulty head
    qh, kh, vh = q[b0, h0].float(), k[b0, h0].float(), v[b0, h0].float()
    s2 = qh @ kh.t()                                          # q is pre-scaled: log2 units
    mx = s2.max(-1).values
    step = S // 64
    ms = s2[:, torch.arange(64, device=dev) * step].max(-1).values
    bound = qh.norm(dim=-1) * kh.norm(dim=-1).max() * 1.0009765625
    mp = torch.minimum(bound, ms + 64.0)
    rep["row_stats_first_head"] = {
        "bound_max": float(bound.max()), "true_max_minus_shift_max": float((mx - mp).max()), "true_max_minus_shift_min": float((mx - mp).min()),
        "bad_rows_true_max_minus_shift": [float(x) for x in (mx - mp)[rows[:8]].tolist()],
        "bad_rows_bound": [float(x) for x in bound[rows[:8]].tolist()], "bad_rows_sampled_max": [float(x) for x in ms[rows[:8]].tolist()],
        "bad_rows_true_max": [float(x) for x in mx[rows[:8]].tolist()],
        "k_finite": bool(torch.isfinite(kh).all()), "v_absmax": float(vh.abs().max()), "q_absmax": float(qh.abs().max()), "k_absmax": float(kh.abs().max())}
    p = torch.softmax(s2 * 0.6931471805599453, -1)
    ref = p @ vh
    good = torch.isfinite(ov[b0, h0].float()).all(-1)
    rep["err_on_finite_rows_first_head"] = float((ov[b0, h0].float()[good] - ref[good]).abs().max()) if good.any() else None
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    torch.save({"q": q[b0, h0].cpu(), "k": k[b0, h0].cpu(), "v": v[b0, h0].cpu(), "o": ov[b0, h0].cpu(), "lse": lse[b0, h0].cpu(), "flags": flags[b0, h0].cpu()}, args.out + "_head.pt")
    with open(args.out + ".json", "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
