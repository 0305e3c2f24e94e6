"""Cycle count of the w1 forward's main loop (s_memtime, -DW1_CLOCKS build) next to its wall time -> cycles per half-step and the
effective shader clock.   VGPA_LIB=var/lib_w1_clk.so python tools/w1_clock.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import _lib, ops  # noqa: E402

B, H, S = 2, 48, 17776
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, S, 3, H, 64, generator=g, device="cuda").to(torch.bfloat16)
q = ops.prescale_q(qkv[:, :, 0].permute(0, 2, 1, 3).contiguous())
k = qkv[:, :, 1].permute(0, 2, 1, 3).contiguous()
v = qkv[:, :, 2].permute(0, 2, 1, 3)
o = torch.empty(B, S, H * 64, dtype=torch.bfloat16, device="cuda")
ov = o.view(B, S, H, 64).permute(0, 2, 1, 3)
lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
wsb = _lib.query("vgpa_attn_fwd_w1_workspace_bytes", B, H, S)
ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
st = ops._bhs_strides
for split in (0,):
    for it in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.call("vgpa_attn_fwd_w1", q, k, v, o, lse, st(q), st(k), st(v), st(ov), B, H, S, 64, 0.125, split, ws, wsb, torch.cuda.current_stream().cuda_stream)
        b.record()
        torch.cuda.synchronize()
    ntask = (S + 255) // 256 * B * H
    cyc = ws.view(torch.int32)[B * H:B * H + ntask].float()
    nhalf = 2 * ((S + 63) // 64 + 1)
    print(f"wall (incl. the redo pass of this diagnostic build) {a.elapsed_time(b):.3f} ms; loop cycles per workgroup: mean {cyc.mean().item():.0f} min {cyc.min().item():.0f} max {cyc.max().item():.0f}"
          f" -> {cyc.mean().item() / nhalf:.1f} cycles per half-step ({nhalf} half-steps)")
