"""Stand-alone timing of the attention kernels at the headline shape (2 sequences x 48 heads x S=17776 x 64).
    python tools/attn_bench.py [--iters 5] [--S 17776] [--B 2] [--H 48] [--which fwd,dkv,dq]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--S", type=int, default=17776)
ap.add_argument("--B", type=int, default=2)
ap.add_argument("--H", type=int, default=48)
ap.add_argument("--which", default="fwd,dkv,dq")
ap.add_argument("--split", type=int, default=-1, help="forward split_mode: -1 automatic, 0 never")
ap.add_argument("--data", default="randn", choices=["randn", "zeros", "const"], help="operand values: zeros / one constant toggle almost no datapath bits -> the time of the "
                "instruction stream without the power throttle that random data brings (DESIGN section 4.2)")
ap.add_argument("--energy", type=float, default=0.0, help="seconds of back-to-back launches per op bracketed by the socket energy counter (tools/energy.py): "
                "joules per launch and mean power next to the time, also for the vendor FF1 GEMM of the step as a yardstick")
a = ap.parse_args()
B, H, S = a.B, a.H, a.S
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, S, 3, H, 64, generator=g, device="cuda").to(torch.bfloat16)
if a.data == "zeros":
    qkv.zero_()
elif a.data == "const":
    qkv.fill_(0.125)
q = qkv[:, :, 0].permute(0, 2, 1, 3).contiguous()
k = qkv[:, :, 1].permute(0, 2, 1, 3).contiguous()
v = qkv[:, :, 2].permute(0, 2, 1, 3)           # token-major view, as in the model
do = torch.randn(B, S, H * 64, generator=g, device="cuda").to(torch.bfloat16)
if a.data != "randn":
    do.fill_(0.0 if a.data == "zeros" else 0.125)
dov = do.view(B, S, H, 64).permute(0, 2, 1, 3)
dq, dk = torch.empty_like(q), torch.empty_like(k)
dv = torch.empty(B, S, H, 64, dtype=torch.bfloat16, device="cuda").permute(0, 2, 1, 3)
o, lse = ops.attention_fwd_raw(q, k, v)
ov = o.view(B, S, H, 64).permute(0, 2, 1, 3)
ops.attention_bwd_raw(q, k, v, ov, dov, lse, dq, dk, dv)
torch.cuda.synchronize()
ops.TIMER = ops.KernelTimer()
for _ in range(a.iters):
    if "fwd" in a.which:
        ops.attention_fwd_raw(q, k, v, split_mode=a.split)
    if "dkv" in a.which or "dq" in a.which:
        ops.attention_bwd_raw(q, k, v, ov, dov, lse, dq, dk, dv, split_mode=a.split)
torch.cuda.synchronize()
unit = 2.0 * S * S * 64 * B * H
hw_units = {"attn_fwd_kernel": 2, "attn_bwd_dkv_kernel": 4, "attn_bwd_dq_kernel": 3, "attn_bwd_fused_kernel": 5}
for name, s in ops.TIMER.summary().items():
    if name not in hw_units:          # HBM-bound helpers (attn_delta_kernel): bytes, not FLOPs
        print(f"{name:22s} avg {s['avg_ms']:8.3f} ms  {s['work_per_launch'] / s['avg_ms'] / 1e6:7.1f} GB/s")
        continue
    print(f"{name:22s} avg {s['avg_ms']:8.3f} ms  algorithmic {s['work_per_launch'] / s['avg_ms'] / 1e9:7.1f} TF/s  "
          f"executed-MFMA {hw_units[name] * unit / s['avg_ms'] / 1e9:7.1f} TF/s")


if a.energy > 0:
    import time
    from energy import read_joules
    from videogpa_amd import _lib
    w1 = set(ops.ATTN_W1)
    scale = 64 ** -0.5
    qs = ops.prescale_q(q, scale)
    delta = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
    stats = torch.empty(B, H, 2, S, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    sb = ops._bhs_strides
    _lib.call("vgpa_attn_bwd_prep_w1_res", ov, None, 0, dov, lse, sb(ov), None, sb(dov), delta, stats, B, H, S, 64, st)
    ws_bytes = _lib.query("vgpa_attn_bwd_split_workspace_bytes", B, H, S) if a.split != 0 else 0
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device="cuda")
    wsp = ws if ws_bytes else None
    x = torch.randn(B * S, 3072, generator=g, device="cuda").to(torch.bfloat16)
    wt = (torch.randn(12288, 3072, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    if a.data != "randn":
        x.fill_(0.0 if a.data == "zeros" else 0.125)
    launches = {
        "attn_fwd": (4.0, lambda: ops.attention_fwd_raw(q, k, v, split_mode=a.split)),
        "attn_bwd_dkv": (6.0, (lambda: _lib.call("vgpa_attn_bwd_dkv_w1", qs, k, v, dov, stats, dk, dv, sb(qs), sb(k), sb(v), sb(dov), sb(dk), sb(dv), B, H, S, 64, scale,
                                                   a.split, wsp, ws_bytes, st)) if "dkv" in w1 else
                         (lambda: _lib.call("vgpa_attn_bwd_dkv_ws", qs, k, v, dov, lse, delta, dk, dv, sb(qs), sb(k), sb(v), sb(dov), sb(dk), sb(dv), B, H, S, 64, scale,
                                            a.split, wsp, ws_bytes, st))),
        "attn_bwd_dq": (2.0, lambda: _lib.call("vgpa_attn_bwd_dq_w1" if "dq" in w1 else "vgpa_attn_bwd_dq_ws", qs, k, v, dov, lse, delta, dq, sb(qs), sb(k), sb(v),
                                                 sb(dov), sb(dq), B, H, S, 64, scale, a.split, wsp, ws_bytes, st)),
        "hipblaslt FF1 gemm": (None, lambda: torch.matmul(x, wt.t())),
    }
    idle0 = read_joules()
    time.sleep(1.0)
    idle_w = read_joules() - idle0
    print(f"energy: idle {idle_w:.0f} W; >= {a.energy:.1f} s of back-to-back launches per op, data = {a.data}")
    for name, (mult, fn) in launches.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        n = max(8, int(a.energy / max(time.perf_counter() - t0, 1e-4)))
        e0, t0 = read_joules(), time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        t1, e1 = time.perf_counter(), read_joules()
        ms, J = (t1 - t0) / n * 1e3, (e1 - e0) / n
        flops = (mult * S * S * 64 * B * H) if mult else 2.0 * B * S * 3072 * 12288
        print(f"{name:20s} {n:5d} launches  {ms:8.3f} ms  {J:8.3f} J/launch  {J / ms * 1e3:7.1f} W  {flops / ms / 1e9:7.1f} TF/s algorithmic  "
              f"{flops / J / 1e12:6.3f} TFLOP/J")
