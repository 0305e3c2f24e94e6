"""Stand-alone timings of two cfg5 row kernels per SHAPE, on cache-cold operands (every call takes the next of several inputs): `wan_rms_rope_fwd` at the
self-attention rows (36 960 x 3072 out of the fused [rows, 9216] projection output) and at the cross-attention key rows (1024 x 3072), `gelu_tanh_fwd_q8` /
`_bwd_q8` at 36 960 x 14 336.  bench.py's per-kernel `frac` averages over all launches of a name, small ones included."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import ops  # noqa: E402
from videogpa_amd.wan_model import _rms_rope_fwd_raw, rope_tables  # noqa: E402


def t(f, n=24):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


D, L = 3072, 18480
cos, sin = rope_tables((21, 22, 40), 128, "cuda")
w = torch.ones(D, device="cuda").bfloat16()
turn = [0]


def nxt(xs):
    turn[0] += 1
    return xs[turn[0] % len(xs)]


qkv = [torch.randn(2 * L, 3 * D, device="cuda").bfloat16() for _ in range(3)]
out = torch.empty(2 * L, D, dtype=torch.bfloat16, device="cuda")
rstd = torch.empty(2 * L, dtype=torch.float32, device="cuda")
us = t(lambda: _rms_rope_fwd_raw(nxt(qkv)[:, :D], 3 * D, w, cos, sin, L, 128, 1e-6, out, D, rstd))
print(f"wan_rms_rope_fwd  36960 x 3072 (strided q slice, RoPE)   {us:7.1f} us  {2 * L * D * 4 / us / 1e6:5.2f} TB/s")
kc = [torch.randn(1024, D, device="cuda").bfloat16() for _ in range(8)]
ok, rk = torch.empty(1024, D, dtype=torch.bfloat16, device="cuda"), torch.empty(1024, dtype=torch.float32, device="cuda")
us = t(lambda: _rms_rope_fwd_raw(nxt(kc), D, w, None, None, 512, 128, 1e-6, ok, D, rk))
print(f"wan_rms_rope_fwd   1024 x 3072 (cross-attention keys)    {us:7.1f} us  {1024 * D * 4 / us / 1e6:5.2f} TB/s")
del qkv
u = [torch.randn(2 * L, 14336, device="cuda").bfloat16() for _ in range(2)]
us = t(lambda: ops.gelu_tanh_fwd_q8(nxt(u)))
print(f"gelu_tanh_fwd_q8  36960 x 14336                          {us:7.1f} us  {2 * L * 14336 * 3 / us / 1e6:5.2f} TB/s")
us = t(lambda: ops.gelu_tanh_bwd_q8(nxt(u), nxt(u)))
print(f"gelu_tanh_bwd_q8  36960 x 14336                          {us:7.1f} us  {2 * L * 14336 * 5 / us / 1e6:5.2f} TB/s")

from videogpa_amd.wan_model import _rms_rope_bwd_raw  # noqa: E402

qkv2 = [torch.randn(2 * L, 3 * D, device="cuda").bfloat16() for _ in range(3)]
dq = [torch.randn(2 * L, D, device="cuda").bfloat16() for _ in range(3)]
dqkv = torch.empty(2 * L, 3 * D, dtype=torch.bfloat16, device="cuda")
us = t(lambda: _rms_rope_bwd_raw(nxt(dq), D, nxt(qkv2)[:, :D], 3 * D, rstd, w, cos, sin, L, 128, dqkv[:, :D], 3 * D))
print(f"wan_rms_rope_bwd  36960 x 3072 (strided)                   {us:7.1f} us  {2 * L * D * 6 / us / 1e6:5.2f} TB/s")
del qkv2, dq, dqkv

# LayerNorm + modulation row kernels (fp32 residual stream, bf16 branch outputs) at 36 960 x 3072, forward and backward, through the autograd nodes of the model
from videogpa_amd.wan_model import gate_ln, ln_mod  # noqa: E402

rows = 2 * L
xs = [torch.randn(rows, D, device="cuda") for _ in range(4)]
ys = [torch.randn(rows, D, device="cuda").bfloat16() for _ in range(4)]
gid = (torch.arange(rows, device="cuda", dtype=torch.int32) % 4).contiguous()
tab = torch.randn(4, 6, D, device="cuda")
lnw, lnb = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
us = t(lambda: ln_mod(nxt(xs), gid, None, None, tab[:, 0], tab[:, 1], 1e-6))
print(f"wan_ln_mod_fwd    36960 x 3072 (fp32 -> bf16)              {us:7.1f} us  {rows * D * 6 / us / 1e6:5.2f} TB/s")
us = t(lambda: gate_ln(nxt(xs), nxt(ys), gid, tab[:, 2], lnw, lnb, None, None, 1e-6))
print(f"wan_gate_ln_fwd   36960 x 3072 (x + y gate -> x', LN -> bf16) {us:7.1f} us  {rows * D * 12 / us / 1e6:5.2f} TB/s")
xg = [x_.clone().requires_grad_(True) for x_ in xs[:2]]
yg = [y_.clone().requires_grad_(True) for y_ in ys[:2]]
outs = [gate_ln(a, b, gid, tab[:, 2], lnw, lnb, None, None, 1e-6) for a, b in zip(xg, yg)]
dh = [torch.randn(rows, D, device="cuda").bfloat16() for _ in range(2)]
dres = [torch.randn(rows, D, device="cuda") for _ in range(2)]


def bwd():
    i = turn[0] = turn[0] + 1
    h, xo = outs[i % 2]
    torch.autograd.grad([h, xo], [xg[i % 2], yg[i % 2]], [dh[i % 2], dres[i % 2]], retain_graph=True)


us = t(bwd)
print(f"wan_ln_gate_bwd   36960 x 3072 (dh, dres -> dx, dy)        {us:7.1f} us  {rows * D * 16 / us / 1e6:5.2f} TB/s")
