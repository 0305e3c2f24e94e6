"""Scorer kernels at the reference's scale: 10 views x 518x518, cloud of 10*518*518 points (train/01_preference_pair.py
NUM_FRAMES=10; utils/projection_utils.py).  Prints GPU time per video, effective HBM rate, and the CPU oracle time for
one view for comparison."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import scorer  # noqa: E402

T, H, W = 10, 518, 518
N = T * H * W
g = torch.Generator(device="cuda").manual_seed(0)
pc = torch.randn(N, 3, generator=g, device="cuda") * torch.tensor([1.5, 1.5, 0.5], device="cuda") + torch.tensor([0, 0, 3.0], device="cuda")
colors = torch.rand(N, 3, generator=g, device="cuda") * 255
K = torch.tensor([[400.0, 0, W / 2], [0, 400.0, H / 2], [0, 0, 1]], device="cuda").repeat(T, 1, 1)
E = torch.eye(4, device="cuda").repeat(T, 1, 1)
for t in range(T):
    E[t, 0, 3] = 0.05 * t
gt = (torch.rand(T, H, W, 3, generator=g, device="cuda") * 255).to(torch.uint8)


def timeit(f, n=10):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        r = f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n, r


ms, rep = timeit(lambda: scorer.batch_reproject(pc, colors, K, E, H, W))
bytes_alg = T * N * 24 + T * H * W * (8 + 8 + 12)      # points+colours read per view; z-buffer write/read; fp32 frames out
print(f"batch_reproject  {N} points x {T} views: {ms:.3f} ms  ({T * N / ms / 1e6:.1f} Gpoint-views/s, {bytes_alg / ms / 1e6:.0f} GB/s algorithmic)")
m = scorer.MSEMetric()
ms2, _ = timeit(lambda: m.compute_device(gt=gt, rep=rep))
print(f"frame MSE (u8 THWC vs f32 TCHW, with range scan): {ms2:.3f} ms ({2 * (T * H * W * 3) * (1 + 4) / ms2 / 1e6:.0f} GB/s)")
ms3, _ = timeit(lambda: scorer.compute_motion_score_vectorized(E))
print(f"motion score: {ms3 * 1e3:.1f} us")
rng = np.random.default_rng(0)
p1 = [rng.random((2048, 2)).astype(np.float32) * 500 for _ in range(9)]
p2 = [p + rng.normal(size=p.shape).astype(np.float32) for p in p1]
ms4, _ = timeit(lambda: scorer.epipolar_errors(p1, p2), n=5)
print(f"8-point + Sampson, 9 frame pairs x 2048 matches (incl. host packing): {ms4:.3f} ms")
from oracle import scorer as osc  # noqa: E402  (comparison only)
t0 = time.time()
osc.project_points(pc.cpu().numpy(), colors.cpu().numpy(), K[0].cpu().numpy(), E[0].cpu().numpy(), H, W)
print(f"CPU oracle, ONE view: {(time.time() - t0) * 1e3:.0f} ms")
# the reference's own formulation (argsort + scatter) on the same GPU through torch, one view
def ref_style():
    R, tr = E[0, :3, :3], E[0, :3, 3]
    pp = (pc @ R.T + tr) @ K[0].T
    z = pp[:, 2]
    u = (pp[:, 0] / (z + 1e-8)).round().long()
    v = (pp[:, 1] / (z + 1e-8)).round().long()
    mk = (u >= 0) & (u < W) & (v >= 0) & (v < H) & (z > 0)
    u, v, z, c = u[mk], v[mk], z[mk], colors[mk]
    si = torch.argsort(z, descending=True)
    canvas = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    canvas[v[si], u[si]] = c[si].clamp(0, 255).to(torch.uint8)
    return canvas
ms5, _ = timeit(ref_style, n=5)
print(f"reference formulation (torch argsort + scatter) on this GPU, ONE view: {ms5:.3f} ms  -> {T * ms5:.1f} ms for {T} views")
