"""The geometry scorer at the reference's scale (10 views x 518 x 518, a cloud of 10 * 518 * 518 points: train/01_preference_pair.py:33-34, utils/projection_utils.py):
the SAME measurement bench.py attaches as its `scorer` block (bench.scorer_report: ms per video, point-views / s, algorithmic GB/s against the 8 TB/s peak, atomics / s,
the reference's argsort + scatter formulation on this GPU, the CPU oracle), plus the kernels that block does not touch (confidence cut, MVCS, 8-point + Sampson) so that
a rocprofv3 pass over this script (tools/profile_round.sh) sees every scorer kernel.
    python tools/scorer_bench.py [--quick] [--json gpurun_out/scorer_bench.json]      # --quick: two launches of everything, no timing loops (PMC passes)"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from videogpa_amd import scorer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
ap.add_argument("--json", default=os.path.join(ROOT, "gpurun_out", "scorer_bench.json"))
a = ap.parse_args()
dev = torch.device("cuda", 0)


def timeit(f, n=10):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


out = {}
if a.quick:
    for pm in (True, False):
        T, H, W, N, pc, colors, K, E, gt = bench.scorer_inputs(dev, pm)
        for _ in range(2):
            rep = scorer.batch_reproject(pc, colors, K, E, H, W)
        scorer.MSEMetric().compute_device(gt=gt, rep=rep)
else:
    out = bench.scorer_report(dev, with_cpu=True)
    for kind in ("pointmap", "random_cloud"):
        r = out[kind]
        print(f"batch_reproject [{kind:12s}] {r['ms_per_video']:.3f} ms per video  {r['gpoint_views_per_s']:.1f} Gpoint-views/s  {r['algorithmic_gbs']:.0f} GB/s algorithmic "
              f"= {r['frac_of_hbm_peak']:.3f} of 8 TB/s  {r['atomics_per_s'] / 1e9:.1f} G atomics/s  ({100 * r['pixels_covered']:.0f} % of the pixels hit)")
    r = out["pointmap"]
    print(f"frame MSE {r['mse_ms']:.3f} ms = {r['mse_gbs']:.0f} GB/s; motion score {r['motion_score_us']:.1f} us; reference formulation (torch argsort + scatter) "
          f"{r['torch_argsort_ms_same_gpu']:.1f} ms per video; CPU oracle {r.get('cpu_oracle_ms', float('nan')):.0f} ms per video")
# the kernels the bench block does not touch: confidence cut (radix select), MVCS, 8-point + Sampson
T, H, W, N, pc, colors, K, E, gt = bench.scorer_inputs(dev, True)
g = torch.Generator(device=dev).manual_seed(1)
conf = torch.rand(N, generator=g, device=dev) * 10
depth = 3.0 + 0.2 * torch.rand(T, H, W, generator=g, device=dev)
rng = np.random.default_rng(0)
p1 = [rng.random((2048, 2)).astype(np.float32) * 500 for _ in range(9)]
p2 = [p + rng.normal(size=p.shape).astype(np.float32) for p in p1]
n = 2 if a.quick else 10
ms_c, _ = timeit(lambda: scorer.confidence_threshold(conf, 50.0), n)
ms_p, _ = timeit(lambda: scorer.reproject_predictions(pc.view(T, H, W, 3), conf.view(T, H, W), colors.view(T, H, W, 3) / 255, K, E, H, W, conf_thres=50.0), n)
ms_m, _ = timeit(lambda: scorer.MVCSMetric().compute_device(depths=depth, intrinsics=K, extrinsics=E[:, :3]), n)
ms_e, _ = timeit(lambda: scorer.epipolar_errors(p1, p2), max(2, n // 2))
out["other_kernels"] = {"conf_threshold_ms": ms_c, "conf_threshold_gbs": N * 4 * 2 / ms_c / 1e6, "reproject_predictions_conf50_ms": ms_p, "mvcs_ms": ms_m,
                        "mvcs_gbs": (T - 1) * H * W * 8 / ms_m / 1e6, "epipolar_9x2048_ms_incl_host_packing": ms_e}
print(f"confidence cut (radix select over {N} values): {ms_c:.3f} ms; fused filter + reproject at conf_thres 50: {ms_p:.3f} ms; MVCS: {ms_m:.3f} ms; "
      f"8-point + Sampson, 9 pairs x 2048 matches: {ms_e:.3f} ms")
if not a.quick:
    os.makedirs(os.path.dirname(a.json), exist_ok=True)
    with open(a.json, "w") as f:
        json.dump(out, f, indent=1)
