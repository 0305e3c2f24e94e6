"""Scorer kernels vs the reference goldens (tests/golden/scorer.pt) and the oracle.  -m gpu only."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import scorer as osc


@pytest.fixture(scope="module")
def sc():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from videogpa_amd import scorer
    return scorer


def test_project_points_bit_exact_vs_reference_golden(sc, golden_dir):
    gold = torch.load(os.path.join(golden_dir, "scorer.pt"), weights_only=False)
    for c in gold["project"]:
        canvas = sc.project_points(c["pc"], c["colors"], c["K"], c["E"], c["H"], c["W"])
        assert canvas.dtype == torch.uint8 and tuple(canvas.shape) == (c["H"], c["W"], 3)
        assert torch.equal(canvas.cpu(), c["canvas"]), f"{(canvas.cpu() != c['canvas']).any(-1).sum().item()} pixels differ"


def test_batch_reproject_large_cloud_vs_oracle(sc):
    """Reference-sized view (518x518), many collisions; bit-exact incl. the [-1,1] float frames; empty frame."""
    rng = np.random.default_rng(0)
    N, T, H, W = 400_000, 3, 518, 518
    pc = (rng.normal(size=(N, 3)) * [1.0, 1.0, 0.4] + [0, 0, 3.0]).astype(np.float32)
    colors = (rng.random((N, 3)) * 255).astype(np.float32)
    K = np.stack([np.array([[300.0 + 10 * t, 0, W / 2], [0, 310.0, H / 2], [0, 0, 1]], np.float32) for t in range(T)])
    E = np.stack([np.eye(4, dtype=np.float32) for _ in range(T)])
    for t in range(T):
        a = 0.05 * t
        E[t, :3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
        E[t, :3, 3] = [0.1 * t, 0, 0.05 * t]
    E[2, 2, 3] = -100.0   # third camera sees nothing -> black frame
    out = sc.batch_reproject(pc, colors, K, E, H, W)
    ref = osc.batch_reproject(pc, colors, K, E, H, W)
    assert out.shape == (T, 3, H, W)
    assert np.array_equal(out.cpu().numpy(), ref)
    assert float(out[2].max()) == -1.0
    out34 = sc.batch_reproject(pc, colors, K, E[:, :3], H, W)       # [T,3,4] extrinsics
    assert torch.equal(out, out34)


def test_confidence_filter_fused(sc):
    g = torch.Generator().manual_seed(4)
    T, h, w = 2, 24, 31
    pts = torch.randn(T, h, w, 3, generator=g) * torch.tensor([1.0, 1.0, 0.3]) + torch.tensor([0, 0, 3.0])
    conf = torch.rand(T, h, w, generator=g)
    conf[0, 0, :5] = float("nan")
    conf[1, 3, :7] = 0.0
    imgs = torch.rand(T, 3, h, w, generator=g)
    K = torch.tensor([[[30.0, 0, w / 2], [0, 30.0, h / 2], [0, 0, 1]]]).repeat(T, 1, 1)
    E = torch.eye(4)[None].repeat(T, 1, 1)
    for thr in (0.0, 40.0):
        out = sc.reproject_predictions(pts, conf, imgs, K, E, h, w, conf_thres=thr)
        v, c = osc.pointcloud_filter(pts, conf, imgs, thr)
        ref = osc.batch_reproject(v.numpy(), c.numpy(), K.numpy(), E.numpy(), h, w)
        assert np.array_equal(out.cpu().numpy(), ref), thr


def test_motion_and_mse_vs_reference_golden(sc, golden_dir):
    gold = torch.load(os.path.join(golden_dir, "scorer.pt"), weights_only=False)
    for c in gold["motion"]:
        got = float(sc.compute_motion_score_vectorized(c["E"]))
        assert abs(got - c["score"]) <= 2e-6 * max(1.0, abs(c["score"])), (got, c["score"])   # fp32 sum order / acosf ulp
    assert float(sc.compute_motion_score_vectorized(torch.eye(4)[None])) == 0.0                # single frame: NaN -> 0
    m = sc.MSEMetric()
    for c in gold["mse"]:
        got = m.compute(gt=c["gt"], rep=c["rep"])
        assert abs(got - c["val"]) <= 2e-6 * c["val"], (got, c["val"])


def test_epipolar_vs_oracle(sc):
    rng = np.random.default_rng(1)
    p1s, p2s = [], []
    for i in range(4):
        n = [64, 300, 2048, 5][i]
        X = rng.normal(size=(n, 3)) + np.array([0, 0, 5.0])
        K = np.array([[400.0, 0, 160], [0, 400.0, 120], [0, 0, 1]])
        a = 0.05 * (i + 1)
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        t = np.array([0.3, 0.05 * i, 0.1])
        a1 = (K @ X.T).T
        a2 = (K @ (R @ X.T + t[:, None])).T
        noise = 0.0 if i == 0 else 0.5
        p1s.append((a1[:, :2] / a1[:, 2:] + noise * rng.normal(size=(n, 2))).astype(np.float32))
        p2s.append((a2[:, :2] / a2[:, 2:] + noise * rng.normal(size=(n, 2))).astype(np.float32))
    err, Fm = sc.epipolar_errors(p1s, p2s, return_F=True)
    err = err.cpu().numpy()
    assert err[3] == -1.0                                   # < 8 matches
    for i in range(3):
        ref = osc.epipolar_pair_error(p1s[i], p2s[i])
        assert abs(err[i] - ref) <= 1e-4 * max(ref, 1e-3), (i, err[i], ref)
        Fr = osc.find_fundamental(p1s[i], p2s[i])
        assert np.allclose(Fm[i].cpu().numpy(), Fr, rtol=2e-3, atol=1e-6)
    assert err[0] < 2e-4                                    # exact correspondences: sqrt(0 + 1e-8)
    em = sc.EpipolarMetric(min_matches=20)
    assert em.compute_from_matches([p1s[3]], [p2s[3]]) == -1.0
    v = em.compute_from_matches(p1s, p2s)
    assert abs(v - np.mean([osc.epipolar_pair_error(p1s[i], p2s[i]) for i in range(3)])) < 1e-4
