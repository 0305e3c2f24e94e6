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


# ---------------------------------------------------------------- scorer2: filter select, PSNR / resized MSE, MVCS, pose decode, DA3 unprojection, VideoProcessor
def _g2(golden_dir):
    return torch.load(os.path.join(golden_dir, "scorer2.pt"), weights_only=False)


def test_get_colored_pointcloud_bit_exact_vs_reference_golden(sc, golden_dir):
    """On-device radix select of the k-th largest confidence (no torch.topk / .item()): same kept set as the reference."""
    for c in _g2(golden_dir)["pointcloud"]:
        preds = {"images": c["images"]}
        if c["mode"] == "pointmap":
            preds.update(world_points=c["points"], world_points_conf=c["conf"])
        else:
            preds.update(world_points_from_depth=c["points"], depth_conf=c["conf"])
        v, col = sc.get_colored_pointcloud(preds, mode=c["mode"], conf_thres=c["conf_thres"])
        assert torch.equal(v.cpu(), c["vertices"]) and torch.equal(col.cpu(), c["colors"]), (c["mode"], c["conf_thres"])


def test_confidence_threshold_fullsize_vs_topk(sc):
    """Reference scale: 10 x 518 x 518 confidences; the device threshold equals torch.topk's k-th value for several cuts."""
    g = torch.Generator(device="cuda").manual_seed(3)
    conf = torch.rand(10 * 518 * 518, generator=g, device="cuda") * 20
    conf[::1001] = float("nan")
    conf[5::777] = 0.0
    conf[7::555] = conf[11]                                    # ties
    valid = torch.isfinite(conf) & (conf > 1e-5)
    n = int(valid.sum())
    for thres in (0.5, 25.0, 50.0, 99.9, 100.0):
        k = max(1, int(np.ceil(n * max(0.0, min(1.0, 1.0 - thres / 100.0)))))
        ref = torch.topk(conf[valid], k)[0][-1]
        got = sc.confidence_threshold(conf, thres)
        assert got.item() == ref.item(), (thres, got.item(), ref.item())
    assert sc.confidence_threshold(conf, 0.0).item() == float("-inf")
    assert sc.confidence_threshold(torch.full((100,), float("nan")), 30.0).item() == float("-inf")


def test_psnr_resized_mse_mvcs_vs_reference_golden(sc, golden_dir):
    g = _g2(golden_dir)
    ps, ms, mv = sc.PSNRMetric(), sc.MSEMetric(), sc.MVCSMetric()
    for c in g["psnr"]:
        got = ps.compute(gt=c["gt"], rep=c["rep"])
        assert abs(got - c["val"]) <= 2e-5 * max(1.0, abs(c["val"])), (got, c["val"])          # fp32 sum order + log10f
    for c in g["mse_resize"]:
        got = ms.compute(gt=c["gt"], rep=c["rep"])
        assert abs(got - c["val"]) <= 5e-6 * max(1.0, abs(c["val"])), (got, c["val"])
    for c in g["mvcs"]:
        d = c["depths"].numpy() if c["depths_is_numpy"] else c["depths"]
        got = mv.compute(gt=None, rep=None, depths=d, intrinsics=c["intrinsics"], extrinsics=c["extrinsics"])
        assert abs(got - c["val"]) <= 1e-5 * max(1.0, abs(c["val"])), (got, c["val"])          # fp32 3x3 inverse / pose chain
    assert mv.compute(depths=g["mvcs"][-1]["depths"], intrinsics=g["mvcs"][-1]["intrinsics"], extrinsics=g["mvcs"][-1]["extrinsics"]) == 0.0


def test_lpips_and_consistency_wrappers_vs_reference_golden(sc, golden_dir):
    """LPIPSMetric / Consistency_Score (metrics/lpips.py:21-63, metrics/consistency_score.py:52-72) with the stand-in perceptual net the
    goldens were made with: the device-side normalisation / layout / resize in front of the caller's network, and the combination."""
    g = _g2(golden_dir)
    net = lambda a, b: (a - b).abs().mean(dim=(1, 2, 3), keepdim=True) + 0.01 * a.mean(dim=(1, 2, 3), keepdim=True)
    lp = sc.LPIPSMetric(device="cuda", lpips_net=net)
    cs = sc.Consistency_Score(net, device="cuda")                # positional, as replicate_scorer.py:68 constructs it
    for c in g["lpips"]:
        got = lp.compute(gt=c["gt"], rep=c["rep"])
        assert abs(got - c["val"]) <= 5e-6 * max(1.0, abs(c["val"])), (got, c["val"])
    for c in g["consistency"]:
        s_, m_ = cs.compute(gt=c["gt"], rep=c["rep"], extrinsics=c["extrinsics"], ratio=c["ratio"])
        assert abs(s_ - c["score"]) <= 5e-6 * max(1.0, abs(c["score"])), (s_, c["score"])
        assert abs(m_ - c["motion"]) <= 2e-6 * max(1.0, abs(c["motion"]))
    pm = sc.frames_to_pm1(g["lpips"][0]["gt"])
    assert pm.shape == (3, 3, 14, 18) and float(pm.min()) >= -1.0 and float(pm.max()) <= 1.0
    with pytest.raises(RuntimeError, match="third-party"):
        sc.LPIPSMetric().compute(gt=g["lpips"][0]["gt"], rep=g["lpips"][0]["rep"])
    assert sc.Consistency_Score(None).compute(gt=g["lpips"][1]["gt"], rep=g["lpips"][1]["rep"], extrinsics=g["consistency"][0]["extrinsics"], ratio=0)[0] > 0
    with pytest.raises(ValueError):
        sc.EpipolarMetric(descriptor_type="orb")


def test_pose_decode_and_unprojection_vs_reference_golden(sc, golden_dir):
    g = _g2(golden_dir)
    for c in g["pose_enc"]:
        ext, K = sc.pose_encoding_to_extri_intri(c["pose_enc"], c["image_size_hw"])
        assert torch.allclose(ext.cpu(), c["extrinsics"], rtol=1e-6, atol=1e-6)
        assert torch.allclose(K.cpu(), c["intrinsics"], rtol=2e-6, atol=1e-4)                     # tanf ulp on ~700 px focal lengths
        ext2, none = sc.pose_encoding_to_extri_intri(c["pose_enc"], build_intrinsics=False)
        assert none is None and torch.equal(ext2, ext)
    for c in g["da3_unproject"]:
        wp = sc.unproject_depth_to_world(c["depths"], c["intrinsics"], c["extrinsics"])
        assert torch.allclose(wp.cpu(), c["world_points"], rtol=1e-5, atol=2e-6)


def test_video_processor_dispatch_matches_oracle_chain(sc):
    """VideoProcessor.process (pipelines/process_video.py:61-196) with a stand-in backbone: both backbones' glue, the
    per-threshold loop, the metric dispatch by name -- against the oracle run step by step on the CPU."""
    from types import SimpleNamespace
    from videogpa_amd.process_video import VideoProcessor
    rng = np.random.default_rng(5)
    T, H, W = 4, 28, 36
    frames = (rng.random((T, H, W, 3)) * 255).astype(np.uint8)
    K = np.stack([np.array([[40.0 + t, 0, W / 2], [0, 42.0, H / 2], [0, 0, 1]], np.float32) for t in range(T)])
    E = np.stack([np.eye(4, dtype=np.float32) for _ in range(T)])
    for t in range(T):
        a = 0.03 * t
        E[t, :3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
        E[t, :3, 3] = [0.05 * t, 0.0, 0.02 * t]
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    depth = np.stack([2.0 + 0.3 * np.sin(xs / W * 3 + 0.2 * t) + 0.01 * rng.normal(size=(H, W)) for t in range(T)]).astype(np.float32)
    conf = (rng.random((T, H, W)) * 5).astype(np.float32)
    conf[0, 0, :4] = np.nan

    class _Const(sc.Metric):
        def __init__(self):
            super().__init__("const")

        def compute(self, *, gt, rep, **kw):
            return 7.0

    metrics = {"Consistency_Score": sc.Consistency_Score(lpips_net=lambda gt, rep: 0.25), "MVCS": sc.MVCSMetric(), "mse": sc.MSEMetric(),
               "other": _Const()}
    world = osc.unproject_depth(depth, K, osc.affine_inverse(E)).numpy()
    images01 = torch.from_numpy(frames).float().div(255).permute(0, 3, 1, 2).contiguous()

    def oracle_results(gt):
        out = {}
        for th in (0, 40):
            v, c = osc.pointcloud_filter(world, conf, images01, th)
            rep = osc.batch_reproject(v.numpy(), c.numpy(), K, E[:, :3], H, W)
            mse_v = osc.mse(gt, torch.from_numpy(rep))
            out[th] = {"Consistency_Score": mse_v + 0.25, "motion_norm": osc.motion_score(E[:, :3]), "MVCS": osc.mvcs(depth, K, E[:, :3]),
                       "mse": mse_v, "other": 7.0}
        return out

    # DA3 branch: prediction object -> unprojection on device
    da3 = VideoProcessor(metrics, backbone_fn=lambda fl: SimpleNamespace(processed_images=frames, extrinsics=E[:, :3], intrinsics=K, depth=depth, conf=conf),
                         frame_sampler=lambda p, n: frames, backbone="da3")
    res = da3.process("video.mp4", thresholds=[0, 40], num_frames=T)
    ref = oracle_results(images01)
    assert res["_extrinsic"] == E[:, :3].tolist()
    for th in (0, 40):
        assert set(res[th]) == set(ref[th])
        for k in ref[th]:
            assert abs(res[th][k] - ref[th][k]) <= 2e-5 * max(1.0, abs(ref[th][k])), (th, k, res[th][k], ref[th][k])

    # VGGT branch: dict predictions (gt = the sampled uint8 frames, as _process_vggt passes them on)
    preds = {"images": images01, "world_points_from_depth": torch.from_numpy(world), "depth_conf": torch.from_numpy(conf),
             "extrinsic": torch.from_numpy(E[:, :3]), "intrinsic": torch.from_numpy(K), "depth": torch.from_numpy(depth)}
    vg = VideoProcessor(metrics, backbone_fn=lambda fr: preds, backbone="vggt")
    res2 = vg.process(frames, thresholds=[0, 40], num_frames=T)
    ref2 = oracle_results(frames)
    for th in (0, 40):
        for k in ref2[th]:
            assert abs(res2[th][k] - ref2[th][k]) <= 2e-5 * max(1.0, abs(ref2[th][k])), (th, k, res2[th][k], ref2[th][k])
    assert VideoProcessor._resolve_backbone(None, "depth-anything/DA3-Large") == "da3" and VideoProcessor._resolve_backbone(None, None) == "vggt"
    with pytest.raises(RuntimeError, match="frame_sampler"):
        vg.process("x.mp4", [0], T)
