"""On-device geometry-consistency scorer: drop-ins for the reference's `batch_reproject` / `project_points`
(utils/projection_utils.py), `get_colored_pointcloud` (utils/pointcloud_utils.py), `Metric` / `MSEMetric`
(metrics/base.py, metrics/mse.py), `compute_motion_score_vectorized` / `Consistency_Score`
(metrics/consistency_score.py) and the geometry half of `EpipolarMetric` (metrics/epipolar.py:161-213).
Kernels: videogpa_amd/csrc/scorer.hip.  Third-party networks (LPIPS-VGG, SIFT / LightGlue matchers, VGGT / DA3
backbones) are outside the hot path: they are passed in as callables / precomputed matches."""
from abc import ABC, abstractmethod
from typing import Any

import numpy as np
import torch

from . import _lib
from .ops import _req, _stream


def _dev_f32(x, device="cuda"):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(device=device, dtype=torch.float32).contiguous()


def project_views(pc, colors, intrinsics, extrinsics, H, W, conf=None, conf_thr=float("-inf"), want_canvas=True, want_float=True):
    """All T views in one launch pair.  -> (canvas u8 [T,H,W,3] | None, frames fp32 [T,3,H,W] in [-1,1] | None).
    conf_thr: python float, or a device fp32 tensor of one element (from `confidence_threshold`: stays on the GPU)."""
    pc, colors = _dev_f32(pc).reshape(-1, 3), _dev_f32(colors).reshape(-1, 3)
    K, E = _dev_f32(intrinsics), _dev_f32(extrinsics)
    if K.dim() == 2:
        K, E = K[None], E[None]
    T = E.shape[0]
    e_rows = E.shape[1]
    N = pc.shape[0]
    dev = pc.device
    canvas = torch.empty(T, H, W, 3, dtype=torch.uint8, device=dev) if want_canvas else None
    out_f = torch.empty(T, 3, H, W, dtype=torch.float32, device=dev) if want_float else None
    if T == 0:
        return canvas, out_f
    cf = None if conf is None else _dev_f32(conf).reshape(-1)
    ws_bytes = _lib.query("vgpa_project_points_workspace_bytes", T, H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    thr_dev = conf_thr if torch.is_tensor(conf_thr) else None
    if thr_dev is not None:
        _req(thr_dev, torch.float32)
    _lib.call("vgpa_project_points", pc if N else None, colors if N else None, cf, 0.0 if thr_dev is not None else float(conf_thr), thr_dev,
              K, E, e_rows, N, T, H, W, canvas, out_f, ws, ws_bytes, _stream())
    return canvas, out_f


def project_points(pc, colors, K, E, H, W, bg=(0, 0, 0)):
    """Single view -> uint8 [H,W,3] (utils/projection_utils.py:12-51).  Only bg = (0,0,0) is used by the reference."""
    if tuple(bg) != (0, 0, 0):
        raise NotImplementedError("non-black background")
    return project_views(pc, colors, K, E[:3] if E.shape[0] == 4 else E, H, W, want_float=False)[0][0]


def batch_reproject(pc, colors, intrinsics, extrinsics, H, W, save_path=None):
    """-> fp32 [T,3,H,W] in [-1,1] (utils/projection_utils.py:57-101).  PNG dumping (`save_path`) is host I/O and not
    part of this path."""
    if save_path is not None:
        raise NotImplementedError("save_path (PNG dump through cv2) is outside the on-device path")
    if len(extrinsics) == 0:
        return torch.zeros((0, 3, H, W), device="cuda", dtype=torch.float32)
    return project_views(pc, colors, intrinsics, extrinsics, H, W, want_canvas=False)[1]


def confidence_threshold(conf, conf_thres):
    """The top-(100 - conf_thres) % confidence cut of utils/pointcloud_utils.py:55-73 -> device fp32 tensor [1] holding the
    k-th largest valid confidence (-inf when nothing is cut).  On-device radix select: no sort, no host round trip."""
    vals = _dev_f32(conf).reshape(-1)
    thr = torch.empty(1, dtype=torch.float32, device=vals.device)
    if vals.numel() == 0:
        return thr.fill_(float("-inf"))
    ws_bytes = _lib.query("vgpa_conf_threshold_workspace_bytes")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=vals.device)
    _lib.call("vgpa_conf_threshold", vals, vals.numel(), float(conf_thres), thr, ws, ws_bytes, _stream())
    return thr


def get_colored_pointcloud(predictions, mode="pointmap", conf_thres=50):
    """Drop-in for utils/pointcloud_utils.py:10-80 -> (vertices [N,3], colors [N,3] in 0..255), compacted on the device.
    (`reproject_predictions` below is the fused form that never materialises the filtered cloud.)"""
    if "pointmap" in mode.lower() and "world_points" in predictions:
        points, conf = predictions["world_points"], predictions.get("world_points_conf")
    else:
        points, conf = predictions["world_points_from_depth"], predictions.get("depth_conf")
    points = _dev_f32(points)
    conf = torch.ones_like(points[..., 0]) if conf is None else _dev_f32(conf)
    images = _dev_f32(predictions["images"])
    colors = (images.permute(0, 2, 3, 1) if (images.dim() == 4 and images.shape[1] == 3) else images).reshape(-1, 3) * 255
    vals = conf.reshape(-1)
    mask = torch.isfinite(vals) & (vals > 1e-5)
    if conf_thres > 0:
        mask = mask & (vals >= confidence_threshold(vals, conf_thres))
    return points.reshape(-1, 3)[mask], colors[mask]


def reproject_predictions(points, conf, images, intrinsics, extrinsics, H, W, conf_thres=0.0):
    """get_colored_pointcloud + batch_reproject without materialising the filtered cloud: the validity / confidence
    predicate runs inside the splat kernel.  points [T,h,w,3], conf [T,h,w], images [T,3,h,w] or [T,h,w,3] in [0,1]."""
    images = _dev_f32(images)
    colors = (images.permute(0, 2, 3, 1) if (images.dim() == 4 and images.shape[1] == 3) else images).reshape(-1, 3) * 255
    thr = confidence_threshold(conf, conf_thres)
    return project_views(_dev_f32(points).reshape(-1, 3), colors, intrinsics, extrinsics, H, W, conf=conf, conf_thr=thr, want_canvas=False)[1]


def compute_motion_score_vectorized(extrinsics, device="cuda"):
    E = _dev_f32(extrinsics, device)
    out = torch.empty(1, dtype=torch.float32, device=E.device)
    _lib.call("vgpa_motion_score", E, E.shape[1], E.shape[0], out, _stream())
    return out[0]


class Metric(ABC):
    def __init__(self, name: str):
        self.name = name

    @abstractmethod
    def compute(self, *, gt, rep, **kwargs) -> float:
        raise NotImplementedError

    def __call__(self, *args: Any, **kwargs: Any) -> float:
        return self.compute(*args, **kwargs)


def _img_desc(x):
    """-> (tensor on device as u8 or f32, dtype code, layout code, is_tensor, T, C, H, W)"""
    is_tensor = isinstance(x, torch.Tensor)
    t = x if is_tensor else torch.from_numpy(np.ascontiguousarray(x))
    if t.dim() == 3:
        t = t.unsqueeze(0)
    t = t.cuda()
    if t.dtype != torch.uint8:
        t = t.float()
    t = t.contiguous()
    if t.shape[-1] == 3:
        T, H, W, C = t.shape
        layout = 1
    else:
        T, C, H, W = t.shape
        layout = 0
    return t, (2 if t.dtype == torch.uint8 else 0), layout, int(is_tensor), T, C, H, W


def _frame_metric(gt, rep, psnr):
    g, gd, gl, gt_t, T, C, H, W = _img_desc(gt)
    r, rd, rl, rt_t, T2, C2, H2, W2 = _img_desc(rep)
    if (T, C) != (T2, C2):
        raise ValueError("gt and rep disagree in frames / channels")
    out = torch.empty(1, dtype=torch.float32, device=g.device)
    ws_bytes = _lib.query("vgpa_frame_metric_workspace_bytes")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=g.device)
    # rep of another spatial size is resized bilinearly to gt's inside the kernel (metrics/mse.py:24-25,65-66)
    _lib.call("vgpa_frame_metric", g, gd, gl, gt_t, r, rd, rl, rt_t, T, C, H, W, H2, W2, 1 if psnr else 0, out, ws, ws_bytes, _stream())
    return out[0]


class MSEMetric(Metric):
    def __init__(self):
        super().__init__(name="mse")

    def compute_device(self, *, gt, rep):
        return _frame_metric(gt, rep, psnr=False)

    def compute(self, *, gt, rep, **kwargs) -> float:
        return float(self.compute_device(gt=gt, rep=rep).item())


class PSNRMetric(Metric):
    """metrics/mse.py:56-80: 10 log10(1 / mse) on [0,1] frames, 100.0 for identical inputs."""

    def __init__(self, device="cuda"):
        super().__init__(name="psnr")
        self.device = device

    def compute(self, *, gt, rep, **kwargs) -> float:
        return float(_frame_metric(gt, rep, psnr=True).item())


class MVCSMetric(Metric):
    """Multi-view depth-consistency score (metrics/mvcs.py:12-114): every pixel of view i is back-projected with its depth,
    moved into view i+1 and compared with the depth sampled there; exp(-mean masked squared error over the pairs)."""

    def __init__(self, device="cuda"):
        super().__init__(name="MVCS")
        self.device = device

    def compute_device(self, *, depths, intrinsics, extrinsics):
        d = _dev_f32(depths)
        if d.dim() == 4:
            d = d.squeeze(1) if d.shape[1] == 1 else (d.squeeze(3) if d.shape[3] == 1 else d)
        d = d.contiguous()
        K, E = _dev_f32(intrinsics), _dev_f32(extrinsics)
        T, H, W = d.shape
        out = torch.empty(1, dtype=torch.float32, device=d.device)
        ws_bytes = _lib.query("vgpa_mvcs_workspace_bytes", T)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=d.device)
        _lib.call("vgpa_mvcs", d, K, K.shape[-1], E, E.shape[-2], T, H, W, out, ws, ws_bytes, _stream())
        return out[0]

    def compute(self, *, gt=None, rep=None, depths, intrinsics, extrinsics, **kwargs) -> float:
        return float(self.compute_device(depths=depths, intrinsics=intrinsics, extrinsics=extrinsics).item())


def pose_encoding_to_extri_intri(pose_encoding, image_size_hw=None, pose_encoding_type="absT_quaR_FoV", build_intrinsics=True):
    """vggt/utils/pose_enc.py:62-124 on device: [...,9] -> ([...,3,4], [...,3,3] | None)."""
    if pose_encoding_type != "absT_quaR_FoV":
        raise NotImplementedError
    pe = _dev_f32(pose_encoding)
    lead = pe.shape[:-1]
    n = pe.numel() // 9
    ext = torch.empty(*lead, 3, 4, dtype=torch.float32, device=pe.device)
    intr = torch.empty(*lead, 3, 3, dtype=torch.float32, device=pe.device) if build_intrinsics else None
    Hh, Ww = image_size_hw if build_intrinsics else (0, 0)
    _lib.call("vgpa_pose_decode", pe, n, float(Hh), float(Ww), ext, intr, _stream())
    return ext, intr


def unproject_depth_to_world(depths, intrinsics, extrinsics):
    """DA3 branch of the reference (pipelines/process_video.py:151-156): c2w = affine_inverse(w2c), world points of every
    pixel = c2w [K^-1 (x, y, 1) depth].  depths [T,H,W], K [T,3,3], E [T,3|4,4] -> [T,H,W,3]."""
    d, K, E = _dev_f32(depths), _dev_f32(intrinsics), _dev_f32(extrinsics)
    T, H, W = d.shape
    world = torch.empty(T, H, W, 3, dtype=torch.float32, device=d.device)
    _lib.call("vgpa_unproject_depth", d, K, E, E.shape[-2], T, H, W, world, _stream())
    return world


def frames_to_pm1(x, size=None):
    """LPIPSMetric._to_tensor_neg1_pos1 (metrics/lpips.py:38-63) on device: any frame container -> fp32 [T,C,H,W] in [-1,1];
    `size` = (H, W) additionally resizes bilinearly (align_corners=False), as :31-32 do for `rep`."""
    t, dt, layout, is_tensor, T, C, H, W = _img_desc(x)
    Ho, Wo = (H, W) if size is None else size
    out = torch.empty(T, C, Ho, Wo, dtype=torch.float32, device=t.device)
    ws_bytes = _lib.query("vgpa_frame_metric_workspace_bytes")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=t.device)
    _lib.call("vgpa_frames_to_pm1", t, dt, layout, is_tensor, T, C, H, W, Ho, Wo, out, ws, ws_bytes, _stream())
    return out


class LPIPSMetric(Metric):
    """Video-level LPIPS (metrics/lpips.py:7-36).  The perceptual network (`lpips.LPIPS(net='vgg')` upstream) is third-party: it is
    passed in, exactly as the reference's own drivers pass it (`LPIPSMetric(device=..., lpips_net=net)`, replicate_scorer.py:63-74);
    the input normalisation / layout / resize in front of it runs on the device."""

    def __init__(self, device=None, lpips_net=None):
        super().__init__(name="lpips")
        self.device = device or "cuda"
        self.lpips = lpips_net

    def compute(self, *, gt, rep, **kwargs) -> float:
        if self.lpips is None:
            raise RuntimeError("LPIPSMetric: the LPIPS-VGG weights are third-party and not bundled; pass lpips_net=lpips.LPIPS(net='vgg')")
        gt_t = frames_to_pm1(gt)
        rep_t = frames_to_pm1(rep, size=tuple(gt_t.shape[-2:]))
        with torch.no_grad():
            d = self.lpips(gt_t, rep_t)
        return float(torch.as_tensor(d, dtype=torch.float32).mean().item())


class Consistency_Score(Metric):
    """MSE + ratio * LPIPS, motion score returned separately (metrics/consistency_score.py:43-72).  `lpips_net` is the caller's
    perceptual network, wrapped in LPIPSMetric like the reference does (:54-58)."""

    def __init__(self, lpips_net=None, device="cuda"):
        super().__init__("Consistency_Score")
        self.device = device
        self.mse_metric = MSEMetric()
        self.lpips_metric = LPIPSMetric(lpips_net=lpips_net, device=device)

    def compute(self, *, gt, rep, extrinsics, ratio=1, **kwargs):
        val_mse = self.mse_metric.compute(gt=gt, rep=rep)
        val_lpips = self.lpips_metric.compute(gt=gt, rep=rep) if ratio != 0 else 0.0     # ratio = 0: usable without the network
        motion = compute_motion_score_vectorized(extrinsics, device=self.device)
        return float(val_mse + ratio * val_lpips), float(motion)


def epipolar_errors(pts1_list, pts2_list, return_F=False):
    """Per pair mean sqrt(sampson^2 + 1e-8) from matched points (lists of [N_i,2]); -1 for pairs with < 8 matches."""
    P = len(pts1_list)
    sizes = [len(p) for p in pts1_list]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64, device="cuda")
    cat = lambda lst: _dev_f32(np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for p in lst], 0) if sum(sizes) else np.zeros((1, 2), np.float32))
    p1, p2 = cat(pts1_list), cat(pts2_list)
    err = torch.empty(P, dtype=torch.float32, device="cuda")
    Fm = torch.empty(P, 9, dtype=torch.float32, device="cuda") if return_F else None
    _lib.call("vgpa_epipolar_sampson", p1, p2, offs, P, err, Fm, _stream())
    return (err, Fm.view(P, 3, 3)) if return_F else err


class EpipolarMetric(Metric):
    """Video-level epipolar consistency (metrics/epipolar.py:142-232).  `matcher(frame_i, frame_j) -> (pts1, pts2)` is the
    caller's SIFT / LightGlue front end (third-party); the 8-point + Sampson geometry runs on device."""

    def __init__(self, descriptor_type: str = "sift", ratio_thresh: float = 0.75, min_matches: int = 20, device: str = None, matcher=None):
        """Reference signature (metrics/epipolar.py:146-158) + `matcher`: the reference builds a cv2-SIFT or LightGlue matcher from
        `descriptor_type`; both are third-party, so the front end is the caller's callable here."""
        super().__init__(name="Epipolar")
        if descriptor_type not in ("sift", "lightglue"):
            raise ValueError(f"Unsupported descriptor type: {descriptor_type}")
        self.descriptor_type, self.ratio_thresh, self.device = descriptor_type, ratio_thresh, device or "cuda"
        self.matcher, self.min_matches = matcher, min_matches

    def compute_from_matches(self, pts1_list, pts2_list) -> float:
        keep = [(a, b) for a, b in zip(pts1_list, pts2_list) if a is not None and b is not None and len(a) >= self.min_matches]
        if not keep:
            return -1.0
        err = epipolar_errors([a for a, _ in keep], [b for _, b in keep]).cpu().numpy()
        err = err[np.isfinite(err) & (err >= 0)]
        return float(err.mean()) if err.size else -1.0

    def compute(self, *, gt, rep=None, **kwargs) -> float:
        if self.matcher is None:
            raise RuntimeError("EpipolarMetric.compute needs a keypoint matcher (SIFT / LightGlue are third-party); "
                               "use compute_from_matches with precomputed correspondences")
        frames = gt
        m = [self.matcher(frames[i], frames[i + 1]) for i in range(len(frames) - 1)]
        return self.compute_from_matches([a for a, _ in m], [b for _, b in m])
