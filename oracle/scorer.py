"""Geometry-consistency scorer math, CPU oracle (numpy / torch fp32).

PINNED (tests/golden/scorer.pt, generated from the importable reference modules):
  project_points   utils/projection_utils.py:12-51   (z-descending sort + scatter == nearest-wins z-buffer)
  batch_reproject  utils/projection_utils.py:57-101  (loop restated; the reference hard-codes 'cuda' :68-81)
  motion_score     metrics/consistency_score.py:8-40
  mse / psnr       metrics/mse.py:14-54,56-80
  pointcloud filter utils/pointcloud_utils.py:10-80
PARITY UNPINNED (kornia>=0.7.3, requirements.txt:33, not vendored / not installed):
  find_fundamental (8-point) and sampson_epipolar_distance, call sites metrics/epipolar.py:201,210-213;
  restated from kornia.geometry.epipolar.{fundamental,_metrics}.

fp32 evaluation order of the projection is fixed (no FMA contraction):
  cam_j  = ((x*R[j,0] + y*R[j,1]) + z*R[j,2]) + t[j]
  proj_i = ((cam_0*K[i,0] + cam_1*K[i,1]) + cam_2*K[i,2])
so the HIP kernel can reproduce it bit-for-bit.
"""
import numpy as np
import torch

f32 = np.float32


def _project_uvz(pc, K, E):
    pc = np.asarray(pc, f32)
    K = np.asarray(K, f32)
    E = np.asarray(E, f32)
    R, t = E[:3, :3], E[:3, 3]
    x, y, z = pc[:, 0], pc[:, 1], pc[:, 2]
    cam = [((x * R[j, 0] + y * R[j, 1]) + z * R[j, 2]) + t[j] for j in range(3)]
    pr = [((cam[0] * K[i, 0] + cam[1] * K[i, 1]) + cam[2] * K[i, 2]) for i in range(3)]
    zz = pr[2]
    with np.errstate(all="ignore"):
        den = zz + f32(1e-8)
        u = np.rint(pr[0] / den)
        v = np.rint(pr[1] / den)
    return u, v, zz


def project_points(pc, colors, K, E, H, W, bg=(0, 0, 0)):
    """-> uint8 [H,W,3].  Nearest z wins a pixel; exact z ties go to the LOWEST point index
    (the reference's tie order is unspecified: unstable argsort + duplicate-index scatter)."""
    u, v, z = _project_uvz(pc, K, E)
    colors = np.asarray(colors, f32)
    with np.errstate(all="ignore"):
        valid = (u >= 0) & (u < W) & (v >= 0) & (v < H) & (z > 0) & np.isfinite(u) & np.isfinite(v)
    canvas = np.empty((H, W, 3), np.uint8)
    canvas[:] = np.asarray(bg, np.uint8)
    idx = np.nonzero(valid)[0]
    if idx.size == 0:
        return canvas
    c = colors[idx]
    if c.max() <= 1.0:
        c = np.clip(c * f32(255), 0, 255).astype(np.uint8)
    else:
        c = np.clip(c, 0, 255).astype(np.uint8)
    pix = v[idx].astype(np.int64) * W + u[idx].astype(np.int64)
    zi = z[idx]
    # sort by (pixel, z asc, index asc); first of each pixel run wins
    order = np.lexsort((idx, zi, pix))
    pix_s = pix[order]
    first = np.ones(pix_s.shape, bool)
    first[1:] = pix_s[1:] != pix_s[:-1]
    win = order[first]
    canvas.reshape(-1, 3)[pix[win]] = c[win]
    return canvas


def batch_reproject(pc, colors, intrinsics, extrinsics, H, W):
    """-> float32 [T,3,H,W] in [-1,1] (utils/projection_utils.py:100-101)."""
    T = len(extrinsics)
    if T == 0:
        return np.zeros((0, 3, H, W), f32)
    imgs = [project_points(pc, colors, intrinsics[i], extrinsics[i], H, W) for i in range(T)]
    stack = np.stack(imgs).transpose(0, 3, 1, 2).astype(f32)
    return (stack / f32(255.0)) * f32(2.0) - f32(1.0)


def motion_score(extrinsics):
    """mean||dt|| + 0.1 * mean acos(clamp((tr(R_{i+1} R_i^T) - 1)/2)); NaN -> 0."""
    E = torch.as_tensor(np.asarray(extrinsics), dtype=torch.float32)
    Rs, ts = E[:, :3, :3], E[:, :3, 3]
    mean_trans = torch.norm(ts[1:] - ts[:-1], dim=1).mean()
    dR = torch.matmul(Rs[1:], Rs[:-1].transpose(-1, -2))
    tr = dR.diagonal(dim1=-2, dim2=-1).sum(-1)
    ang = torch.acos(torch.clamp((tr - 1) / 2, -1.0, 1.0))
    s = mean_trans + 0.1 * ang.mean()
    return 0.0 if torch.isnan(s) else float(s)


def to_01(x):
    """metrics/mse.py:31-54 range heuristics; -> float32 [B,3,H,W] in [0,1]."""
    t = torch.as_tensor(x).float()
    if t.ndim == 3:
        t = t.unsqueeze(0)
    if t.shape[-1] == 3:
        t = t.permute(0, 3, 1, 2)
    if isinstance(x, torch.Tensor):
        if t.min() < 0:
            t = (t + 1.0) / 2.0
        elif t.max() > 1.0:
            t = t / 255.0
    else:
        if t.max() > 1.0:
            t = t / 255.0
    return t


def mse(gt, rep):
    return float((to_01(gt) - to_01(rep)).pow(2).mean().item())


def pointcloud_filter(points, conf, images, conf_thres=0.0):
    """utils/pointcloud_utils.py:10-80 -> (vertices [N,3], colors [N,3] in 0..255 float)."""
    points = torch.as_tensor(points).float()
    conf = torch.as_tensor(conf).float()
    images = torch.as_tensor(images).float()
    vertices = points.reshape(-1, 3)
    colors = images.permute(0, 2, 3, 1) if (images.ndim == 4 and images.shape[1] == 3) else images
    colors = colors.reshape(-1, 3) * 255
    vals = conf.reshape(-1)
    valid = torch.isfinite(vals) & (vals > 1e-5)
    if conf_thres <= 0:
        mask = valid
    else:
        n = int(valid.sum())
        if n == 0:
            mask = valid
        else:
            keep = max(0.0, min(1.0, 1.0 - conf_thres / 100.0))
            k = max(1, int(np.ceil(n * keep)))
            thr = torch.topk(vals[valid], k)[0][-1]
            mask = valid & (vals >= thr)
    return vertices[mask], colors[mask]


# ---------------------------------------------------------------- epipolar (kornia restatement, fp64 math)
def _normalize_points(p, eps=1e-8):
    mean = p.mean(axis=0)
    scale = np.sqrt(2.0) / (np.linalg.norm(p - mean, axis=1).mean() + eps)
    T = np.array([[scale, 0, -scale * mean[0]], [0, scale, -scale * mean[1]], [0, 0, 1.0]])
    return (p - mean) * scale, T


def find_fundamental(p1, p2):
    """Normalised 8-point, unit weights; p1,p2 [N,2] -> F [3,3] scaled so F[2,2]=1 (when |F22|>1e-8)."""
    p1 = np.asarray(p1, np.float64)
    p2 = np.asarray(p2, np.float64)
    n1, T1 = _normalize_points(p1)
    n2, T2 = _normalize_points(p2)
    x1, y1, x2, y2 = n1[:, 0], n1[:, 1], n2[:, 0], n2[:, 1]
    X = np.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, np.ones_like(x1)], axis=1)
    _, _, Vt = np.linalg.svd(X.T @ X)
    Fm = Vt[-1].reshape(3, 3)
    U, S, Vt2 = np.linalg.svd(Fm)
    S[2] = 0.0
    Fm = U @ np.diag(S) @ Vt2
    Fm = T2.T @ Fm @ T1
    if abs(Fm[2, 2]) > 1e-8:
        Fm = Fm / (Fm[2, 2] + 1e-8)
    return Fm


def sampson_distance_sq(p1, p2, Fm):
    p1 = np.asarray(p1, np.float64)
    p2 = np.asarray(p2, np.float64)
    h1 = np.concatenate([p1, np.ones((len(p1), 1))], axis=1)
    h2 = np.concatenate([p2, np.ones((len(p2), 1))], axis=1)
    l1 = h1 @ Fm.T  # F x1
    l2 = h2 @ Fm    # F^T x2
    num = (h2 * l1).sum(axis=1) ** 2
    den = l1[:, 0] ** 2 + l1[:, 1] ** 2 + l2[:, 0] ** 2 + l2[:, 1] ** 2
    return num / den


def epipolar_pair_error(p1, p2):
    """metrics/epipolar.py:177-213: mean sqrt(sampson^2 + 1e-8) for one frame pair."""
    Fm = find_fundamental(p1, p2)
    return float(np.sqrt(sampson_distance_sq(p1, p2, Fm) + 1e-8).mean())
