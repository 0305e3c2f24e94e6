"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of the VGGT input preprocessing, only imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline.

Path: utils/model_utils.py:16-85 `preprocess_images_from_numpy` (frames [T,H,W,3] uint8 -> [1,T,3,h,518] float in [0,1]):
  sizing           :36-48   crop: width 518, height round(H * (518 / W) / 14) * 14;   pad: longer side 518, the other rounded to 14
  resize           :51      PIL `img.resize((w, h), Image.Resampling.BICUBIC)`
  ToTensor         :52      torchvision ToTensor of a uint8 RGB image = CHW float32 / 255
  centre crop      :54-56   crop mode, height > 518
  white pad        :58-71   pad mode, to 518 x 518, value 1.0

The resize is a third-party algorithm that is NOT in /root/reference: Pillow (installed here: 12.2.0), src/libImaging/Resample.c
`ImagingResample` for 8-bit images, restated below from its published algorithm: separable two-pass (horizontal, then vertical)
convolution, bicubic kernel a = -0.5, support 2 * max(scale, 1), coefficients normalised in double then fixed to 22 fractional bits
(round half away from zero), each pass accumulating in int32 from 1 << 21 and clipping (acc >> 22) to [0, 255], i.e. the intermediate
image between the passes is uint8.  Pinned against Pillow itself: tests/golden/preprocess.npz is made by calling PIL (make_golden.py
golden_preprocess) and tests/test_oracle_golden.py compares bit for bit."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
TARGET = 518


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size): bounds [out, 2] and int32 kk [out, ksize]."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """one pass along `axis` (0 = rows / vertical, 1 = columns / horizontal) of a uint8 [H, W, C] image"""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for xx in range(bounds.shape[0]):
        xmin, xmax = bounds[xx]
        acc = np.tensordot(kk[xx, :xmax].astype(np.int64), src[xmin:xmin + xmax], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def pil_bicubic_resize(img, new_w, new_h):
    """uint8 [H, W, 3] -> uint8 [new_h, new_w, 3], as PIL Image.resize((new_w, new_h), BICUBIC)"""
    H, W, _ = img.shape
    if W != new_w:
        img = _pass(img, *precompute_coeffs(W, new_w), axis=1)
    if H != new_h:
        img = _pass(img, *precompute_coeffs(H, new_h), axis=0)
    return img


def output_size(H, W, mode):
    """utils/model_utils.py:36-48 (Python round = half to even)"""
    if mode == "pad":
        if W >= H:
            new_w = TARGET
            new_h = round(H * (new_w / W) / 14) * 14
        else:
            new_h = TARGET
            new_w = round(W * (new_h / H) / 14) * 14
    else:
        new_w = TARGET
        new_h = round(H * (new_w / W) / 14) * 14
    return new_w, new_h


def preprocess_u8(frames, mode="crop"):
    """the function's result BEFORE the division by 255: uint8 [T, 3, h, w] (white pad = 255)"""
    if frames.ndim != 4 or frames.shape[-1] != 3:
        raise ValueError("Input frames_np_array must be [T, H, W, 3] (RGB).")
    if mode not in ("crop", "pad"):
        raise ValueError("Mode must be either 'crop' or 'pad'")
    out = []
    for f in frames:
        H, W, _ = f.shape
        new_w, new_h = output_size(H, W, mode)
        r = pil_bicubic_resize(f, new_w, new_h)
        if mode == "crop" and new_h > TARGET:
            s = (new_h - TARGET) // 2
            r = r[s:s + TARGET]
        if mode == "pad":
            hp, wp = TARGET - r.shape[0], TARGET - r.shape[1]
            if hp > 0 or wp > 0:
                r = np.pad(r, ((hp // 2, hp - hp // 2), (wp // 2, wp - wp // 2), (0, 0)), constant_values=255)
        out.append(r.transpose(2, 0, 1))
    return np.stack(out)


def preprocess_images_from_numpy(frames, mode="crop"):
    """[1, T, 3, h, w] float32 in [0, 1]"""
    return (preprocess_u8(frames, mode).astype(np.float32) / np.float32(255.0))[None]
