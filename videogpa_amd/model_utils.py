"""Host-side mirror of utils/model_utils.py (reference), on the HIP kernels.

`preprocess_images_from_numpy` (:16-85) keeps the reference's name, arguments, error messages and result ([1, T, 3, h, w] float32 in
[0, 1]); the frames go to the device as uint8 once and everything after that (PIL-exact bicubic resize, /255, crop or white pad) is
one C-ABI call, vgpa_preprocess_frames (csrc/preprocess.hip).  No CPU fallback.

`run_model_gpu` (:89-123) in the reference wraps the external VGGT network; the part of it that is arithmetic of this repository --
preprocessing in front, pose decoding behind -- is `prepare_inputs` / scorer.pose_encoding_to_extri_intri."""
import ctypes

import numpy as np
import torch

from . import _lib

_MODES = {"crop": 0, "pad": 1}


def preprocessed_size(H, W, mode="crop"):
    """(h, w) of the preprocessed frames: utils/model_utils.py:36-48,54-71"""
    h, w = ctypes.c_int32(), ctypes.c_int32()
    rc = _lib.query("vgpa_preprocess_shape", int(H), int(W), _MODES[mode], ctypes.byref(h), ctypes.byref(w))
    if rc != 0:
        raise ValueError(f"no preprocessed size for frames of {H} x {W}")
    return h.value, w.value


def preprocess_images_from_numpy(frames_np_array, mode: str = "crop", device="cuda") -> torch.Tensor:
    """utils/model_utils.py:16-85.  `frames_np_array`: [T, H, W, 3] uint8 RGB (numpy, or a uint8 tensor already on the device)."""
    if frames_np_array.ndim != 4 or frames_np_array.shape[-1] != 3:
        raise ValueError("Input frames_np_array must be [T, H, W, 3] (RGB).")
    if mode not in _MODES:
        raise ValueError("Mode must be either 'crop' or 'pad'")
    if isinstance(frames_np_array, np.ndarray):
        if frames_np_array.dtype != np.uint8:
            raise TypeError("frames must be uint8 (PIL's Image.fromarray(frame, 'RGB') reads them as bytes)")
        frames = torch.from_numpy(np.ascontiguousarray(frames_np_array)).to(device)
    else:
        if frames_np_array.dtype != torch.uint8:
            raise TypeError("frames must be uint8")
        frames = frames_np_array.to(device).contiguous()
    T, H, W, _ = frames.shape
    if T == 0:
        raise RuntimeError("stack expects a non-empty TensorList")   # torch.stack([]) at :77
    h, w = preprocessed_size(H, W, mode)
    out = torch.empty(T, 3, h, w, dtype=torch.float32, device=frames.device)
    ws_bytes = _lib.query("vgpa_preprocess_workspace_bytes", T, H, W, _MODES[mode])
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=frames.device)
    _lib.call("vgpa_preprocess_frames", frames, T, H, W, _MODES[mode], out, ws, ws_bytes, torch.cuda.current_stream(frames.device).cuda_stream)
    return out.unsqueeze(0)


def prepare_inputs(frames_np_array, device="cuda"):
    """the front of run_model_gpu (:98-101): images [1, T, 3, h, 518] on the device; raises like :100-101 on an empty clip"""
    if frames_np_array.shape[0] == 0:
        raise ValueError("No frames processed from input array.")
    return preprocess_images_from_numpy(frames_np_array, device=device)
