"""ctypes binding of the C-ABI HIP library (include/videogpa_hip.h).  No fallback: if the library is missing
or a call fails, this raises -- the product path never silently runs on something else."""
import ctypes
import os

# torch first: its wheel bundles the HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  Loading our
# library before torch would pull /opt/rocm's copy in as a SECOND runtime instance and launches on torch's streams
# would fail; after torch, our NEEDED libamdhip64.so.7 resolves to the instance already loaded.
import torch  # noqa: F401

from ctypes import c_float, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# VGPA_LIB=<path> selects another build of the SAME C-ABI (tools/build_variant.sh writes var/lib_NAME.so) for in-session A/B
# measurements, so nothing ever has to be copied over the product library.
LIB_PATH = os.environ.get("VGPA_LIB") or os.path.join(_HERE, "csrc", "libvgpa_hip.so")

P, I64, I32, F32, SZ = c_void_p, c_int64, c_int32, c_float, c_size_t

# name -> (restype, [argtypes]) ; must mirror include/videogpa_hip.h (tests/test_cabi.py checks the symbol set)
SIGNATURES = {
    "vgpa_dpo_loss_workspace_bytes": (SZ, [I64]),
    "vgpa_dpo_loss_fwd": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I32, F32, F32, I32, I32, P, P, P, P, SZ, P]),
    "vgpa_dpo_loss_bwd": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I32, F32, I32, P, P, P, P, P]),
    "vgpa_noise_velocity_paired": (I32, [P, P, P, P, P, I64, I64, I32, I32, P, P, P]),
    "vgpa_flow_noise_velocity_paired": (I32, [P, P, P, I64, I64, I32, I32, P, P, P]),
    "vgpa_ln_modulate_fwd": (I32, [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, F32, P, P, P, P]),
    "vgpa_ln_modulate_bwd": (I32, [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, P, P, P]),
    "vgpa_residual_ln_fwd": (I32, [P, P, P, P, I64, P, P, P, P, P, P, I64, I64, I64, I64, I64, F32, P, P, I64, P, P, P]),
    "vgpa_residual_ln_bwd": (I32, [P, P, P, P, P, P, P, I64, P, P, I64, P, I64, I64, I64, I64, P, P, I64, P]),
    "vgpa_gate_residual": (I32, [P, P, P, P, I64, I64, I64, I64, I64, P, P]),
    "vgpa_gelu_tanh_fwd": (I32, [P, I64, P, P]),
    "vgpa_gelu_tanh_bwd": (I32, [P, P, I64, P, P]),
    "vgpa_qknorm_rope_fwd": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, F32, F32, I32, P]),
    "vgpa_qknorm_rope_bwd": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, F32, I32, P]),
    "vgpa_attn_fwd": (I32, [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, P]),
    "vgpa_attn_fwd_workspace_bytes": (SZ, [I64, I64, I64]),
    "vgpa_attn_fwd_ws": (I32, [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, I32, P, SZ, P]),
    "vgpa_attn_bwd_workspace_bytes": (SZ, [I64, I64, I64]),
    "vgpa_grad_norm_workspace_bytes": (SZ, []),
    "vgpa_grad_norm": (I32, [P, I64, F32, P, P, SZ, P]),
    "vgpa_adamw_step": (I32, [P, P, P, P, I64, F32, F32, F32, F32, F32, I64, F32, F32, P, P]),
    "vgpa_attn_bwd_delta": (I32, [P, P, P, P, P, I64, I64, I64, I64, P]),
    "vgpa_attn_bwd_delta_res": (I32, [P, P, I32, P, P, P, P, P, I64, I64, I64, I64, P]),
    "vgpa_attn_bwd_dkv": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, P]),
    "vgpa_attn_bwd_dq": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, P]),
    "vgpa_attn_bwd_dq_w1": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, I32, P, SZ, P]),
    "vgpa_attn_fwd_w1_workspace_bytes": (SZ, [I64, I64, I64]),
    "vgpa_attn_fwd_w1": (I32, [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, I32, P, SZ, P]),
    "vgpa_attn_bwd_prep_w1": (I32, [P, P, P, P, P, P, P, I64, I64, I64, I64, P]),
    "vgpa_attn_bwd_prep_w1_res": (I32, [P, P, I32, P, P, P, P, P, P, P, I64, I64, I64, I64, P]),
    "vgpa_attn_fwd_w1_res": (I32, [P, P, P, P, P, I32, P, P, P, P, P, P, I64, I64, I64, I64, F32, I32, P, SZ, P]),
    "vgpa_attn_fwd_online_res": (I32, [P, P, P, P, P, I32, P, P, P, P, P, P, I64, I64, I64, I64, F32, I32, P, SZ, P]),
    "vgpa_attn_bwd_dkv_w1": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, I32, P, SZ, P]),
    "vgpa_attn_bwd_split_workspace_bytes": (SZ, [I64, I64, I64]),
    "vgpa_attn_bwd_dkv_ws": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, I32, P, SZ, P]),
    "vgpa_attn_bwd_dq_ws": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, I32, P, SZ, P]),
    "vgpa_attn_bwd": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, P, SZ, P]),
    "vgpa_lora_down": (I32, [P, I64, P, P, I64, I64, I64, I64, P]),
    "vgpa_lora_up_add": (I32, [P, I64, P, I64, P, I64, F32, I64, I64, I64, I32, P]),
    "vgpa_lora_grad": (I32, [P, I64, P, I64, P, I64, F32, I64, I64, I64, P]),
    "vgpa_lora_grad_workspace_bytes": (SZ, [I64, I64, I64]),
    "vgpa_lora_grad_ws": (I32, [P, I64, P, I64, P, I64, F32, I64, I64, I64, P, SZ, P]),
    "vgpa_lora_ext_refresh": (I32, [P, P, F32, I64, I64, I64, P, P, I64, P, I64, P, P]),
    "vgpa_attn128_fwd_workspace_bytes": (SZ, [I64, I64, I64]),
    "vgpa_attn128_fwd": (I32, [P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, P, SZ, P]),
    "vgpa_attn128_fwd_f8_workspace_bytes": (SZ, [I64, I64, I64, I64]),
    "vgpa_attn128_fwd_f8": (I32, [P] * 17 + [I64, I64, I64, I64, F32, P, SZ, P]),
    "vgpa_attn128_bwd_workspace_bytes": (SZ, [I64, I64, I64]),
    "vgpa_attn128_bwd": (I32, [P] * 19 + [I64, I64, I64, I64, F32, I32, P, SZ, P]),
    "vgpa_attn128_bwd_prescaled": (I32, [P] * 19 + [I64, I64, I64, I64, F32, I32, P, SZ, P]),
    "vgpa_wan_ln_mod_fwd": (I32, [P, I32, P, P, P, P, P, I64, I64, I64, F32, I32, P, I64, P, P, P, P, P]),
    "vgpa_wan_ln_mod_bwd": (I32, [P, P, I32, P, P, P, P, P, I64, I64, I64, P, P, P]),
    "vgpa_wan_ln_mod_fwd_f32": (I32, [P, P, P, P, I64, I64, I64, F32, P, P, P, P]),
    "vgpa_wan_ln_mod_bwd_f32": (I32, [P, P, P, P, P, P, I64, I64, I64, P, P]),
    "vgpa_wan_gate_ln_mod_fwd": (I32, [P, P, P, P, P, P, P, P, I64, I64, I64, F32, P, P, I64, P, P, P, P, P]),
    "vgpa_wan_ln_mod_bwd_gate": (I32, [P, P, P, P, P, P, P, I64, I64, I64, P, P, P, P, I64, P]),
    "vgpa_wan_gate_residual": (I32, [P, P, P, P, I64, I64, I64, P, P]),
    "vgpa_wan_gate_bwd": (I32, [P, P, P, I64, I64, I64, P, I64, P]),
    "vgpa_wan_gate_bwd_q8": (I32, [P, P, P, I64, I64, I64, P, P, P]),
    "vgpa_wan_rms_rope_fwd": (I32, [P, I64, P, P, P, I64, I64, I64, I64, F32, P, I64, P, P]),
    "vgpa_wan_rms_rope_bwd": (I32, [P, I64, P, I64, P, P, P, P, I64, I64, I64, I64, P, I64, P]),
    "vgpa_quant_fp8_rows": (I32, [P, I64, P, P, I64, I64, P]),
    "vgpa_gelu_tanh_fwd_q8": (I32, [P, I64, I64, P, P, P]),
    "vgpa_gelu_tanh_bwd_q8": (I32, [P, P, I64, I64, P, P, P]),
    "vgpa_preprocess_shape": (I32, [I32, I32, I32, P, P]),
    "vgpa_preprocess_workspace_bytes": (SZ, [I32, I32, I32, I32]),
    "vgpa_preprocess_frames": (I32, [P, I32, I32, I32, I32, P, P, SZ, P]),
    "vgpa_project_points_workspace_bytes": (SZ, [I64, I64, I64]),
    "vgpa_project_points": (I32, [P, P, P, F32, P, P, P, I32, I64, I64, I64, I64, P, P, P, SZ, P]),
    "vgpa_conf_threshold_workspace_bytes": (SZ, []),
    "vgpa_conf_threshold": (I32, [P, I64, F32, P, P, SZ, P]),
    "vgpa_frame_metric_workspace_bytes": (SZ, []),
    "vgpa_frame_metric": (I32, [P, I32, I32, I32, P, I32, I32, I32, I64, I64, I64, I64, I64, I64, I32, P, P, SZ, P]),
    "vgpa_frames_to_pm1": (I32, [P, I32, I32, I32, I64, I64, I64, I64, I64, I64, P, P, SZ, P]),
    "vgpa_mvcs_workspace_bytes": (SZ, [I64]),
    "vgpa_mvcs": (I32, [P, P, I32, P, I32, I64, I64, I64, P, P, SZ, P]),
    "vgpa_unproject_depth": (I32, [P, P, P, I32, I64, I64, I64, P, P]),
    "vgpa_pose_decode": (I32, [P, I64, F32, F32, P, P, P]),
    "vgpa_frame_mse_workspace_bytes": (SZ, []),
    "vgpa_frame_mse": (I32, [P, I32, I32, I32, P, I32, I32, I32, I64, I64, I64, I64, P, P, SZ, P]),
    "vgpa_motion_score": (I32, [P, I32, I64, P, P]),
    "vgpa_epipolar_sampson": (I32, [P, P, P, I64, P, P, P]),
}

# exported only by variant builds (tools/build_variant.sh -> VGPA_LIB=...): measured-slower experiments kept out of the product library
OPTIONAL_SIGNATURES = {
    "vgpa_attn_bwd_fused": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, P]),
    "vgpa_gemm_bf16": (I32, [P, I64, P, I64, P, P, I64, P, I64, I32, I32, I32, I32, P]),
}

_ERR = {-1: "invalid argument", -2: "kernel launch failed", -3: "workspace too small"}
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"videogpa_amd: HIP library not built ({LIB_PATH} missing). Run `python -m videogpa_amd.build` "
                "(or __graft_entry__.build()). There is no CPU / PyTorch fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so is stale: fail loudly
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in OPTIONAL_SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        _lib = lib
    return _lib


def _conv(a):
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a


def call(name, *args):
    """Call a status-returning entry point; tensors are passed as raw device pointers."""
    fn = getattr(load(), name)
    rc = fn(*[_conv(a) for a in args])
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")


def query(name, *args):
    return getattr(load(), name)(*args)


def has(name):
    """Is an optional (variant-build) entry point present in the loaded library?"""
    return hasattr(load(), name)
