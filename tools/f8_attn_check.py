"""The e4m3 head_dim-128 forward (vgpa_attn128_fwd_f8) against fp64 and against the bf16 w1 forward: error table over shapes (ragged tails, token-major
views, sharp / diffuse softmax, outlier rows), then timing at the Wan2.2 self-attention shape.   python tools/f8_attn_check.py [--time]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from videogpa_amd import ops  # noqa: E402
from test_gpu_wan_kernels import _ref_e4m3  # noqa: E402


def ref(q, k, v, scale):
    s = (q.double() @ k.double().transpose(-1, -2)) * scale
    return torch.softmax(s, dim=-1) @ v.double(), torch.logsumexp(s, -1) / math.log(2.0)


def stats(got, want):
    got, want = got.double().flatten(), want.double().flatten()
    return (got - want).abs().max().item() / want.abs().max().item(), float(got @ want / (got.norm() * want.norm())), float((got - want).norm() / want.norm())


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    scale = 128 ** -0.5
    print(f"{'case':38s} {'f8 max/range':>12s} {'cos':>9s} {'rel':>8s} {'lse2 err':>9s} | {'bf16 max/range':>14s} {'rel':>8s}")
    for name, B, H, Sq, Skv, qmul, vmean in (("1x2 1024x1024", 1, 2, 1024, 1024, 1.0, 0.0), ("ragged 700x1500", 1, 3, 700, 1500, 1.0, 0.0),
                                              ("ragged 1030x1091 B2", 2, 2, 1030, 1091, 1.0, 0.0), ("sharp q x4", 1, 2, 512, 2048, 4.0, 0.0),
                                              ("diffuse q x0.1, v mean 2", 1, 2, 512, 4096, 0.1, 2.0), ("64x4096", 1, 1, 64, 4096, 1.0, 0.0),
                                              ("wan-like 2048x2048 rmsnormed", 1, 4, 2048, 2048, 1.0, 0.0)):
        q = (qmul * torch.randn(B, Sq, H, 128, device="cuda", generator=g)).bfloat16().permute(0, 2, 1, 3)
        k = torch.randn(B, Skv, H, 128, device="cuda", generator=g).bfloat16().permute(0, 2, 1, 3)
        v = (torch.randn(B, Skv, H, 128, device="cuda", generator=g) + vmean).bfloat16().permute(0, 2, 1, 3)
        o8, l8 = ops.attention128_fwd_raw(q, k, v, scale, f8=True)
        ob, lb = ops.attention128_fwd_raw(q, k, v, scale)
        ro, rl = ref(q, k, v, scale)
        a, c, r = stats(o8, ro)
        ab, _, rb = stats(ob, ro)
        r8, rl8 = _ref_e4m3(q, k, v, scale)
        a8, c8, rr8 = stats(o8, r8)
        print(f"{name:38s} {a:12.4f} {c:9.6f} {r:8.4f} {(l8.double() - rl).abs().max().item():9.4f} | {ab:14.4f} {rb:8.4f}   finite {bool(torch.isfinite(o8).all())}"
              f" | vs e4m3-operand fp64: max/range {a8:.4f} cos {c8:.6f} rel {rr8:.4f} lse2 {(l8.double() - rl8).abs().max().item():.4f}")
    # outliers: one huge query row, one huge key row (redo path), first-tile-only mass
    B, H, Sq, Skv = 1, 2, 600, 1200
    for which in ("q", "k"):
        q = torch.randn(B, H, Sq, 128, device="cuda", generator=g).bfloat16()
        k = torch.randn(B, H, Skv, 128, device="cuda", generator=g).bfloat16()
        v = torch.randn(B, H, Skv, 128, device="cuda", generator=g).bfloat16()
        if which == "q":
            q[0, 0, 300] *= 40
        else:
            k[0, 1, 17] *= 40
        o8, _ = ops.attention128_fwd_raw(q, k, v, scale, f8=True)
        ro, _ = ref(q, k, v, scale)
        a, c, r = stats(o8, ro)
        print(f"{'outlier ' + which:38s} {a:12.4f} {c:9.6f} {r:8.4f}   finite {bool(torch.isfinite(o8).all())}")
    if "--time" in sys.argv:
        B, H, S = 2, 24, 18480
        q, k, v = (torch.randn(B, S, H, 128, device="cuda", generator=g).bfloat16().permute(0, 2, 1, 3) for _ in range(3))
        for f8 in (False, True, False, True):
            for _ in range(2):
                ops.attention128_fwd_raw(q, k, v, scale, f8=f8)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                ops.attention128_fwd_raw(q, k, v, scale, f8=f8)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            print(f"Wan2.2 self-attention forward B={B} H={H} S={S}: {'e4m3' if f8 else 'bf16'} {ms:.3f} ms = {4.0 * B * H * S * S * 128 / ms / 1e9:.0f} TFLOP/s (incl. prep passes)")
        o8, l8 = ops.attention128_fwd_raw(q, k, v, scale, f8=True)
        ob, lb = ops.attention128_fwd_raw(q, k, v, scale)
        a, c, r = stats(o8, ob)
        print(f"full size: e4m3 vs bf16 output: max/range {a:.4f} cos {c:.6f} rel {r:.4f}; lse2 max diff {(l8 - lb).abs().max().item():.4f}")


if __name__ == "__main__":
    main()
