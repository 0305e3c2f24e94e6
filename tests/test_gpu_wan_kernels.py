"""Kernels of the Wan2.2-TI2V-5B attention shapes (BASELINE.json configs[5]): head_dim-128 attention with separate query and key
lengths (self-attention, cross-attention over the 512 text tokens), against an fp64 torch reference on the same bf16 inputs.
bf16 outputs: within 2 % of the tensor's range and cosine >= 0.9995 for the output and every gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, do, scale):
    q, k, v = (t.double().requires_grad_(True) for t in (q, k, v))
    p = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
    o = p @ v
    o.backward(do.double())
    return o, q.grad, k.grad, v.grad


def _close(got, ref, what, tol=0.02):
    got, ref = got.detach().double().cpu(), ref.detach().cpu()
    err = (got - ref).abs().max().item()
    if ref.abs().max().item() == 0.0:        # one key: P = 1, so dq and dk vanish identically
        assert err <= 1e-4, (what, err)
        return
    cos = float((got.flatten() @ ref.flatten()) / (got.norm() * ref.norm()).clamp_min(1e-300))
    assert err <= tol * ref.abs().max().item() and cos >= 0.9995, (what, err, ref.abs().max().item(), cos)


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 128, 128), (2, 3, 300, 300), (1, 2, 257, 64), (1, 4, 200, 512), (1, 1, 64, 77), (1, 2, 1, 130), (1, 1, 130, 1)])
def test_attention128_forward_backward_vs_fp64(B, H, Sq, Skv):
    from videogpa_amd import ops
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    g = torch.Generator(device="cuda").manual_seed(Sq * 1000 + Skv)
    q = torch.randn(B, H, Sq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(B, H, Skv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(B, H, Skv, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(B, H, Sq, 128, device="cuda", generator=g).bfloat16()
    scale = 128 ** -0.5
    qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
    o = ops.attention128(qg, kg, vg, scale)
    o.backward(do)
    ro, rq, rk, rv = _ref(q, k, v, do, scale)
    _close(o, ro, "o")
    _close(qg.grad, rq, "dq")
    _close(kg.grad, rk, "dk")
    _close(vg.grad, rv, "dv")


def test_attention128_strided_views_and_sharp_softmax():
    """[B, S, H, 128] storage viewed as [B, H, S, 128] (how the projection leaves q/k/v), and scores large enough that the running max
    moves tile to tile (scale 1.0 on unit-variance rows of 128: |s| up to ~40)."""
    from videogpa_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    B, H, Sq, Skv = 2, 3, 190, 333
    qs = torch.randn(B, Sq, H, 128, device="cuda", generator=g).bfloat16()
    kvs = torch.randn(B, Skv, 2, H, 128, device="cuda", generator=g).bfloat16()
    q, k, v = qs.permute(0, 2, 1, 3), kvs[:, :, 0].permute(0, 2, 1, 3), kvs[:, :, 1].permute(0, 2, 1, 3)
    do = torch.randn(B, H, Sq, 128, device="cuda", generator=g).bfloat16()
    qg, kg, vg = (t.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for t in (q, k, v))
    assert not qg.is_contiguous()
    o = ops.attention128(qg, kg, vg, 1.0)
    o.backward(do)
    ro, rq, rk, rv = _ref(q, k, v, do, 1.0)
    _close(o, ro, "o")
    _close(qg.grad, rq, "dq", 0.03)
    _close(kg.grad, rk, "dk", 0.03)
    _close(vg.grad, rv, "dv", 0.03)
