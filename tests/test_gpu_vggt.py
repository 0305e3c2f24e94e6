"""The HIP attention kernels behind the reference-HELD attention (vggt/layers/attention.py, rope.py, block.py; aggregator's frame /
global alternation) against outputs of those modules themselves (tests/golden/vggt_attention.pt, made by importing the reference):
the first model-kernel parity that does not rest on a restated third-party library.  Inputs and weights of the fixture are
bf16-representable; the HIP path computes in bf16 with fp32 accumulation: outputs within 2 % of range (bf16 activations through
qkv GEMM -> QK-norm -> RoPE -> attention -> proj), gradients within 3 % of range and cosine >= 0.999."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.load(os.path.join(HERE, "golden", "vggt_attention.pt"))


def _close(got, ref, tol, what):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs().max().item()
    cos = float((got.double().flatten() @ ref.double().flatten()) / (got.double().norm() * ref.double().norm()).clamp_min(1e-300))
    assert err <= tol * ref.abs().max().item() and cos >= 0.999, (what, err, ref.abs().max().item(), cos)


def _load(module, params):
    missing, unexpected = module.load_state_dict({k: v.float() for k, v in params.items()}, strict=True)
    return module.to(device="cuda", dtype=torch.bfloat16)


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_attention_matches_reference_module_outputs(gold, case):
    from videogpa_amd.vggt import Attention, RotaryPositionEmbedding2D
    c = gold["attention"][case]
    att = _load(Attention(c["dim"], num_heads=c["heads"], qk_norm=True, rope=RotaryPositionEmbedding2D(100.0) if c["rope"] else None), c["params"])
    x = c["x"].cuda().requires_grad_(True)
    pos = None if c["pos"] is None else c["pos"].long().cuda()[None].expand(x.shape[0], -1, -1)
    y = att(x, pos=pos)
    y.backward(c["grad_out"].cuda())
    _close(y, c["y"], 0.02, "y")
    _close(x.grad, c["grad_x"], 0.03, "grad_x")
    for k in ("qkv.weight", "qkv.bias", "proj.weight", "proj.bias"):
        _close(dict(att.named_parameters())[k].grad, c["grad_params"][k], 0.03, k)


def test_frame_and_global_blocks_match_reference_module_outputs(gold):
    from videogpa_amd.vggt import Block, RotaryPositionEmbedding2D, alternating_attention
    c = gold["block"]
    rope = RotaryPositionEmbedding2D(100.0)
    blocks = [_load(Block(c["dim"], c["heads"], mlp_ratio=2.0, init_values=0.01, qk_norm=True, rope=rope), p) for p in c["params"]]
    tok = c["tokens"].cuda().requires_grad_(True)
    pos = c["pos"].long().cuda()[None].expand(c["B"] * c["S"], -1, -1).contiguous()
    outs, last = alternating_attention(tok, blocks[:1], blocks[1:], c["B"], c["S"], pos)
    last.backward(c["grad_out"].cuda().reshape(last.shape))
    C = c["dim"]
    _close(outs[0][..., :C].reshape(c["frame_out"].shape), c["frame_out"], 0.02, "frame block")
    _close(outs[0][..., C:].reshape(c["global_out"].shape), c["global_out"], 0.02, "global block")
    _close(tok.grad, c["grad_tokens"], 0.03, "grad_tokens")
    for b, gp in zip(blocks, c["grad_params"]):
        named = dict(b.named_parameters())
        for k in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.bias", "mlp.fc2.bias"):
            _close(named[k].grad, gp[k], 0.03, k)


@pytest.mark.parametrize("case", [0, 1])
def test_aggregator_matches_reference_module_outputs(case):
    """videogpa_amd.vggt.Aggregator (special tokens, positions, aa_block_num x aa_order x aa_block_size loop on the HIP blocks) loaded from the
    REFERENCE aggregator's own state dict, against that module's outputs (tests/golden/vggt_aggregator.pt: vggt/models/aggregator.py imported,
    patch_embed="conv"; aa_block_size 1 and 2): every per-depth [B, S, P, 2C] intermediate within 2 % of its range and cosine >= 0.999."""
    from videogpa_amd.vggt import Aggregator
    c = torch.load(os.path.join(HERE, "golden", "vggt_aggregator.pt"))[case]
    agg = Aggregator(img_size=c["H"], patch_size=14, embed_dim=c["embed_dim"], depth=c["depth"], num_heads=c["num_heads"], mlp_ratio=c["mlp_ratio"],
                     num_register_tokens=4, patch_embed="conv", aa_block_size=c["aa_block_size"], qk_norm=True, rope_freq=100, init_values=0.01)
    agg.load_state_dict({k: v.float() for k, v in c["params"].items()}, strict=True)       # the reference's names, strictly
    agg = agg.to(device="cuda", dtype=torch.bfloat16).eval()
    with torch.no_grad():
        outs, start = agg(c["images"].cuda().float())
    assert start == c["patch_start_idx"] and len(outs) == len(c["outputs"])
    for i, (o, r) in enumerate(zip(outs, c["outputs"])):
        assert o.shape == r.shape
        _close(o, r, 0.02, f"depth {i}")
    agg.train()                                   # training mode checkpoints every block, like the reference: same values
    x = c["images"].cuda().float().requires_grad_(True)
    outs_t, _ = agg(x)
    outs_t[-1].float().sum().backward()
    assert torch.isfinite(x.grad).all() and x.grad.abs().max() > 0
    _close(outs_t[-1], c["outputs"][-1], 0.02, "training-mode last depth")


# ---------------------------------------------------------------------------------------- second witness: Depth Anything 3's DINOv2 layers
@pytest.fixture(scope="module")
def gold_da3():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.load(os.path.join(HERE, "golden", "da3_attention.pt"))


@pytest.mark.parametrize("case", [0, 1, 2])
def test_da3_attention_matches_reference_module_outputs(gold_da3, case):
    """The same HIP attention path (fused qkv GEMM -> QK-norm + half-split 2-D RoPE kernel -> attention -> proj) against outputs of
    depth_anything_3/model/dinov2/layers/attention.py run by make_golden.py::golden_da3_attention: grid positions (cases 0, 2) and the all-zero
    positions of DA3's global pass (case 1).  Outputs within 2 % of range, gradients within 3 % of range, cosine >= 0.999 (bf16 arithmetic)."""
    from videogpa_amd.vggt import Attention, RotaryPositionEmbedding2D
    c = gold_da3["attention"][case]
    att = _load(Attention(c["dim"], num_heads=c["heads"], qk_norm=True, rope=RotaryPositionEmbedding2D(100.0)), c["params"])
    x = c["x"].cuda().requires_grad_(True)
    pos = c["pos"].long().cuda()[None].expand(x.shape[0], -1, -1)
    y = att(x, pos=pos)
    y.backward(c["grad_out"].cuda())
    _close(y, c["y"], 0.02, "y")
    _close(x.grad, c["grad_x"], 0.03, "grad_x")
    for k in ("qkv.weight", "qkv.bias", "proj.weight", "proj.bias"):
        _close(dict(att.named_parameters())[k].grad, c["grad_params"][k], 0.03, k)


def test_da3_local_and_global_blocks_match_reference_module_outputs(gold_da3):
    """DA3's local / global alternation (vision_transformer.py:351-364) on the HIP blocks with the block LayerNorm eps of 1e-6: local attention per
    view with grid positions, global attention across views with zero positions."""
    from videogpa_amd.vggt import Block, RotaryPositionEmbedding2D
    c = gold_da3["block"]
    rope = RotaryPositionEmbedding2D(100.0)
    blocks = [_load(Block(c["dim"], c["heads"], mlp_ratio=2.0, init_values=0.01, qk_norm=True, rope=rope, ln_eps=c["ln_eps"]), p) for p in c["params"]]
    assert blocks[0].norm1.eps == 1e-6 and blocks[0].attn.q_norm.eps == 1e-5
    tok = c["tokens"].cuda().requires_grad_(True)
    pos = c["pos"].long().cuda()[None].expand(c["B"] * c["S"], -1, -1).contiguous()
    t1 = blocks[0](tok, pos=pos)
    t2 = blocks[1](t1.view(c["B"], c["S"] * c["N"], c["dim"]), pos=torch.zeros(c["B"], c["S"] * c["N"], 2, dtype=torch.long, device="cuda"))
    t2.backward(c["grad_out"].cuda().reshape(t2.shape))
    _close(t1, c["local_out"], 0.02, "local block")
    _close(t2, c["global_out"], 0.02, "global block")
    _close(tok.grad, c["grad_tokens"], 0.03, "grad_tokens")
    for b, gp in zip(blocks, c["grad_params"]):
        named = dict(b.named_parameters())
        for k in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.bias", "mlp.fc2.bias"):
            _close(named[k].grad, gp[k], 0.03, k)
