"""CPU restatement of the reference-held VGGT attention path (test infrastructure only -- see oracle/__init__.py):

  rope2d            vggt/layers/rope.py:60-188   RotaryPositionEmbedding2D (frequency 100): features [0, d/2) rotate with y, [d/2, d) with x,
                                                 rotate-half pairing inside each half
  attention         vggt/layers/attention.py:20-72   fused qkv Linear, LayerNorm(head_dim) on q and k (nn.LayerNorm: eps 1e-5), RoPE,
                                                 softmax(q k^T / sqrt(d)) v, proj
  block             vggt/layers/block.py:30-108  x + ls1(attn(norm1(x))) ; x + ls2(mlp(norm2(x))), Mlp = fc1 -> GELU (erf) -> fc2
  frame_global_pair vggt/models/aggregator.py:260-306  frame attention on (B*S, P, C), global attention on (B, S*P, C)
  aggregator        vggt/models/aggregator.py:184-258  ImageNet normalisation, patch embedding (the "conv" form), camera / register tokens (entry 0 for
                                                 a sequence's first frame, entry 1 for the others), positions (0 for special tokens, grid + 1 for
                                                 patches), aa_block_num x aa_order x aa_block_size blocks, per-depth [frame | global] intermediates

  da3_local_global_pair  depth_anything_3/model/dinov2/layers/{attention,block,rope}.py + vision_transformer.py:282-364: the same family as the second
                                                 reference-held witness (LayerNorm eps 1e-6 in the block, zero positions in the global pass)

Pinned: tests/test_oracle_golden.py::test_vggt_* / test_da3_* check every function against tests/golden/vggt_attention.pt / vggt_aggregator.pt /
da3_attention.pt, which tests/golden/make_golden.py::golden_vggt_attention / golden_vggt_aggregator / golden_da3_attention made by importing the
reference modules."""
import torch
import torch.nn.functional as F


def rope2d(t, pos, frequency=100.0):
    """t [B, H, N, d], pos [B, N, 2] integer (y, x)."""
    d = t.shape[-1]
    half = d // 2
    inv = 1.0 / (frequency ** (torch.arange(0, half, 2, dtype=torch.float32) / half))

    def one(feat, p):
        ang = p.to(torch.float32)[:, :, None] * inv[None, None, :]          # [B, N, half/2]
        ang = torch.cat([ang, ang], dim=-1).to(feat.dtype)[:, None]          # [B, 1, N, half]
        x1, x2 = feat[..., : half // 2], feat[..., half // 2:]
        return feat * ang.cos() + torch.cat([-x2, x1], dim=-1) * ang.sin()
    return torch.cat([one(t[..., :half], pos[..., 0]), one(t[..., half:], pos[..., 1])], dim=-1)


def attention(x, p, heads, pos=None, prefix=""):
    B, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, p[prefix + "qkv.weight"], p[prefix + "qkv.bias"]).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = F.layer_norm(q, (hd,), p[prefix + "q_norm.weight"], p[prefix + "q_norm.bias"], 1e-5)
    k = F.layer_norm(k, (hd,), p[prefix + "k_norm.weight"], p[prefix + "k_norm.bias"], 1e-5)
    if pos is not None:
        q, k = rope2d(q, pos), rope2d(k, pos)
    a = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, p[prefix + "proj.weight"], p[prefix + "proj.bias"])


def block(x, p, heads, pos=None, ln_eps=1e-5):
    """ln_eps: VGGT's blocks use nn.LayerNorm's default 1e-5; Depth Anything 3's DINOv2 blocks pass 1e-6
    (depth_anything_3/model/dinov2/layers/block.py:26-75) -- q_norm / k_norm keep 1e-5 in both."""
    h = F.layer_norm(x, (x.shape[-1],), p["norm1.weight"], p["norm1.bias"], ln_eps)
    x = x + p["ls1.gamma"] * attention(h, p, heads, pos, prefix="attn.")
    h = F.layer_norm(x, (x.shape[-1],), p["norm2.weight"], p["norm2.bias"], ln_eps)
    h = F.linear(F.gelu(F.linear(h, p["mlp.fc1.weight"], p["mlp.fc1.bias"])), p["mlp.fc2.weight"], p["mlp.fc2.bias"])
    return x + p["ls2.gamma"] * h


def da3_local_global_pair(tokens, p_local, p_global, heads, B, S, pos, ln_eps=1e-6):
    """Depth Anything 3's alternation (depth_anything_3/model/dinov2/vision_transformer.py:282-364): "local" attention per view on (B*S, N, C) with the
    grid positions, "global" attention across views on (B, S*N, C) with all-zero positions (pos_nodiff).  Pinned by tests/golden/da3_attention.pt."""
    N, C = tokens.shape[1], tokens.shape[2]
    t1 = block(tokens, p_local, heads, pos, ln_eps)
    t2 = block(t1.view(B, S * N, C), p_global, heads, torch.zeros(B, S * N, 2, dtype=pos.dtype), ln_eps)
    return t1, t2


def frame_global_pair(tokens, p_frame, p_global, heads, B, S, pos):
    """tokens [B*S, P, C], pos [B*S, P, 2] -> (frame block output [B*S, P, C], global block output [B, S*P, C])."""
    P, C = tokens.shape[1], tokens.shape[2]
    t1 = block(tokens, p_frame, heads, pos)
    t2 = block(t1.view(B, S * P, C), p_global, heads, pos.view(B, S * P, 2))
    return t1, t2


def aggregator(images, p, heads, depth, patch_size, n_register=4, aa_order=("frame", "global"), aa_block_size=1, rope=True):
    """images [B, S, 3, H, W] in [0, 1]; p: the reference aggregator's state dict (patch_embed = "conv") -> (list of [B, S, P, 2C], patch_start_idx)"""
    B, S, _, H, W = images.shape
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=images.dtype).view(1, 1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=images.dtype).view(1, 1, 3, 1, 1)
    x = ((images - mean) / std).view(B * S, 3, H, W)
    patches = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=patch_size).flatten(2).transpose(1, 2)

    def special(t):          # [1, 2, X, C] -> [B*S, X, C]
        return torch.cat([t[:, 0:1].expand(B, 1, -1, -1), t[:, 1:2].expand(B, S - 1, -1, -1)], dim=1).reshape(B * S, t.shape[2], t.shape[3])
    tokens = torch.cat([special(p["camera_token"]), special(p["register_token"]), patches], dim=1)
    start = 1 + n_register
    pos = None
    if rope:
        gh, gw = H // patch_size, W // patch_size
        grid = torch.cartesian_prod(torch.arange(gh), torch.arange(gw)) + 1
        pos = torch.cat([torch.zeros(start, 2, dtype=torch.long), grid], dim=0)[None].expand(B * S, -1, -1)
    P, C = tokens.shape[1], tokens.shape[2]
    sub = lambda prefix: {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}
    fi = gi = 0
    out = []
    for _ in range(depth // aa_block_size):
        inter = {"frame": [], "global": []}
        for kind in aa_order:
            for _ in range(aa_block_size):
                if kind == "frame":
                    tokens = block(tokens.reshape(B * S, P, C), sub(f"frame_blocks.{fi}."), heads, None if pos is None else pos.reshape(B * S, P, 2))
                    fi += 1
                else:
                    tokens = block(tokens.reshape(B, S * P, C), sub(f"global_blocks.{gi}."), heads, None if pos is None else pos.reshape(B, S * P, 2))
                    gi += 1
                inter[kind].append(tokens.reshape(B, S, P, C))
        out += [torch.cat([f, g], dim=-1) for f, g in zip(inter["frame"], inter["global"])]
    return out, start
