"""debug helper: does the w1 dq kernel of the selected library give finite output (S = 200, 2 heads)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import ops
B, H, S = 1, 2, 200
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, S, 3, H, 64, generator=g, device="cuda").to(torch.bfloat16)
q = qkv[:, :, 0].permute(0, 2, 1, 3).contiguous(); k = qkv[:, :, 1].permute(0, 2, 1, 3).contiguous(); v = qkv[:, :, 2].permute(0, 2, 1, 3)
dov = torch.randn(B, S, H * 64, generator=g, device="cuda").to(torch.bfloat16).view(B, S, H, 64).permute(0, 2, 1, 3)
o, lse = ops.attention_fwd_raw(q, k, v, split_mode=0)
ov = o.view(B, S, H, 64).permute(0, 2, 1, 3)
dq, dk = torch.full_like(q, float("nan")), torch.full_like(k, float("nan"))
dv = torch.full((B, S, H, 64), float("nan"), dtype=torch.bfloat16, device="cuda").permute(0, 2, 1, 3)
ref = torch.empty_like(q)
ops.attention_bwd_raw(q, k, v, ov, dov, lse, ref, dk, dv, split_mode=0)
ops.ATTN_W1 = {"dq"}
ops.attention_bwd_raw(q, k, v, ov, dov, lse, dq, dk, dv, split_mode=0)
torch.cuda.synchronize()
f = torch.isfinite(dq.float())
print(os.environ.get("VGPA_LIB", "product"), "finite frac", f.float().mean().item(), "max|diff|", (dq.float() - ref.float())[f].abs().max().item() if f.any() else None,
      "nan rows per head", (~f).any(-1).sum(-1).tolist())
print("new", dq[0, 0, :3, :6].float().tolist())
print("ref", ref[0, 0, :3, :6].float().tolist())
