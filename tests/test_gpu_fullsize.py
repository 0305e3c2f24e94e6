"""BASELINE config-2 sizes (S = 17 776 tokens, 48 heads, D = 3072) checked through size-independent properties -- no
O(S^2) reference needed: row-stochastic softmax, linearity in V, key-permutation invariance, gradient checksums,
scheduler identities, LayerNorm statistics, ln 2 at B = 0.  -m gpu only."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

S, H, TEXT = 17776, 48, 226


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from videogpa_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def qkv():
    g = torch.Generator(device="cuda").manual_seed(2)
    q, k, v = (torch.randn(1, H, S, 64, generator=g, device="cuda").to(torch.bfloat16) for _ in range(3))
    return q, k, v


def _o(ops, q, k, v):
    o, lse = ops.attention_fwd_raw(q, k, v)
    return o.view(1, S, H, 64).permute(0, 2, 1, 3), lse


def test_softmax_rows_sum_to_one_constant_v(ops, qkv):
    q, k, _ = qkv
    c = torch.linspace(-2, 2, 64, device="cuda").to(torch.bfloat16)
    o, _ = _o(ops, q, k, c.expand(1, H, S, 64).contiguous())
    assert (o.float() - c.float()).abs().max().item() <= 0.016     # one bf16 ulp at |c| <= 2


def test_linearity_in_v_and_key_permutation_invariance(ops, qkv):
    q, k, v = qkv
    g = torch.Generator(device="cuda").manual_seed(3)
    v2 = torch.randn(1, H, S, 64, generator=g, device="cuda").to(torch.bfloat16)
    o1, lse1 = _o(ops, q, k, v)
    o2, _ = _o(ops, q, k, v2)
    o12, _ = _o(ops, q, k, (v.float() + v2.float()).to(torch.bfloat16))
    assert (o12.float() - (o1.float() + o2.float())).abs().max().item() < 0.03
    perm = torch.randperm(S, generator=g, device="cuda")
    op, lsep = _o(ops, q, k[:, :, perm].contiguous(), v[:, :, perm].contiguous())
    assert (op.float() - o1.float()).abs().max().item() < 0.01
    assert (lsep - lse1).abs().max().item() < 2e-3


def test_backward_checksums(ops, qkv):
    """sum_k dV[k] = sum_q dO[q] (rows of P sum to 1);  sum_k dK[k] = 0 and dQ = 0 for identical keys (rows of dS sum to 0)."""
    q, k, v = qkv
    g = torch.Generator(device="cuda").manual_seed(4)
    do = torch.randn(1, H, S, 64, generator=g, device="cuda").to(torch.bfloat16)
    o, lse = ops.attention_fwd_raw(q, k, v)
    ov = o.view(1, S, H, 64).permute(0, 2, 1, 3)
    dq, dk, dv = (torch.empty(1, H, S, 64, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    ops.attention_bwd_raw(q, k, v, ov, do, lse, dq, dk, dv)
    sdv, sdo = dv.float().sum(2), do.float().sum(2)
    assert (sdv - sdo).abs().max().item() < 0.02 * sdo.abs().max().item() + 0.5
    col = dk.float().sum(2).abs().max().item()
    assert col < 0.02 * dk.float().abs().sum(2).max().item() + 0.05, col
    kc = k[:, :, :1].expand(1, H, S, 64).contiguous()                 # all keys identical -> uniform attention
    o2, lse2 = ops.attention_fwd_raw(q, kc, v)
    s_uniform = (ops.prescale_q(q).float() * kc.float()).sum(-1)       # every key gives the same (log2-domain) score
    assert (lse2 - (s_uniform + math.log2(S))).abs().max().item() < 2e-2
    ops.attention_bwd_raw(q, kc, v, o2.view(1, S, H, 64).permute(0, 2, 1, 3), do, lse2, dq, dk, dv)
    assert dq.float().abs().max().item() < 2e-3
    assert (o2.view(1, S, H, 64).permute(0, 2, 1, 3).float() - v.float().mean(2, keepdim=True)).abs().max().item() < 0.01


def test_scheduler_identity_and_loss_ln2_full_latent(ops):
    from videogpa_amd.scheduler import CogVideoXDPMScheduler
    g = torch.Generator(device="cuda").manual_seed(5)
    B = 2
    x = (0.7 * torch.randn(B, 2, 13, 16, 60, 90, generator=g, device="cuda")).to(torch.bfloat16)
    eps = torch.randn(B, 13, 16, 60, 90, generator=g, device="cuda").to(torch.bfloat16)
    t = torch.tensor([311, 999], device="cuda")
    sch = CogVideoXDPMScheduler()
    xt, vt = sch.noise_velocity_paired(x, eps, t)
    a = sch.alphas_cumprod.to(torch.bfloat16)[t.cpu()].float().cuda().view(B, 1, 1, 1, 1, 1)
    rec = a.sqrt() * xt.float() - (1 - a).sqrt() * vt.float()          # sqrt(abar) x_t - sqrt(1-abar) v = x
    assert (rec - x.float()).abs().max().item() < 0.06
    assert torch.equal(xt[1], eps[1:2].expand(2, -1, -1, -1, -1).reshape(xt[1].shape))   # zero terminal SNR: x_999 = eps
    out = ops.dpo_loss_paired(xt, xt.clone(), vt, beta=500.0)
    assert abs(out[0].item() - math.log(2.0)) < 1e-6


def test_layernorm_statistics_full_stream(ops):
    g = torch.Generator(device="cuda").manual_seed(6)
    x = (3 * torch.randn(2, S, 3072, generator=g, device="cuda") + 1.5).to(torch.bfloat16)
    n = ops.ln_modulate(x, torch.ones(3072, device="cuda"), torch.zeros(3072, device="cuda"), None, 0, 1e-5).float()
    assert n.mean(-1).abs().max().item() < 2e-3
    assert (n.var(-1, unbiased=False) - 1).abs().max().item() < 5e-3


def test_tail_round_split_at_the_headline_launch_shape(ops):
    """2 x 48 heads x 17 776 tokens = 6720 forward / dQ tasks and 13 344 dK/dV tasks: the launchers cut the leftover 64 / 32
    tasks into chunks (vgpa_attn_*_ws, automatic mode).  Same results as the single launches up to fp32 summation order."""
    g = torch.Generator(device="cuda").manual_seed(3)
    q, k, v, do = (torch.randn(2, H, S, 64, generator=g, device="cuda").to(torch.bfloat16) for _ in range(4))
    o0, lse0 = ops.attention_fwd_raw(q, k, v, split_mode=0)
    o1, lse1 = ops.attention_fwd_raw(q, k, v, split_mode=-1)
    diff = (o1.float() - o0.float()).abs()
    assert (diff > 0).any() and diff.max().item() < 4e-3            # the split rows differ (rounding), and only by an ulp of |o| ~ 0.1
    assert (lse1 - lse0).abs().max().item() < 1e-3
    outs = []
    ov = o0.view(2, S, H, 64).permute(0, 2, 1, 3)
    for sm in (0, -1):
        dq, dk, dv = (torch.empty(2, H, S, 64, dtype=torch.bfloat16, device="cuda") for _ in range(3))
        ops.attention_bwd_raw(q, k, v, ov, do, lse0, dq, dk, dv, split_mode=sm)
        outs.append((dq, dk, dv))
    for a, b in zip(*outs):
        scale = b.float().abs().max().item()
        assert (a.float() - b.float()).abs().max().item() < 0.02 * scale + 1e-4
        assert torch.isfinite(a).all()


# ---------------------------------------------------------------- BASELINE configs[3]: CogVideoX1.5-5B, S = 41 026 tokens
S4 = 41026        # 226 text + (20/2) x 48 x 85 video tokens (81f x 768 x 1360, 21 latent frames even-cropped to 20, patch_size_t 2)


def test_cfg4_attention_properties_at_41026_tokens(ops):
    """The attention kernels at the config-4 sequence length (6 heads are enough for the properties; the launch geometry per
    head is the same): row-stochastic softmax, sum_k dV = sum_q dO, sum_k dK = 0, and forward / backward consistency of the
    tail-split launch."""
    Hh = 6
    g = torch.Generator(device="cuda").manual_seed(12)
    q, k, v, do = (torch.randn(1, Hh, S4, 64, generator=g, device="cuda").to(torch.bfloat16) for _ in range(4))
    c = torch.linspace(-2, 2, 64, device="cuda").to(torch.bfloat16)
    o, _ = ops.attention_fwd_raw(q, k, c.expand(1, Hh, S4, 64).contiguous())
    assert (o.view(1, S4, Hh, 64).float() - c.float()).abs().max().item() <= 0.016
    o, lse = ops.attention_fwd_raw(q, k, v)
    ov = o.view(1, S4, Hh, 64).permute(0, 2, 1, 3)
    dq, dk, dv = (torch.empty(1, Hh, S4, 64, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    ops.attention_bwd_raw(q, k, v, ov, do, lse, dq, dk, dv)
    sdv, sdo = dv.float().sum(2), do.float().sum(2)
    assert (sdv - sdo).abs().max().item() < 0.02 * sdo.abs().max().item() + 0.8
    assert dk.float().sum(2).abs().max().item() < 0.02 * dk.float().abs().sum(2).max().item() + 0.05
    assert torch.isfinite(dq).all() and dq.float().abs().max().item() > 0


def test_cfg4_checkpointed_block_step_at_41026_tokens(ops):
    """One CogVideoX1.5-shaped block (D = 3072, 48 heads, patch_size_t = 2 geometry: [1,2,21,16,96,170] -> crop -> S = 41 026) through the
    1.5 step with per-block recompute (what configs[3] needs beyond ~22k tokens): loss = ln 2 exactly at B = 0, finite adapter
    gradients only in lora_B, and recompute on / off give the same gradients (to fp32 summation order)."""
    from videogpa_amd.lora import LoraConfig, get_peft_model
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    from videogpa_amd.transformer import COGVIDEOX_1_5_5B, CogVideoXTransformer3DModel
    torch.manual_seed(0)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("cuda"):
            model = CogVideoXTransformer3DModel(**dict(COGVIDEOX_1_5_5B, num_layers=1))
    finally:
        torch.set_default_dtype(prev)
    g = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("norm.weight") or n == "norm_final.weight" or ".norm_q.weight" in n or ".norm_k.weight" in n:
                p.fill_(1.0)
            elif n.endswith(".bias"):
                p.normal_(0.0, 0.02, generator=g)
            else:
                p.normal_(0.0, 0.02, generator=g)
    pm = get_peft_model(model, LoraConfig(r=64, lora_alpha=128, target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    x_pair = (0.7 * torch.randn(1, 2, 21, 16, 96, 170, generator=g, device="cuda")).to(torch.bfloat16)      # frame-major pair: not permuted, cropped to 20
    prompt = (0.2 * torch.randn(1, 226, 4096, generator=g, device="cuda")).to(torch.bfloat16)
    t = torch.tensor([500], device="cuda")
    eps = torch.randn(1, 20, 16, 96, 170, generator=g, device="cuda").to(torch.bfloat16)
    grads = []
    for ckpt in (True, False):
        tr = CogVideoXDPOTrainer({"beta": 1.0, "enable_gradient_checkpointing": ckpt}, transformer=pm)
        if not ckpt:
            pm.disable_gradient_checkpointing()
        tr.train()
        for p in pm.parameters():
            p.grad = None
        out = tr._shared_step({"x_pair": x_pair, "prompt_emb": prompt}, timesteps=t, noise=eps)
        assert abs(out.loss.item() - math.log(2.0)) < 1e-6              # B = 0: policy == reference bit for bit, also at S = 41 026
        out.loss.backward()
        gd = {n: p.grad.clone() for n, p in pm.named_parameters() if p.grad is not None}
        assert len(gd) == 8 and all(torch.isfinite(v).all() for v in gd.values())
        assert all(float(v.abs().max()) == 0 for n, v in gd.items() if "lora_A" in n)       # dA = dT^T x with dT = dy B = 0
        assert any(float(v.abs().max()) > 0 for n, v in gd.items() if "lora_B" in n)
        grads.append(gd)
    for n in grads[0]:       # same kernels and inputs; the token-contracted fp32 atomics of lora_grad leave summation-order noise
        a, b = grads[0][n].double(), grads[1][n].double()
        assert (a - b).abs().max().item() <= 1e-3 * b.abs().max().item() + 1e-12, n
