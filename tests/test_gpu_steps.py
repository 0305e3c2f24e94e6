"""The reference's other three `_shared_step` functions on the HIP path vs the oracle restatements (oracle/steps.py):
CogVideoX1.5 (train/CogVideoX1.5-5B/03_train.py:118-186), CogVideoX-I2V (train/CogVideoX-I2V-5B/03_train.py:114-148) and
the Wan2.2-TI2V step logic (train/Wan2.2-TI2V-5B/03_train.py:189-242) around a stand-in model.  -m gpu only.

Tolerances: loss within 1e-3 of the fp64 oracle (north_star); LoRA gradients within 6 % of each tensor's max |grad|
(bf16 GEMM chain, as tests/test_gpu_model.py)."""
import math

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import cogvideox as ocv
from oracle import scheduler as osch
from oracle import steps as ost


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _model(r=4, b_std=0.05, seed=0, **extra):
    from videogpa_amd.lora import LoraConfig, get_peft_model
    from videogpa_amd.transformer import CogVideoXTransformer3DModel
    kw = dict(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=32, text_embed_dim=48, sample_width=12,
              sample_height=8, sample_frames=13, max_text_seq_length=6)
    kw.update(extra)
    cfg = ocv.CogVideoXConfig(**kw)
    sd = {k: v.to(torch.bfloat16) for k, v in ocv.init_state_dict(cfg, seed=seed, std=0.05, mod_std=0.2).items()}
    model = CogVideoXTransformer3DModel(use_rotary_positional_embeddings=True, **kw)
    model.load_state_dict(sd, strict=True)
    model = model.to(device="cuda", dtype=torch.bfloat16)
    pm = get_peft_model(model, LoraConfig(r=r, lora_alpha=2 * r, target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    lora = ocv.init_lora(cfg, r=r, seed=seed + 1, b_std=b_std)
    own = pm.state_dict()
    for k, v in lora.items():
        own[k[:-len(".weight")] + ".default.weight"].copy_(v)
    return cfg, {k: v.double() for k, v in sd.items()}, {k: v.to(torch.bfloat16).double() for k, v in lora.items()}, pm


def _check_grads(pm, lora_ref):
    own = dict(pm.named_parameters())
    n = 0
    for k, v in lora_ref.items():
        p = own[k[:-len(".weight")] + ".default.weight"]
        assert p.grad is not None, k
        err = (p.grad.double().cpu() - v.grad).abs().max().item()
        assert err < 0.06 * v.grad.abs().max().item() + 1e-5, (k, err, v.grad.abs().max().item())
        n += 1
    return n


def test_cogvideox15_step_crop_cast_permute_with_backward():
    """patch_size_t = 2 model; odd F, H, W in the stored latents -> even-crop; fp32 inputs -> bf16 cast; [B,16,F,H,W] -> permute."""
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    cfg, sd64, lora64, pm = _model(patch_size_t=2, patch_bias=False)
    tr = CogVideoXDPOTrainer({"beta": 1.0}, transformer=pm)
    g = torch.Generator().manual_seed(5)
    B = 2
    xw = 0.7 * torch.randn(B, 16, 5, 9, 13, generator=g)          # fp32 on purpose: the 1.5 step casts to bf16 itself
    xl = 0.7 * torch.randn(B, 16, 5, 9, 13, generator=g)
    txt = 0.5 * torch.randn(B, 6, cfg.text_embed_dim, generator=g)
    t = torch.tensor([417, 80])
    eps = torch.randn(B, 4, 16, 8, 12, generator=g).to(torch.bfloat16)     # the CROPPED shape
    out = tr._shared_step({"x_win": xw.cuda(), "x_lose": xl.cuda(), "prompt_emb": txt.cuda()}, timesteps=t.cuda(), noise=eps.cuda())
    out.loss.backward()
    lr = {k: v.clone().requires_grad_(True) for k, v in lora64.items()}
    ref = ost.cogvideox15_pair_step(sd64, cfg, lr, osch.alphas_cumprod(), xw, xl, txt, t, eps.double(), beta=1.0)
    assert ref["cropped_shape"] == (B, 4, 16, 8, 12)
    assert abs(out.loss.item() - ref["loss"].item()) < 1e-3, (out.loss.item(), ref["loss"].item())
    ref["loss"].backward()
    assert _check_grads(pm, lr) == 2 * 4 * cfg.num_layers
    # already frame-major input ([B,F,16,H,W]: dim 1 != 16) is NOT permuted (:127-129) -> same result
    out2 = tr._shared_step({"x_win": xw.permute(0, 2, 1, 3, 4).contiguous().cuda(), "x_lose": xl.permute(0, 2, 1, 3, 4).contiguous().cuda(),
                            "prompt_emb": txt.cuda()}, timesteps=t.cuda(), noise=eps.cuda())
    assert out2.loss.item() == out.loss.item()


class _ToyEncoder:
    """Stand-in for vae.encode(x).latent_dist.sample() * scaling_factor: 8x8 average pool + a fixed 3 -> 16 channel mix."""

    def __init__(self):
        g = torch.Generator().manual_seed(3)
        self.mix = torch.randn(16, 3, generator=g) * 0.5

    def __call__(self, img):                                   # [B,3,1,H,W] -> [B,16,1,H/8,W/8]
        x = torch.nn.functional.avg_pool2d(img[:, :, 0].float(), 8)
        return torch.einsum("oc,bchw->bohw", self.mix.to(x.device), x)[:, :, None]


def test_i2v_step_image_emb_encoder_zero_condition_and_errors():
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    cfg, sd64, lora64, pm = _model(in_channels=32, use_learned_positional_embeddings=True, seed=4)
    enc = _ToyEncoder()
    tr = CogVideoXDPOTrainer({"beta": 1.0}, transformer=pm, image_encoder=enc)
    g = torch.Generator().manual_seed(6)
    B = 1
    xw = (0.7 * torch.randn(B, 16, 4, 8, 12, generator=g)).to(torch.bfloat16)
    xl = (0.7 * torch.randn(B, 16, 4, 8, 12, generator=g)).to(torch.bfloat16)
    txt = (0.5 * torch.randn(B, 6, cfg.text_embed_dim, generator=g)).to(torch.bfloat16)
    img = torch.rand(B, 3, 40, 50, generator=g).to(torch.bfloat16)     # any size: resized to (H*8, W*8) = (64, 96), nearest
    t = torch.tensor([650])
    eps = torch.randn(B, 4, 16, 8, 12, generator=g).to(torch.bfloat16)
    batch = {"x_win": xw.cuda(), "x_lose": xl.cuda(), "prompt_emb": txt.cuda(), "image_emb": img.cuda()}
    out = tr._shared_step(batch, timesteps=t.cuda(), noise=eps.cuda())
    out.loss.backward()
    lr = {k: v.clone().requires_grad_(True) for k, v in lora64.items()}
    enc64 = lambda x: enc(x).to(torch.bfloat16).double()                # the product casts the encoder output to the latent dtype
    ref = ost.i2v_pair_step(sd64, cfg, lr, osch.alphas_cumprod(), xw.double(), xl.double(), txt.double(), t, eps.double(), img.double(), enc64)
    assert abs(out.loss.item() - ref["loss"].item()) < 1e-3, (out.loss.item(), ref["loss"].item())
    ref["loss"].backward()
    _check_grads(pm, lr)
    # no image in the batch -> zero condition (train/CogVideoX-I2V-5B/03_train.py:130), still a valid 32-channel step
    out0 = tr._shared_step({k: v for k, v in batch.items() if k != "image_emb"}, timesteps=t.cuda(), noise=eps.cuda())
    ref0 = ost.i2v_pair_step(sd64, cfg, lora64, osch.alphas_cumprod(), xw.double(), xl.double(), txt.double(), t, eps.double(), None, None)
    assert abs(out0.loss.item() - ref0["loss"].item()) < 1e-3
    # pre-encoded latent is the same as running the encoder
    il = enc(torch.nn.functional.interpolate(img.float(), size=(64, 96)).unsqueeze(2)).to(torch.bfloat16)
    out1 = tr._shared_step({"x_win": xw.cuda(), "x_lose": xl.cuda(), "prompt_emb": txt.cuda(), "image_latent": il.cuda()}, timesteps=t.cuda(), noise=eps.cuda())
    assert abs(out1.loss.item() - out.loss.item()) < 2e-4
    # error behaviour: image_emb without an encoder; image condition handed to a 16-channel (T2V) model
    tr_noenc = CogVideoXDPOTrainer({"beta": 1.0}, transformer=pm)
    with pytest.raises(RuntimeError, match="image_encoder"):
        tr_noenc._shared_step(batch, timesteps=t.cuda(), noise=eps.cuda())
    _, _, _, pm_t2v = _model()
    with pytest.raises(RuntimeError, match="input channels"):
        CogVideoXDPOTrainer({"beta": 1.0}, transformer=pm_t2v)._shared_step(batch, timesteps=t.cuda(), noise=eps.cuda())


class _ToyWan(nn.Module):
    """WanModel's call convention (list of [C,F,H,W], t=[B,seq_len], context=list, seq_len) on a tiny token mixer with
    q/k/v/o linears (the reference's LoRA targets) and a per-token timestep modulation."""

    def __init__(self, C=8, dim=64, text_dim=48, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.C, self.dim = C, dim
        self.embed = nn.Linear(C * 4, dim)
        self.text = nn.Linear(text_dim, dim)
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim) for _ in range(4))
        self.head = nn.Linear(dim, C * 4)
        for p in self.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * 0.08)

    def forward(self, x_list, t, context, seq_len):
        outs = []
        for b, x in enumerate(x_list):
            C, Fr, H, W = x.shape
            dt = self.embed.weight.dtype
            tok = x.to(dt).reshape(C, Fr, H // 2, 2, W // 2, 2).permute(1, 2, 4, 0, 3, 5).reshape(Fr * (H // 2) * (W // 2), C * 4)
            h = self.embed(tok) * (1.0 + torch.sin(t[b, :tok.shape[0], None].to(dt) * 0.01))
            c = self.text(context[b].to(dt))
            a = torch.softmax((self.q(h) @ self.k(c).t()).float() / 8.0, dim=-1).to(dt) @ self.v(c)
            h = h + self.o(a)
            y = self.head(h).reshape(Fr, H // 2, W // 2, C, 2, 2).permute(3, 0, 1, 4, 2, 5).reshape(C, Fr, H, W)
            outs.append(y)
        return outs


@pytest.mark.parametrize("latent_dtype", [torch.bfloat16, torch.float32])
def test_wan_ti2v_step_logic_matches_oracle(latent_dtype):
    """latents as stored by the dataset: bf16, or fp32 (the reference's encoded Wan latents; every squared error is then formed from fp32
    values, train/Wan2.2-TI2V-5B/03_train.py:236-242 with train/loss.py:73-77)"""
    from videogpa_amd.wan import WanDPOTrainer, ti2v_timestep_tensor
    from videogpa_amd.lora import LoraConfig, get_peft_model
    B, C, Fr, H, W = 2, 8, 3, 8, 12
    toy = _ToyWan(C=C).to("cuda", torch.bfloat16)
    pm = get_peft_model(toy, LoraConfig(r=4, lora_alpha=8, target_modules=["q", "k", "v", "o"]))
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for n, p in pm.named_parameters():
            if ".lora_B." in n:
                p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(torch.bfloat16).float())
            elif ".lora_A." in n:
                p.copy_(p.to(torch.bfloat16).float())
    tr = WanDPOTrainer({"beta": 1.0, "lora_rank": 4, "lora_alpha": 8.0}, transformer=pm)
    xw = (0.7 * torch.randn(B, C, Fr, H, W, generator=g)).to(torch.bfloat16)
    xl = (0.7 * torch.randn(B, C, Fr, H, W, generator=g)).to(torch.bfloat16)
    txt = (0.5 * torch.randn(B, 5, 48, generator=g)).to(torch.bfloat16)
    il = (0.7 * torch.randn(B, C, 1, H, W, generator=g)).to(torch.bfloat16)
    t = torch.tensor([417, 999])
    eps = torch.randn(B, C, Fr, H, W, generator=g).to(torch.bfloat16)
    ld = latent_dtype
    out = tr._shared_step({"x_win": xw.to(ld).cuda(), "x_lose": xl.to(ld).cuda(), "prompt_emb": txt.cuda(), "image_latent": il.to(ld).cuda()},
                          timesteps=t.cuda(), noise=eps.to(ld).cuda())
    out.loss.backward()
    # oracle: the same toy model in fp64 on the CPU, adapter on / off, through oracle.steps.wan_pair_step
    import copy
    cpu_pol = copy.deepcopy(pm).to("cpu", torch.float64)
    cpu_ref = copy.deepcopy(cpu_pol)
    for m in cpu_ref.lora_layers():
        m.disable_adapters = True
    ref = ost.wan_pair_step(cpu_pol, cpu_ref, xw.double(), xl.double(), txt.double(), t, eps.double(), image_latent=il.double(), beta=1.0)
    assert ref["seq_len"] == Fr * (H // 2) * (W // 2)
    tb = ti2v_timestep_tensor(t.cuda(), (C, Fr, H, W), ref["seq_len"])
    assert torch.equal(tb.cpu(), ref["t_batch"])                                    # zeros on first-frame tokens, t elsewhere
    assert abs(out.loss.item() - ref["loss"].item()) < 1e-3, (out.loss.item(), ref["loss"].item())
    ref["loss"].backward()
    gp = {n: p.grad for n, p in pm.named_parameters() if p.grad is not None}
    gr = {n: p.grad for n, p in cpu_pol.named_parameters() if p.grad is not None}
    assert set(gp) == set(gr) and len(gp) == 8 and all(".lora_" in n for n in gp)
    for n in gp:
        err = (gp[n].double().cpu() - gr[n]).abs().max().item()
        assert err < 0.06 * gr[n].abs().max().item() + 1e-6, (n, err, gr[n].abs().max().item())
    assert math.isfinite(out.loss.item())


def test_attention_processor_seam_matches_oracle_block_attention():
    """diffusers' `attn.set_processor(P)` seam: P(attn, hidden_states, encoder_hidden_states, image_rotary_emb=...)
    -> (hidden_states, encoder_hidden_states), text tokens first, LoRA wrappers honoured; against the oracle's attention half
    of a block (oracle.cogvideox.block_forward capture)."""
    from videogpa_amd.attn_processor import MI355XCogVideoXAttnProcessor, install
    cfg, sd64, lora64, pm = _model(b_std=0.05, seed=2)
    assert install(pm) == cfg.num_layers
    blk = pm.transformer_blocks[0]
    P = blk.attn1.processor
    g = torch.Generator().manual_seed(8)
    B, Sv, Lt, D = 2, 3 * 4 * 6, 6, cfg.inner_dim
    hid = torch.randn(B, Sv, D, generator=g).to(torch.bfloat16)
    enc = torch.randn(B, Lt, D, generator=g).to(torch.bfloat16)
    cos, sin = ocv.rope_3d_tables(3, 4, 6, 64)
    for rope in (None, (cos, sin)):
        h_out, e_out = P(blk.attn1, hid.cuda(), enc.cuda(), image_rotary_emb=None if rope is None else (rope[0].cuda(), rope[1].cuda()))
        assert h_out.shape == (B, Sv, D) and e_out.shape == (B, Lt, D)
        # oracle: q, k, v -> QK-norm -> RoPE -> SDPA -> to_out on cat([enc, hid]) with the same adapters
        b = "transformer_blocks.0."
        x = torch.cat([enc, hid], dim=1).double()
        q = ocv._lora_linear(x, sd64, lora64, b + "attn1.to_q", 2.0)
        k = ocv._lora_linear(x, sd64, lora64, b + "attn1.to_k", 2.0)
        v = ocv._lora_linear(x, sd64, lora64, b + "attn1.to_v", 2.0)
        H = cfg.num_attention_heads
        q, k, v = (t.view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
        q = ocv.layer_norm(q, sd64[b + "attn1.norm_q.weight"], sd64[b + "attn1.norm_q.bias"], cfg.qk_norm_eps)
        k = ocv.layer_norm(k, sd64[b + "attn1.norm_k.weight"], sd64[b + "attn1.norm_k.bias"], cfg.qk_norm_eps)
        if rope is not None:
            q = torch.cat([q[:, :, :Lt], ocv.apply_rotary_emb(q[:, :, Lt:], cos.double(), sin.double())], dim=2)
            k = torch.cat([k[:, :, :Lt], ocv.apply_rotary_emb(k[:, :, Lt:], cos.double(), sin.double())], dim=2)
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, -1, D)
        o = ocv._lora_linear(o, sd64, lora64, b + "attn1.to_out.0", 2.0)
        got = torch.cat([e_out, h_out], dim=1).double().cpu()
        assert (got - o).abs().max().item() < 0.03 * o.abs().max().item()
    with pytest.raises(NotImplementedError):
        P(blk.attn1, hid.cuda(), enc.cuda(), attention_mask=torch.ones(1))
