"""Seeded inputs of BASELINE.json configs[0] (CogVideoX-5B T2V width, 2 transformer blocks, 13f x 64 x 64 paired
latents, LoRA r=8) shared by the fixture generator (tests/golden/make_cfg1_golden.py, CPU oracle) and the `-m gpu`
parity test (tests/test_gpu_cfg1.py, HIP path).  Everything comes from CPU torch generators, so both sides see the
same bits; weights are rounded to bf16 once (the HIP path computes in bf16) and the oracle runs them in fp32.

The step is train/CogVideoX-5B/03_train.py:116-157 at D = 3072 / 48 heads: the geometry the headline bench runs, at a
size the fp32 CPU oracle finishes in minutes.
"""
import torch

from oracle import cogvideox as ocv

FRAMES, HEIGHT, WIDTH, TEXT_LEN = 13, 64, 64, 226
TIMESTEP = 417
# (name, LoRA rank, std of the synthetic lora_B).  r=8 with B ~ N(0,1e-3) is BASELINE cfg1 / BASELINE.md section 3;
# r=64 walks the rank-64 (rp = 64 / 192) kernels of the headline config with a larger B so that the loss is clearly
# away from ln 2 and the lora_A gradients are well above bf16 noise.
VARIANTS = {"r8": (8, 1e-3), "r64": (64, 1e-2)}
N_SAMPLES = 256


def config():
    return ocv.CogVideoXConfig(num_layers=2, sample_height=HEIGHT, sample_width=WIDTH)


def base_state_dict(cfg):
    """bf16-rounded random base weights (diffusers names).  std 0.02 like BASELINE.md section 3; the AdaLN modulation
    linears get std 0.3 so that gates / scales are O(0.3) and both blocks contribute visibly to the prediction."""
    sd = ocv.init_state_dict(cfg, seed=0, std=0.02, mod_std=0.3)
    return {k: v.to(torch.bfloat16) for k, v in sd.items()}


def trained_like_state_dict(cfg, qk_gain=2.5):
    """base_state_dict with the QK-norm affines of a TRAINED model's shape (round 6; bench.py --weights trained_like): gains of qk_gain +- 20 %, three 3 x outlier
    channels, biases of 0.1 qk_gain -- sharp attention rows with scores spread over +-100 log2 units instead of the nearly flat rows unit gains give."""
    sd = base_state_dict(cfg)
    g = torch.Generator().manual_seed(7)
    for k in sorted(sd):
        if k.endswith(("attn1.norm_q.weight", "attn1.norm_k.weight")):
            w = qk_gain * (1 + 0.2 * torch.randn(sd[k].shape[0], generator=g))
            w[:3] *= 3.0
            sd[k] = w.to(torch.bfloat16)
        elif k.endswith(("attn1.norm_q.bias", "attn1.norm_k.bias")):
            sd[k] = (0.1 * qk_gain * torch.randn(sd[k].shape[0], generator=g)).to(torch.bfloat16)
    return sd


def lora_state_dict(cfg, variant):
    r, b_std = VARIANTS[variant]
    lora = ocv.init_lora(cfg, r=r, seed=1, b_std=b_std)
    return {k: v.to(torch.bfloat16).float() for k, v in lora.items()}, r      # fp32 values that are bf16-representable


def inputs():
    g = torch.Generator().manual_seed(1234)
    x_win = (0.7 * torch.randn(1, 16, FRAMES, HEIGHT, WIDTH, generator=g)).to(torch.bfloat16)
    x_lose = (0.7 * torch.randn(1, 16, FRAMES, HEIGHT, WIDTH, generator=g)).to(torch.bfloat16)
    prompt = (0.2 * torch.randn(1, TEXT_LEN, 4096, generator=g)).to(torch.bfloat16)
    noise = torch.randn(1, FRAMES, 16, HEIGHT, WIDTH, generator=g).to(torch.bfloat16)
    t = torch.tensor([TIMESTEP])
    return x_win, x_lose, prompt, t, noise


def sample_index(numel, tag):
    """Fixed pseudo-random flat indices into a tensor of `numel` elements."""
    g = torch.Generator().manual_seed(hash_tag(tag))
    return torch.randint(0, numel, (min(N_SAMPLES, numel),), generator=g)


def hash_tag(tag):
    h = 2166136261
    for ch in tag.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h
