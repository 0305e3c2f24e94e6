"""Denoising loop of the reference's generate path (generate/CogVideoX-5B.py:17-31,70-77 -> diffusers
CogVideoXPipeline.__call__): classifier-free guidance over a [uncond, cond] batch, 3D RoPE tables, the DPM scheduler,
with the MI355X transformer (RoPE kernel live) and a merged LoRA adapter.  Text encoding (T5) and latent decoding (VAE)
are third-party networks outside the hot path: the caller passes prompt embeddings in and gets latents out.
PARITY UNPINNED (diffusers not vendored): restated from pipeline_cogvideox.py / embeddings.get_3d_rotary_pos_embed."""
import torch


def rope_3d_tables(num_frames, grid_h, grid_w, head_dim=64, theta=10000.0, device="cuda"):
    """(cos, sin) fp32 [F*h*w, head_dim], pair-repeated, t/h/w split = d/4, 3d/8, 3d/8 (native-resolution grid)."""
    dim_t, dim_h, dim_w = head_dim // 4, head_dim // 8 * 3, head_dim // 8 * 3

    def axis(dim, n):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
        ang = torch.outer(torch.arange(n, dtype=torch.float32), freqs)
        return ang.cos().repeat_interleave(2, dim=1), ang.sin().repeat_interleave(2, dim=1)

    (ct, st), (ch, sh), (cw, sw) = axis(dim_t, num_frames), axis(dim_h, grid_h), axis(dim_w, grid_w)

    def combine(t, h, w):
        t = t[:, None, None, :].expand(-1, grid_h, grid_w, -1)
        h = h[None, :, None, :].expand(num_frames, -1, grid_w, -1)
        w = w[None, None, :, :].expand(num_frames, grid_h, -1, -1)
        return torch.cat([t, h, w], dim=-1).reshape(num_frames * grid_h * grid_w, head_dim)

    return combine(ct, ch, cw).to(device), combine(st, sh, sw).to(device)


def dynamic_guidance_scale(guidance_scale, num_inference_steps, t):
    """`use_dynamic_cfg=True` of the pipeline (generate/CogVideoX1.5-5B.py:85): the scale grows with a cosine ramp in the timestep
    VALUE t (as upstream writes it): 1 + g * (1 - cos(pi * ((n - t) / n) ** 5)) / 2."""
    import math
    return 1 + guidance_scale * ((1 - math.cos(math.pi * ((num_inference_steps - float(t)) / num_inference_steps) ** 5.0)) / 2)


@torch.no_grad()
def denoise(transformer, scheduler, prompt_embeds, negative_prompt_embeds=None, latent_frames=13, height=60, width=90,
            num_inference_steps=50, guidance_scale=6.0, generator=None, latents=None, step_noise=None, use_dynamic_cfg=False):
    """-> latents [B, F, C, H, W].  `latents` / `step_noise` ([steps, 2, B, F, C, H, W]) may be injected for parity tests.
    With patch_size_t (CogVideoX1.5) and a latent frame count that is not a multiple of it, the pipeline denoises with extra
    leading frames and drops them at the end; the same is done here."""
    cfg = transformer.config
    dev, dt = prompt_embeds.device, prompt_embeds.dtype
    B = prompt_embeds.shape[0]
    do_cfg = guidance_scale > 1.0
    if do_cfg:
        if negative_prompt_embeds is None:
            negative_prompt_embeds = torch.zeros_like(prompt_embeds)
        embeds = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
    else:
        embeds = prompt_embeds
    pt_ = cfg.patch_size_t or 1
    extra = (pt_ - latent_frames % pt_) % pt_
    latent_frames = latent_frames + extra
    shape = (B, latent_frames, cfg.in_channels, height, width)
    if latents is None:
        latents = torch.randn(shape, generator=generator, device=dev, dtype=dt)
    latents = latents * scheduler.init_noise_sigma
    rope = None
    if cfg.use_rotary_positional_embeddings:
        p, pt = cfg.patch_size, (cfg.patch_size_t or 1)
        rope = rope_3d_tables((latent_frames + pt - 1) // pt, height // p, width // p, cfg.attention_head_dim, device=dev)
    scheduler.set_timesteps(num_inference_steps, device=dev)
    ts = scheduler.timesteps
    old_x0 = None
    for i, t in enumerate(ts):
        inp = torch.cat([latents] * 2) if do_cfg else latents
        inp = scheduler.scale_model_input(inp, t)
        v = transformer(hidden_states=inp, encoder_hidden_states=embeds, timestep=t.expand(inp.shape[0]), image_rotary_emb=rope,
                        return_dict=False)[0].float()
        if do_cfg:
            g_now = dynamic_guidance_scale(guidance_scale, num_inference_steps, t.item()) if use_dynamic_cfg else guidance_scale
            v_u, v_c = v.chunk(2)
            v = v_u + g_now * (v_c - v_u)
        latents, old_x0 = scheduler.step(v, old_x0, t, ts[i - 1] if i > 0 else None, latents, generator=generator,
                                         noise=None if step_noise is None else step_noise[i])
        latents = latents.to(dt)
    return latents[:, extra:] if extra else latents
