"""The reference's other three `_shared_step` functions, CPU oracle (plain torch).  TEST INFRASTRUCTURE ONLY.

  cogvideox15_pair_step   train/CogVideoX1.5-5B/03_train.py:118-186   (bf16 cast, conditional permute, even-crop F/H/W)
  i2v_pair_step           train/CogVideoX-I2V-5B/03_train.py:114-148   (image resize + encode + zero-pad + channel concat)
  wan_pair_step           train/Wan2.2-TI2V-5B/03_train.py:103-125,189-242 (shifted-sigma flow matching, clean first frame,
                                                                        per-token timesteps, reference forwards first)

The step logic is the reference's own code and is restated line by line; the networks it calls are not in the tree:
the CogVideoX transformer is oracle/cogvideox.py (parity unpinned, see there), the VAE encoder of the I2V step and
Wan2.2's `WanModel` (un-vendored sibling checkout `../../Wan2.2`, train/Wan2.2-TI2V-5B/03_train.py:43-46) are passed in
as callables.
"""
import torch
import torch.nn.functional as F

from . import cogvideox as ocv
from . import dpo, scheduler


def _four_forwards(sd, cfg, lora, xw_n, xl_n, prompt, t, lora_scale):
    v_w = ocv.forward(sd, cfg, xw_n, prompt, t, lora, lora_scale)
    v_l = ocv.forward(sd, cfg, xl_n, prompt, t, lora, lora_scale)
    with torch.no_grad():
        v_wr = ocv.forward(sd, cfg, xw_n, prompt, t, None)
        v_lr = ocv.forward(sd, cfg, xl_n, prompt, t, None)
    return v_w, v_l, v_wr, v_lr


def even_crop(x):
    """[B,F,C,H,W] -> F, H, W trimmed to even sizes (train/CogVideoX1.5-5B/03_train.py:131-142)."""
    B, Fr, C, H, W = x.shape
    return x[:, :Fr - Fr % 2, :, :H - H % 2, :W - W % 2]


def cogvideox15_pair_step(sd, cfg, lora, abar, x_win, x_lose, prompt_emb, t, noise, beta=1.0, lora_scale=2.0, compute_dtype=torch.float64):
    """x_win / x_lose as stored ([B,16,F,H,W]) or already [B,F,16,H,W]: permuted only when dim 1 is 16 (:127-129), values
    first rounded to bf16 (:122-124), then even-cropped; `noise` has the CROPPED shape [B,F',16,H',W']."""
    xw = x_win.to(torch.bfloat16).to(compute_dtype)
    xl = x_lose.to(torch.bfloat16).to(compute_dtype)
    prompt = prompt_emb.to(torch.bfloat16).to(compute_dtype)
    if xw.shape[1] == 16:
        xw, xl = xw.permute(0, 2, 1, 3, 4), xl.permute(0, 2, 1, 3, 4)
    xw, xl = even_crop(xw), even_crop(xl)
    xw_n = scheduler.add_noise(abar, xw, noise, t)
    xl_n = scheduler.add_noise(abar, xl, noise, t)
    v_w, v_l, v_wr, v_lr = _four_forwards(sd, cfg, lora, xw_n, xl_n, prompt, t, lora_scale)
    out = dpo.dpo_loss(v_w, v_l, v_wr, v_lr, scheduler.get_velocity(abar, xw, noise, t), scheduler.get_velocity(abar, xl, noise, t), beta=beta)
    out.update(v_win=v_w, v_lose=v_l, cropped_shape=tuple(xw.shape))
    return out


def i2v_condition(image_emb, like, image_encoder):
    """[B,3,h,w] image -> [B,F,C,H,W] condition: nearest resize to the latents' pixel size (F.interpolate default, :123),
    encode the single frame (:124-125), frame-major (:126), zero-pad to F frames (:127-128); zeros without an image (:130)."""
    B, Fr, C, H, W = like.shape
    if image_emb is None:
        return torch.zeros_like(like)
    img = F.interpolate(image_emb, size=(H * 8, W * 8))
    lat = image_encoder(img.unsqueeze(2)).permute(0, 2, 1, 3, 4).to(like.dtype)      # [B,1,C,H,W]
    pad = torch.zeros(B, Fr - 1, *lat.shape[2:], dtype=lat.dtype)
    return torch.cat([lat, pad], dim=1)


def i2v_pair_step(sd, cfg, lora, abar, x_win, x_lose, prompt_emb, t, noise, image_emb, image_encoder, beta=1.0, lora_scale=2.0):
    xw = x_win.permute(0, 2, 1, 3, 4)
    xl = x_lose.permute(0, 2, 1, 3, 4)
    cond = i2v_condition(image_emb, xw, image_encoder)
    xw_n = torch.cat([scheduler.add_noise(abar, xw, noise, t), cond], dim=2)           # channel concat (:135-136)
    xl_n = torch.cat([scheduler.add_noise(abar, xl, noise, t), cond], dim=2)
    v_w, v_l, v_wr, v_lr = _four_forwards(sd, cfg, lora, xw_n, xl_n, prompt_emb, t, lora_scale)
    out = dpo.dpo_loss(v_w, v_l, v_wr, v_lr, scheduler.get_velocity(abar, xw, noise, t), scheduler.get_velocity(abar, xl, noise, t), beta=beta)
    out.update(v_win=v_w, v_lose=v_l, cond=cond)
    return out


# ---------------------------------------------------------------- Wan2.2-TI2V flow matching
def flow_sigma(timestep, num_train_timesteps=1000, shift=5.0):
    s = timestep.float() / num_train_timesteps
    return shift * s / (1 + (shift - 1) * s)


def ti2v_timestep_tensor(timestep, z_shape, seq_len, patch_size=(1, 2, 2)):
    """Per-token timestep of ONE sample: 0 on the tokens of latent frame 0 (kept clean), `timestep` elsewhere, padded with
    `timestep` up to seq_len (create_ti2v_timestep_tensor :119-125 with the mask of _create_mask :181-187)."""
    z_dim, f, h, w = z_shape
    mask = torch.ones(f, h, w)
    mask[0] = 0.0
    timestep = float(timestep)                      # a 0-dim tensor on any device
    ts = (mask[:, ::patch_size[1], ::patch_size[2]] * timestep).flatten()
    return torch.cat([ts, ts.new_ones(seq_len - ts.numel()) * timestep]).unsqueeze(0)


def wan_pair_step(model, ref_model, x_win, x_lose, prompt_emb, t, noise, image_latent=None, beta=1.0, shift=5.0, num_train_timesteps=1000,
                  patch_size=(1, 2, 2)):
    """x_win / x_lose [B,C,F,H,W] (Wan latents stay channel-first); model(list of [C,F,H,W], t=[B,seq_len], context=list,
    seq_len=int) -> list of [C,F,H,W]."""
    B, C, Fr, H, W = x_win.shape
    seq_len = Fr * (H // patch_size[1]) * (W // patch_size[2])
    sigma = flow_sigma(t, num_train_timesteps, shift)
    sg = sigma.view(B, 1, 1, 1, 1).to(x_win.device, x_win.dtype)
    xw_n = (1.0 - sg) * x_win + sg * noise
    xl_n = (1.0 - sg) * x_lose + sg * noise
    if image_latent is not None:
        xw_n[:, :, 0:1] = image_latent
        xl_n[:, :, 0:1] = image_latent
    t_batch = torch.cat([ti2v_timestep_tensor(t[b], (C, Fr, H, W), seq_len, patch_size) for b in range(B)], dim=0).to(x_win.device)
    ctx = [prompt_emb[b] for b in range(B)]
    with torch.no_grad():                                                            # reference forwards FIRST (:227-229)
        v_wr = torch.stack(ref_model([xw_n[b] for b in range(B)], t=t_batch, context=ctx, seq_len=seq_len))
        v_lr = torch.stack(ref_model([xl_n[b] for b in range(B)], t=t_batch, context=ctx, seq_len=seq_len))
    v_w = torch.stack(model([xw_n[b] for b in range(B)], t=t_batch, context=ctx, seq_len=seq_len))
    v_l = torch.stack(model([xl_n[b] for b in range(B)], t=t_batch, context=ctx, seq_len=seq_len))
    out = dpo.dpo_loss(v_w, v_l, v_wr, v_lr, noise - x_win, noise - x_lose, beta=beta)
    out.update(t_batch=t_batch, x_win_noisy=xw_n, x_lose_noisy=xl_n, seq_len=seq_len, v_win=v_w, v_lose=v_l, v_win_ref=v_wr, v_lose_ref=v_lr)
    return out
