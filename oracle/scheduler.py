"""CogVideoXDPMScheduler training helpers (add_noise / get_velocity), CPU oracle.

PARITY UNPINNED: diffusers (requirements.txt:20, >=0.31.0) is not vendored.
Call sites restated: train/CogVideoX-5B/03_train.py:113,125,129-130,154-155.
Published algorithm (diffusers scheduling_dpm_cogvideox.py / scheduling_ddim_cogvideox.py):
  betas   = linspace(sqrt(beta_start), sqrt(beta_end), T, float64)**2      ("scaled_linear")
  abar    = cumprod(1 - betas)
  abar    = abar / (snr_shift_scale + (1 - snr_shift_scale) * abar)
  rescale_betas_zero_snr: sqrt(abar) shifted/scaled so sqrt(abar_T-1) = 0, sqrt(abar_0) kept
  add_noise   : sqrt(abar_t) x + sqrt(1 - abar_t) eps
  get_velocity: sqrt(abar_t) eps - sqrt(1 - abar_t) x
"""
import torch


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                   snr_shift_scale=1.0, rescale_betas_zero_snr=True):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
    abar = torch.cumprod(1.0 - betas, dim=0)
    abar = abar / (snr_shift_scale + (1 - snr_shift_scale) * abar)
    if rescale_betas_zero_snr:
        s = abar.sqrt()
        s0, sT = s[0].clone(), s[-1].clone()
        s = (s - sT) * (s0 / (s0 - sT))
        abar = s ** 2
    return abar  # float64 [T]


def _coeffs(abar, t, like):
    a = abar.to(dtype=like.dtype)[t]
    sa, sb = a.sqrt(), (1 - a).sqrt()
    while sa.dim() < like.dim():
        sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
    return sa, sb


def add_noise(abar, x, noise, t):
    sa, sb = _coeffs(abar, t, x)
    return sa * x + sb * noise


def get_velocity(abar, x, noise, t):
    sa, sb = _coeffs(abar, t, x)
    return sa * noise - sb * x
