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


# ---------------------------------------------------------------- sampling side (generate/CogVideoX-5B.py:18,70-77)
# PARITY UNPINNED like the rest of this file: restated from diffusers' scheduling_dpm_cogvideox.py
# (CogVideoXDPMScheduler.set_timesteps / get_variables / get_mult / step), written in the (alpha, sigma, lambda)
# parametrisation of the DPM-Solver++ papers rather than the scheduler's variable names, so that it is an
# independent derivation of the same update and not a transcription of the product's class.
def trailing_timesteps(num_inference_steps, num_train_timesteps=1000):
    """timestep_spacing='trailing': round(T, T - T/n, ...) - 1, descending."""
    import numpy as np
    t = np.round(np.arange(num_train_timesteps, 0, -num_train_timesteps / num_inference_steps)) - 1
    return torch.from_numpy(t.astype("int64"))


def dpm_step(abar, v, old_x0, t, t_back, x, num_inference_steps, noise, num_train_timesteps=1000):
    """One SDE-DPM-Solver++(2M) step on a v-prediction model.  abar float64 [T]; v model output; old_x0 the previous step's
    x0 prediction (None on the first step); t current timestep; t_back the PREVIOUS (larger) timestep or None; x current
    sample; noise [2, *x.shape] the two Gaussian draws the scheduler would make.  -> (x_prev, x0_pred), float64."""
    import math
    t = int(t)
    t_prev = t - num_train_timesteps // num_inference_steps
    a_t = float(abar[t])
    a_s = float(abar[t_prev]) if t_prev >= 0 else 1.0                       # final_alpha_cumprod = 1 (set_alpha_to_one)
    al_t, sg_t = math.sqrt(a_t), math.sqrt(1.0 - a_t)
    al_s, sg_s = math.sqrt(a_s), math.sqrt(1.0 - a_s)
    x = x.double()
    x0 = al_t * x - sg_t * v.double()                                       # v-prediction -> data prediction
    def lam(al, sg):                                                        # half log-SNR; -inf at zero terminal SNR (t = 999)
        return -math.inf if al == 0 else (math.inf if sg == 0 else math.log(al / sg))

    lam_t, lam_s = lam(al_t, sg_t), lam(al_s, sg_s)
    h = lam_s - lam_t
    decay = math.exp(-h)                                                    # 0 at the final step (h = inf)
    c_x = (sg_s / sg_t) * decay
    c_d = al_s * (1.0 - math.exp(-2.0 * h))                                 # = -alpha_s * expm1(-2h)
    c_n = sg_s * math.sqrt(1.0 - math.exp(-2.0 * h))
    if old_x0 is None or t_prev < 0:
        return c_x * x + c_d * x0 + c_n * noise[0].double(), x0
    a_b = float(abar[int(t_back)])
    r = (lam_t - lam(math.sqrt(a_b), math.sqrt(1.0 - a_b))) / h             # inf right after the zero-SNR step -> first order
    d = (1.0 + 0.5 / r) * x0 - (0.5 / r) * old_x0.double()                  # second-order multistep data estimate
    return c_x * x + c_d * d + c_n * noise[1].double(), x0
