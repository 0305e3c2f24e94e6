"""Training-side drop-in for diffusers' `CogVideoXDPMScheduler` (train/CogVideoX-5B/03_train.py:113,125,129-130,
154-155): `.config.num_train_timesteps`, `.add_noise`, `.get_velocity`, plus the fused paired form the MI355X
trainer uses (one HIP pass over [B,2,F,C,H,W]: videogpa_amd/csrc/noise.hip)."""
import json
import os

import torch

from . import ops
from .transformer import _Config


class CogVideoXDPMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 snr_shift_scale=1.0, rescale_betas_zero_snr=True, prediction_type="v_prediction", timestep_spacing="trailing", **kw):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError(beta_schedule)
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, snr_shift_scale=snr_shift_scale,
                              rescale_betas_zero_snr=rescale_betas_zero_snr, prediction_type=prediction_type,
                              timestep_spacing=timestep_spacing, **kw)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        abar = torch.cumprod(1.0 - betas, dim=0)
        abar = abar / (snr_shift_scale + (1 - snr_shift_scale) * abar)
        if rescale_betas_zero_snr:
            s = abar.sqrt()
            s0, sT = s[0].clone(), s[-1].clone()
            s = (s - sT) * (s0 / (s0 - sT))
            abar = s ** 2
        self.alphas_cumprod = abar
        self.final_alpha_cumprod = torch.tensor(1.0, dtype=torch.float64)   # set_alpha_to_one (CogVideoX scheduler_config)
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.num_inference_steps = None
        self._tabs = {}

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, "scheduler_config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        cfg.update(kw)
        return cls(**cfg)

    # ------------------------------------------------------------------ sampling side (generate/CogVideoX-5B.py:18,70-77)
    # PARITY UNPINNED: restated from diffusers' scheduling_dpm_cogvideox.py (SDE-DPM-Solver++ 2M on v-prediction).
    @classmethod
    def from_config(cls, config, **kw):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(kw)
        return cls(**cfg)

    def set_timesteps(self, num_inference_steps, device=None):
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        spacing = self.config.get("timestep_spacing", "trailing")
        if spacing == "trailing":
            ts = torch.round(torch.arange(T, 0, -T / num_inference_steps, dtype=torch.float64)) - 1
        elif spacing == "leading":
            ts = (torch.arange(0, num_inference_steps, dtype=torch.float64) * (T // num_inference_steps)).round().flip(0)
        else:
            ts = torch.linspace(0, T - 1, num_inference_steps, dtype=torch.float64).round().flip(0)
        self.timesteps = ts.to(torch.int64).to(device) if device is not None else ts.to(torch.int64)

    def scale_model_input(self, sample, timestep=None):
        return sample

    @staticmethod
    def _lambda(a):
        return 0.5 * torch.log(a / (1 - a))

    def step(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, eta=0.0, generator=None, noise=None,
             return_dict=False):
        """-> (prev_sample, pred_original_sample).  `noise` (two draws, [2, *sample.shape]) may be injected for parity tests."""
        T = self.config.num_train_timesteps
        t = int(timestep)
        prev_t = t - T // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        a_back = self.alphas_cumprod[int(timestep_back)] if timestep_back is not None else None
        x = sample.to(torch.float64)
        v = model_output.to(torch.float64)
        pred_x0 = a_t.sqrt() * x - (1 - a_t).sqrt() * v                      # v-prediction
        lamb, lamb_next = self._lambda(a_t), self._lambda(a_prev)
        h = lamb_next - lamb
        m1 = ((1 - a_prev) / (1 - a_t)).sqrt() * torch.exp(-h)
        m2 = torch.expm1(-2 * h) * a_prev.sqrt()
        m_noise = (1 - a_prev).sqrt() * (1 - torch.exp(-2 * h)).sqrt()

        def draw(i):
            if noise is not None:
                return noise[i].to(torch.float64)
            return torch.randn(sample.shape, generator=generator, device=sample.device, dtype=sample.dtype).to(torch.float64)

        prev = m1 * x - m2 * pred_x0 + m_noise * draw(0)
        if old_pred_original_sample is not None and prev_t >= 0:
            r = (lamb - self._lambda(a_back)) / h
            d = (1 + 1 / (2 * r)) * pred_x0 - (1 / (2 * r)) * old_pred_original_sample.to(torch.float64)
            prev = m1 * x - m2 * d + m_noise * draw(1)
        return prev.to(sample.dtype), pred_x0.to(sample.dtype)

    def tables(self, dtype, device):
        """fp32 device tables of sqrt(abar), sqrt(1-abar) with the table first cast to the sample dtype (as upstream)."""
        key = (dtype, str(device))
        if key not in self._tabs:
            a = self.alphas_cumprod.to(dtype)
            self._tabs[key] = ((a ** 0.5).float().to(device).contiguous(), ((1 - a) ** 0.5).float().to(device).contiguous())
        return self._tabs[key]

    def _pair(self, x, noise, timesteps):
        sa, sb = self.tables(x.dtype, x.device)
        xp = torch.stack([x, x], dim=1).contiguous()
        return ops.noise_velocity_paired(xp, noise.contiguous(), timesteps.to(torch.int64), sa, sb)

    def add_noise(self, original_samples, noise, timesteps):
        return self._pair(original_samples, noise, timesteps)[0][:, 0]

    def get_velocity(self, sample, noise, timesteps):
        return self._pair(sample, noise, timesteps)[1][:, 0]

    def noise_velocity_paired(self, x_pair, noise, timesteps):
        """x_pair [B,2,F,C,H,W] (win, lose), shared noise [B,F,C,H,W] -> (x_t pair, v-target pair) in one pass."""
        sa, sb = self.tables(x_pair.dtype, x_pair.device)
        return ops.noise_velocity_paired(x_pair, noise, timesteps.to(torch.int64), sa, sb)
