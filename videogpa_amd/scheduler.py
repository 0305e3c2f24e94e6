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
        self._tabs = {}

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, "scheduler_config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        cfg.update(kw)
        return cls(**cfg)

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
