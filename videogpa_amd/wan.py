"""Wan2.2-TI2V flow-matching DPO step (train/Wan2.2-TI2V-5B/03_train.py:103-125,130-242) on the MI355X kernels.

The denoiser, `wan.modules.model.WanModel`, comes in the reference from an un-vendored sibling checkout
(`sys.path.insert(0, '../../Wan2.2')`, 03_train.py:43-46); here it is `videogpa_amd.wan_model.WanModel` (same constructor, module names and call
convention, restated from the published architecture -- parity unpinned, see oracle/wan.py).  This trainer takes the transformer as an argument --
that class, or any `nn.Module` with WanModel's call convention
        model(list of [C,F,H,W] latents, t=[B, seq_len], context=list of [L, D_text], seq_len=int) -> list of [C,F,H,W]
-- and provides the step the reference defines: shifted-sigma noising and the velocity target (fused HIP pass over the paired layout,
csrc/noise.hip), the clean first latent frame, the per-token timestep tensor with zeros on first-frame tokens, reference forwards before policy
forwards, PEFT-LoRA on the q/k/v/o linears (their A.B contractions run the MFMA kernels of csrc/lora.hip through LoraLinear), the Diffusion-DPO loss
kernel and the flat AdamW / all-reduce engine shared with the CogVideoX trainers.
"""
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .lora import LoraConfig, PeftModel, get_peft_model
from .loss import LossOutput, create_loss_strategy
from .optim import FlatAdamW, FlatParams

DEFAULT_CONFIG: Dict[str, Any] = {           # train/Wan2.2-TI2V-5B/03_train.py:54-97 where it matters here
    "learning_rate": 5e-6, "beta": 1.0, "max_steps": 10000, "warmup_steps": 500, "batch_size": 1, "accumulate_grad_batches": 2,
    "gradient_clip_val": 1.0, "weight_decay": 0.01, "num_train_timesteps": 1000, "shift": 5.0,
    "lora_rank": 64, "lora_alpha": 128.0, "lora_dropout": 0.0, "lora_target_modules": ["q", "k", "v", "o"],
    "patch_size": (1, 2, 2), "seed": 0, "tuned_gemms": True,
    "pair_batch": True,      # MI355X-first: win and lose as one batch through the denoiser (False = two calls, like the reference)
    # the reference recomputes every block in the backward (03_train.py:150-159, sized for 80 GB parts); 288 GB hold the activations of
    # the full-size pair step (measured 206 GB), which saves one forward in five: None leaves the model as the caller configured it,
    # True / False (+ stride k = recompute every k-th block) is applied to a model that has enable_gradient_checkpointing
    "enable_gradient_checkpointing": None, "gradient_checkpointing_stride": 1,
}


def ti2v_timestep_tensor(timesteps, latent_shape, seq_len, patch_size=(1, 2, 2)):
    """[B] timesteps -> [B, seq_len] per-token timesteps: 0 on the tokens of latent frame 0, t elsewhere (incl. padding up to
    seq_len).  create_ti2v_timestep_tensor + _create_mask (03_train.py:119-125,181-187), all samples at once."""
    _, f, h, w = latent_shape
    n0 = (-(-h // patch_size[1])) * (-(-w // patch_size[2]))            # tokens of one latent frame: mask[:, ::p, ::p]
    t = timesteps.to(torch.float32)[:, None].expand(-1, seq_len).clone()
    t[:, :n0] = 0.0
    return t


class WanDPOTrainer(nn.Module):
    def __init__(self, config: Dict[str, Any], transformer: nn.Module, ref_transformer: Optional[nn.Module] = None):
        super().__init__()
        cfg = dict(DEFAULT_CONFIG)
        cfg.update(config)
        self.config = cfg
        if isinstance(transformer, PeftModel):
            self.transformer = transformer
        else:
            self.transformer = get_peft_model(transformer, LoraConfig(r=cfg["lora_rank"], lora_alpha=cfg["lora_alpha"],
                                                                      lora_dropout=cfg["lora_dropout"], target_modules=cfg["lora_target_modules"]))
        # the reference loads a second frozen copy (:164-168); the same frozen base with the adapter switched off is
        # bit-identical and halves the weight memory; a separate copy can still be passed in
        self.ref_transformer = ref_transformer
        if ref_transformer is not None:
            ref_transformer.requires_grad_(False).eval()
        self.loss_fn = create_loss_strategy(strategy="dpo", beta=cfg["beta"])
        self.num_train_timesteps, self.shift, self.patch_size = cfg["num_train_timesteps"], cfg["shift"], tuple(cfg["patch_size"])
        self.global_step = 0
        self._rng = None
        self.after_reference = None      # DPOEngine's hook: a deferred optimizer step lands between the reference and the policy pass
        self.tuned_gemms = ops.use_tuned_gemms(bool(cfg.get("tuned_gemms", True)))
        if cfg["enable_gradient_checkpointing"] is not None:
            base = self.transformer.get_base_model()
            if not hasattr(base, "enable_gradient_checkpointing"):
                raise TypeError("enable_gradient_checkpointing is set but the transformer has no enable_gradient_checkpointing()")
            base.enable_gradient_checkpointing(bool(cfg["enable_gradient_checkpointing"]), stride=int(cfg["gradient_checkpointing_stride"]))

    def rng(self, device):
        if self._rng is None or self._rng.device != device:
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            self._rng = torch.Generator(device=device).manual_seed(int(self.config.get("seed", 0)) + rank)
        return self._rng

    def _compute_seq_len(self, z):
        _, _, f, h, w = z.shape
        return f * (h // self.patch_size[1]) * (w // self.patch_size[2])

    def _ref(self, x_list, **kw):
        with torch.no_grad():
            if self.ref_transformer is not None:
                return self.ref_transformer(x_list, **kw)
            with self.transformer.disable_adapter():
                return self.transformer(x_list, **kw)

    def _shared_step(self, batch, timesteps=None, noise=None) -> LossOutput:
        x_win, x_lose, prompt_emb = batch["x_win"], batch["x_lose"], batch["prompt_emb"]     # latents stay [B,C,F,H,W]
        image_latent = batch.get("image_latent")
        B, dev = x_win.shape[0], x_win.device
        seq_len = self._compute_seq_len(x_win)
        if timesteps is None:
            timesteps = torch.randint(1, self.num_train_timesteps, (B,), device=dev, generator=self.rng(dev))            # :198-201
        if noise is None:
            noise = torch.randn(x_win.shape, dtype=x_win.dtype, device=dev, generator=self.rng(dev))
        sigma = ops.flow_sigma(timesteps, self.num_train_timesteps, self.shift)
        x_pair = torch.stack([x_win, x_lose], dim=1).contiguous()
        # one pass: x_t = (1 - sigma) x + sigma eps for win and lose (fp32, as the reference's promoted result) and eps - x
        xt_pair, vt_pair = ops.flow_noise_velocity_paired(x_pair, noise.contiguous(), sigma)
        if image_latent is not None:                                                                                      # :209-211
            xt_pair[:, :, :, 0:1] = image_latent[:, None].to(xt_pair.dtype)
        t_batch = ti2v_timestep_tensor(timesteps, x_win.shape[1:], seq_len, self.patch_size)
        ctx = [prompt_emb[b] for b in range(B)]
        kw = dict(t=t_batch, context=ctx, seq_len=seq_len)
        xw_in = [xt_pair[b, 0] for b in range(B)]
        xl_in = [xt_pair[b, 1] for b in range(B)]
        if self.config.get("pair_batch", True):
            # win and lose through the model as ONE batch of 2B samples (the reference calls it twice, :227-233): rows are independent, results
            # unchanged, and every launch gets twice the rows -- at 18 480 tokens one sample leaves the last scheduling round of the attention
            # grids 16 % empty (1752 workgroups on 256 CUs), two samples 2 %
            kw2 = dict(t=torch.cat([t_batch, t_batch]), context=ctx + ctx, seq_len=seq_len)
            r = self._ref(xw_in + xl_in, **kw2)                                                                             # reference first (:227-229)
            v_wr, v_lr = torch.stack(r[:B]), torch.stack(r[B:])
            if self.after_reference is not None:
                self.after_reference()
            p = self.transformer(xw_in + xl_in, **kw2)
            v_w, v_l = torch.stack(p[:B]), torch.stack(p[B:])
        else:
            v_wr, v_lr = torch.stack(self._ref(xw_in, **kw)), torch.stack(self._ref(xl_in, **kw))                         # reference first (:227-229)
            if self.after_reference is not None:
                self.after_reference()
            v_w, v_l = torch.stack(self.transformer(xw_in, **kw)), torch.stack(self.transformer(xl_in, **kw))
        # WanModel returns fp32 (`[u.float() for u in x]` upstream) and the reference forms (pred - target) with an fp32 pred, so every squared error of
        # train/loss.py:73-77 is an fp32 quantity whatever the latents' stored dtype (a bf16 target promotes exactly): predictions stay fp32 here, the
        # velocity target is widened, and nothing is rounded before the square (round_diff is the CogVideoX bf16-prediction rule only)
        v_pol = torch.stack([v_w, v_l], dim=1).float().contiguous()
        v_ref = torch.stack([v_wr, v_lr], dim=1).float().contiguous()
        lf = self.loss_fn
        loss, margin, wr, lr, acc, _ = ops.dpo_loss_paired(v_pol, v_ref, vt_pair.float(), beta=lf.beta, label_smoothing=lf.label_smoothing,
                                                            loss_type=lf.loss_type, round_diff=False)
        return LossOutput(loss=loss, reward_margin=margin.detach(), winner_reward=wr.detach(), loser_reward=lr.detach(), accuracy=acc.detach())

    def training_step(self, batch, batch_idx=0):
        out = self._shared_step(batch)
        return out.loss, {"train/loss": out.loss.detach(), "train/reward_margin": out.reward_margin,
                          "train/reward_accuracy": (out.reward_margin > 0).float().mean()}

    def validation_step(self, batch, batch_idx=0):
        """never moves the adapters: DPOEngine's deferred optimizer step (`after_reference`) is held off, and a step still pending is an error (flush first)"""
        hook, self.after_reference = self.after_reference, None
        try:
            if getattr(getattr(hook, "__self__", None), "_pending", None) is not None:
                raise RuntimeError("validation_step with an optimizer step still pending in DPOEngine: call engine.flush() before validating")
            with torch.no_grad():
                out = self._shared_step(batch)
        finally:
            self.after_reference = hook
        return {"val/loss": out.loss, "val/reward_margin": out.reward_margin, "val/reward_accuracy": (out.reward_margin > 0).float().mean()}

    def configure_optimizers(self, process_group=None):
        cfg = self.config
        flat = FlatParams(self.transformer.parameters())
        return FlatAdamW(flat, lr=cfg["learning_rate"], weight_decay=cfg["weight_decay"], max_grad_norm=cfg["gradient_clip_val"],
                         warmup_steps=cfg.get("warmup_steps", 500), total_steps=cfg["max_steps"], process_group=process_group)
