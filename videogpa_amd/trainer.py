"""The DPO training step on MI355X: the harness around the reference's LightningModule logic
(`CogVideoXDPOTrainer`, train/CogVideoX-5B/03_train.py:87-213) without Lightning.

Step logic kept from the reference (`_shared_step`, :116-157): one (t, eps) per pair shared by win and lose,
v-prediction targets from the DPM scheduler, policy and frozen-reference predictions, Diffusion-DPO loss; AdamW
lr 5e-6 with 500-step cosine warm-up, clip 1.0, gradient accumulation, bf16 (:39-81,208-213,257-266).

MI355X-first differences (results are unchanged):
  * the reference model is the SAME weights with the adapter switched off (LoRA B=0 at init makes them identical
    to the reference's second 5B copy; the base is frozen in both) -- saves 11 GB and a second weight stream;
  * win and lose go through the transformer as ONE batch of 2B sequences (the reference calls it twice);
  * noising + targets and the loss are single fused passes over the paired layout [B,2,F,C,H,W];
  * activations stay resident (no per-block recompute) -- 288 GB HBM;
  * data parallel = one flat-buffer RCCL all-reduce of the LoRA gradients per optimizer step.
"""
import time
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .lora import LoraConfig, PeftModel, get_peft_model
from .loss import LossOutput, create_loss_strategy
from .optim import FlatAdamW, FlatParams
from .scheduler import CogVideoXDPMScheduler
from .transformer import CogVideoXTransformer3DModel

# same keys / defaults as the reference's DEFAULT_CONFIG (train/CogVideoX-5B/03_train.py:39-81) where they matter here
DEFAULT_CONFIG: Dict[str, Any] = {
    "model_path": "THUDM/CogVideoX-5b",
    "lora_rank": 64, "lora_alpha": 128, "lora_dropout": 0.0,
    "lora_target_modules": ["to_q", "to_k", "to_v", "to_out.0"],
    "beta": 1.0, "learning_rate": 5e-6, "weight_decay": 0.01, "warmup_steps": 500, "max_steps": 10000,
    "batch_size": 1, "accumulate_grad_batches": 2, "gradient_clip_val": 1.0,
    "enable_gradient_checkpointing": False,   # reference: True (80 GB GPUs); 288 GB keeps activations instead
    "metric_name": "consistency_score", "min_gap": 0.05, "motion_threshold": 1e-3,
    "log_every_n_steps": 10,
}


class CogVideoXDPOTrainer(nn.Module):
    def __init__(self, config: Dict[str, Any], transformer: Optional[nn.Module] = None, scheduler=None, separate_ref: bool = False):
        super().__init__()
        cfg = dict(DEFAULT_CONFIG)
        cfg.update(config)
        self.config = cfg
        if transformer is None:
            transformer = CogVideoXTransformer3DModel.from_pretrained(cfg["model_path"], subfolder="transformer", torch_dtype=torch.bfloat16)
        if isinstance(transformer, PeftModel):
            self.transformer = transformer
        else:
            lora_config = LoraConfig(r=cfg["lora_rank"], lora_alpha=cfg["lora_alpha"], lora_dropout=cfg["lora_dropout"],
                                     target_modules=cfg["lora_target_modules"])
            self.transformer = get_peft_model(transformer, lora_config)
        if cfg.get("enable_gradient_checkpointing"):
            self.transformer.enable_gradient_checkpointing()
        self.ref_transformer = None
        if separate_ref:  # the reference's layout: a second frozen copy (:110-111)
            import copy
            ref = copy.deepcopy(self.transformer)
            ref.requires_grad_(False).eval()
            self.ref_transformer = ref
        self.scheduler = scheduler if scheduler is not None else CogVideoXDPMScheduler()
        self.loss_fn = create_loss_strategy(strategy="dpo", beta=cfg["beta"])
        self.start_time = None
        self.global_step = 0

    # ------------------------------------------------------------------ forward pieces
    def _ref_forward(self, hs, prompt, tt):
        with torch.no_grad():
            if self.ref_transformer is not None:
                with self.ref_transformer.disable_adapter():
                    return self.ref_transformer(hs, encoder_hidden_states=prompt, timestep=tt, return_dict=True).sample
            with self.transformer.disable_adapter():
                return self.transformer(hs, encoder_hidden_states=prompt, timestep=tt, return_dict=True).sample

    def shared_step_paired(self, x_pair, prompt_emb, timesteps=None, noise=None, cond_pair=None) -> LossOutput:
        """x_pair [B,2,F,C,H,W] bf16 (win, lose); prompt_emb [B,L,4096]; optional fixed (timesteps, noise) for parity
        tests; cond_pair [B,2,F,Cc,H,W]: extra conditioning channels concatenated after noising (I2V, :135-136)."""
        B = x_pair.shape[0]
        dev = x_pair.device
        if timesteps is None:
            timesteps = torch.randint(0, self.scheduler.config.num_train_timesteps, (B,), device=dev)
        if noise is None:
            noise = torch.randn(x_pair[:, 0].shape, dtype=x_pair.dtype, device=dev)
        xt_pair, vt_pair = self.scheduler.noise_velocity_paired(x_pair.contiguous(), noise.contiguous(), timesteps)
        hs = xt_pair if cond_pair is None else torch.cat([xt_pair, cond_pair], dim=3)
        hs = hs.reshape(2 * B, *hs.shape[2:])
        prompt2 = prompt_emb.repeat_interleave(2, dim=0)
        t2 = timesteps.repeat_interleave(2)
        v_ref = self._ref_forward(hs, prompt2, t2)
        v_pol = self.transformer(hs, encoder_hidden_states=prompt2, timestep=t2, return_dict=True).sample
        v_pol = v_pol.reshape(B, 2, *v_pol.shape[1:])
        v_ref = v_ref.reshape(B, 2, *v_ref.shape[1:])
        lf = self.loss_fn
        loss, margin, wr, lr, acc, _ = ops.dpo_loss_paired(v_pol.contiguous(), v_ref.contiguous(), vt_pair, beta=lf.beta,
                                                            label_smoothing=lf.label_smoothing, loss_type=lf.loss_type)
        return LossOutput(loss=loss, reward_margin=margin.detach(), winner_reward=wr.detach(), loser_reward=lr.detach(), accuracy=acc.detach())

    @staticmethod
    def i2v_condition_pair(image_latent, num_frames):
        """I2V conditioning (train/CogVideoX-I2V-5B/03_train.py:127-136): the first-frame latent [B,1,C,H,W] (already VAE-encoded
        and scaled -- the VAE is a third-party network outside this path) zero-padded to `num_frames` frames, shared by win
        and lose -> [B,2,F,C,H,W], concatenated on the channel axis after noising."""
        B, one, C, H, W = image_latent.shape
        pad = image_latent.new_zeros(B, num_frames - one, C, H, W)
        cond = torch.cat([image_latent, pad], dim=1)
        return torch.stack([cond, cond], dim=1)

    def _shared_step(self, batch, timesteps=None, noise=None) -> LossOutput:
        """Reference-shaped entry: batch['x_win'/'x_lose'] [B,C,F,H,W], batch['prompt_emb'] (:116-157); an optional
        batch['image_latent'] [B,1,C,H,W] (or [B,C,1,H,W]) switches to the I2V form (32 input channels)."""
        if "x_pair" in batch:
            x_pair = batch["x_pair"]
        else:
            x_pair = torch.stack([batch["x_win"].permute(0, 2, 1, 3, 4), batch["x_lose"].permute(0, 2, 1, 3, 4)], dim=1)
        cond_pair = None
        if batch.get("image_latent") is not None:
            il = batch["image_latent"]
            if il.shape[1] != 1:
                il = il.permute(0, 2, 1, 3, 4)
            cond_pair = self.i2v_condition_pair(il.to(x_pair.dtype), x_pair.shape[2])
        return self.shared_step_paired(x_pair.contiguous(), batch["prompt_emb"], timesteps, noise, cond_pair=cond_pair)

    def training_step(self, batch, batch_idx=0):
        if self.start_time is None:
            self.start_time = time.time()
        out = self._shared_step(batch)
        logs = {"train/loss": out.loss.detach(), "train/reward_margin": out.reward_margin,
                "train/reward_accuracy": (out.reward_margin > 0).float().mean()}   # as the reference logs it (:170)
        return out.loss, logs

    def validation_step(self, batch, batch_idx=0):
        with torch.no_grad():
            out = self._shared_step(batch)
        return {"val/loss": out.loss, "val/reward_margin": out.reward_margin, "val/reward_accuracy": (out.reward_margin > 0).float().mean()}

    def configure_optimizers(self, process_group=None):
        cfg = self.config
        flat = FlatParams(self.transformer.parameters())
        return FlatAdamW(flat, lr=cfg["learning_rate"], weight_decay=cfg["weight_decay"], max_grad_norm=cfg["gradient_clip_val"],
                         warmup_steps=cfg.get("warmup_steps", 500), total_steps=cfg["max_steps"], process_group=process_group)


class DPOEngine:
    """Micro-step / optimizer-step driver: accumulation, flat-gradient all-reduce, fused clip + AdamW."""

    def __init__(self, trainer: CogVideoXDPOTrainer, process_group=None):
        self.trainer = trainer
        self.opt = trainer.configure_optimizers(process_group)
        self.accum = int(trainer.config.get("accumulate_grad_batches", 1))
        self.micro = 0
        # every rank must start from rank 0's adapter values (DDP does this broadcast at construction)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            dist.broadcast(self.opt.flat.flat, src=0, group=process_group)
        self.opt.zero_grad()

    def micro_step(self, batch) -> Dict[str, Any]:
        loss, logs = self.trainer.training_step(batch, self.micro)
        (loss / self.accum).backward()
        self.micro += 1
        if self.micro % self.accum == 0:
            pending = self.opt.all_reduce_grads()
            logs["lr"] = self.opt.step(pending)
            self.opt.zero_grad()
            self.trainer.global_step += 1
        return logs
