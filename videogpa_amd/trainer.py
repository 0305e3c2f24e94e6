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
import os
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
    # keep 23 % less per block (LN output and normalised q / k made again in the backward; results bit-identical): True / False, or "auto" = decided from the
    # first batch's shape against the device's memory (memory_policy below) and logged -- cfg2 runs full, cfg3 at batch 2 and cfg4 (S = 41 026) run lean
    "lean_activations": "auto",
    "metric_name": "consistency_score", "min_gap": 0.05, "motion_threshold": 1e-3,
    "log_every_n_steps": 10,
    "tuned_gemms": True,                      # pick the hipBLASLt solutions of videogpa_amd/tuned/ (ops.use_tuned_gemms); False = the library's default heuristic
    "seed": 0,                                # (t, eps) stream = seed + rank: every rank draws its own
}


# What the reference's four training scripts change against the T2V defaults (their DEFAULT_CONFIG dicts and optimizer lines):
#   I2V  train/CogVideoX-I2V-5B/03_train.py:59-60   batch 2, no accumulation
#   1.5  train/CogVideoX1.5-5B/03_train.py:54,210   1500 steps, AdamW weight_decay 1e-3 (hard-coded there)
#   Wan  train/Wan2.2-TI2V-5B/03_train.py:54-97     q/k/v/o targets, flow-matching shift 5 (videogpa_amd/wan.py)
VARIANT_CONFIGS: Dict[str, Dict[str, Any]] = {
    "t2v": {},
    "i2v": {"batch_size": 2, "accumulate_grad_batches": 1},
    "1.5": {"max_steps": 1500, "weight_decay": 1e-3},
}


def variant_config(variant: str, **overrides) -> Dict[str, Any]:
    """DEFAULT_CONFIG of the named reference script ("t2v", "i2v", "1.5") as a dict for CogVideoXDPOTrainer / fit."""
    if variant not in VARIANT_CONFIGS:
        raise ValueError(f"unknown variant {variant!r}; one of {sorted(VARIANT_CONFIGS)}")
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(VARIANT_CONFIGS[variant])
    cfg.update(overrides)
    return cfg


class CogVideoXDPOTrainer(nn.Module):
    def __init__(self, config: Dict[str, Any], transformer: Optional[nn.Module] = None, scheduler=None, separate_ref: bool = False,
                 image_encoder=None):
        """image_encoder: I2V only -- callable `[B,3,1,H,W] image in the dataset's value range -> [B,C_lat,1,h,w]` standing
        for `vae.encode(...).latent_dist.sample() * vae.config.scaling_factor` (train/CogVideoX-I2V-5B/03_train.py:124-125);
        the VAE itself is a third-party network outside this path."""
        super().__init__()
        cfg = dict(DEFAULT_CONFIG)
        cfg.update(config)
        self.config = cfg
        self.image_encoder = image_encoder
        local_model = isinstance(cfg.get("model_path"), str) and os.path.isdir(cfg["model_path"])
        if transformer is None:
            if not local_model:
                raise FileNotFoundError(
                    f"model_path {cfg['model_path']!r} is not a local directory: there is no hub access on this path; pass a local "
                    "diffusers checkpoint directory (transformer/ + scheduler/) or a constructed `transformer`")
            transformer = CogVideoXTransformer3DModel.from_pretrained(cfg["model_path"], subfolder="transformer", torch_dtype=torch.bfloat16)
        if isinstance(transformer, PeftModel):
            self.transformer = transformer
        else:
            lora_config = LoraConfig(r=cfg["lora_rank"], lora_alpha=cfg["lora_alpha"], lora_dropout=cfg["lora_dropout"],
                                     target_modules=cfg["lora_target_modules"])
            self.transformer = get_peft_model(transformer, lora_config)
        if cfg.get("enable_gradient_checkpointing"):
            stride = int(cfg.get("gradient_checkpointing_stride", 1))
            if stride != 1:     # the stride is this package's extension; a diffusers / PEFT model takes no arguments here
                self.transformer.enable_gradient_checkpointing(stride=stride)
            else:
                self.transformer.enable_gradient_checkpointing()
        self.memory_policy_log = None     # what "auto" decided, and from which numbers (set at the first step)
        cfg["lean_activations"] = self._lean_setting(cfg.get("lean_activations", "auto"))
        if cfg["lean_activations"] is True:
            self.transformer.enable_lean_activations(True)       # this package's transformer only (AttributeError on anything else: say so loudly)
        self.ref_transformer = None
        if separate_ref:  # the reference's layout: a second frozen copy (:110-111)
            import copy
            ref = copy.deepcopy(self.transformer)
            ref.requires_grad_(False).eval()
            self.ref_transformer = ref
        if scheduler is None:
            # the reference reads the checkpoint's own scheduler_config (:113); the 5B defaults only stand in when
            # there is no checkpoint directory (synthetic weights)
            sched_dir = os.path.join(cfg["model_path"], "scheduler") if local_model else None
            if sched_dir and os.path.isfile(os.path.join(sched_dir, "scheduler_config.json")):
                scheduler = CogVideoXDPMScheduler.from_pretrained(cfg["model_path"], subfolder="scheduler")
            else:
                scheduler = CogVideoXDPMScheduler()
        self.scheduler = scheduler
        self.loss_fn = create_loss_strategy(strategy="dpo", beta=cfg["beta"])
        self.start_time = None
        self.global_step = 0
        self._rng = None
        self.after_reference = None     # hook between the frozen-reference pass and the policy pass (set by DPOEngine)
        self.tuned_gemms = ops.use_tuned_gemms(bool(cfg.get("tuned_gemms", True)))      # vendor-GEMM solutions of tools/gemm_tune.py, when the file matches this stack

    @staticmethod
    def _lean_setting(v):
        """config value -> True / False / "auto": bool-likes (1, 0, "true", "False", ...) mean what they say, anything else but "auto" is an error (a truthy
        non-bool used to run FULL activations silently)"""
        if isinstance(v, str):
            low = v.strip().lower()
            if low == "auto":
                return "auto"
            if low in ("true", "1", "yes", "on"):
                return True
            if low in ("false", "0", "no", "off"):
                return False
            raise ValueError(f'lean_activations: True, False or "auto", got {v!r}')
        if v is None:
            return "auto"
        if isinstance(v, (bool, int)) and v in (0, 1, True, False):
            return bool(v)
        raise ValueError(f'lean_activations: True, False or "auto", got {v!r}')

    def rng(self, device):
        """Per-rank generator of the (t, eps) stream: seed + rank, so data-parallel ranks draw different noise
        (Lightning seeds every rank's global generator differently through the DistributedSampler/rank offset)."""
        if self._rng is None or self._rng.device != device:
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            self._rng = torch.Generator(device=device).manual_seed(int(self.config.get("seed", 0)) + rank)
        return self._rng

    # ------------------------------------------------------------------ forward pieces
    def _ref_forward(self, hs, prompt, tt):
        with torch.no_grad():
            if self.ref_transformer is not None:
                with self.ref_transformer.disable_adapter():
                    return self.ref_transformer(hs, encoder_hidden_states=prompt, timestep=tt, return_dict=True).sample
            with self.transformer.disable_adapter():
                return self.transformer(hs, encoder_hidden_states=prompt, timestep=tt, return_dict=True).sample

    # bytes the backward keeps per token and block (bf16 activations of DESIGN section 3: residual stream x2, LN output + LoRA tail, fused q/k/v, normalised
    # q and k, attention output + its res8 bytes, FF pre-activation, statistics), full set / lean set, at D = 3072 with the int8 output residual; measured 2.9 /
    # 2.2 GB per block and pair at S = 17 776.  The bf16 residual form of "Precise delta" keeps 2 bytes per output element instead of 1: + D bytes per token.
    SAVED_BYTES_PER_TOKEN_BLOCK = {False: 79.0e3, True: 62.5e3}
    WORKSPACE_BYTES = 12e9          # transient buffers of one block's backward + allocator slack that are not in the per-block table

    def memory_policy(self, n_seq, tokens, device):
        """lean_activations = "auto": full activations when the estimated saved activations fit into 80 % of what this process can still get on the device --
        free memory as the driver reports it now (torch.cuda.mem_get_info: other processes, a co-resident scorer / VAE, `separate_ref`'s second copy are all
        already subtracted there) plus what this process's caching allocator holds but has not handed out -- the caching allocator needs slack (a 269 GB step
        fragmented into an out-of-memory error in round 4); else lean; if lean does not fit either, say so instead of failing somewhere inside the backward.
        Decided once per trainer, at the first step (the weights, their caches and the optimizer state are resident by then), logged to stderr."""
        base = self.transformer.get_base_model() if hasattr(self.transformer, "get_base_model") else self.transformer
        if not hasattr(base, "enable_lean_activations") or getattr(base, "gradient_checkpointing", False):
            return None
        free, total = torch.cuda.mem_get_info(device)
        avail = free + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        layers, D = len(base.transformer_blocks), base.config.num_attention_heads * base.config.attention_head_dim
        precise = getattr(base.transformer_blocks[0].attn1.core, "precise_delta", "int8") if layers else "int8"
        extra = {"bf16": 1.0 * D, None: -1.0 * D}.get(precise, 0.0)          # residual bytes per token against the int8 form the table was measured with
        # caches built lazily by the first forward (fused QKV, K-extended and transposed operands: DESIGN section 3) are not resident yet at the first step
        lazy = 0.0 if getattr(base.transformer_blocks[0].attn1.core, "_fused", None) is not None else 14e9 * layers / 42 * (D / 3072) ** 2
        est = {lean: n_seq * tokens * layers * (b * D / 3072 + extra) + self.WORKSPACE_BYTES + lazy for lean, b in self.SAVED_BYTES_PER_TOKEN_BLOCK.items()}
        lean = est[False] > 0.80 * avail
        base.enable_lean_activations(lean)
        self.memory_policy_log = {"lean_activations": lean, "sequences": n_seq, "tokens": tokens, "layers": layers, "precise_delta": precise,
                                  "estimate_full_gb": est[False] / 1e9, "estimate_lean_gb": est[True] / 1e9, "available_gb": avail / 1e9, "device_gb": total / 1e9}
        import sys
        print(f"videogpa_amd: lean_activations=auto -> {lean} ({n_seq} sequences x {tokens} tokens x {layers} blocks: full {est[False] / 1e9:.0f} GB, "
              f"lean {est[True] / 1e9:.0f} GB, available {avail / 1e9:.0f} of {total / 1e9:.0f} GB)" +
              ("" if est[True] <= 0.92 * avail else "; even lean is unlikely to fit: enable_gradient_checkpointing with a stride"), file=sys.stderr, flush=True)
        return lean

    def shared_step_paired(self, x_pair, prompt_emb, timesteps=None, noise=None, cond_pair=None) -> LossOutput:
        """x_pair [B,2,F,C,H,W] bf16 (win, lose); prompt_emb [B,L,4096]; optional fixed (timesteps, noise) for parity
        tests; cond_pair [B,2,F,Cc,H,W]: extra conditioning channels concatenated after noising (I2V, :135-136)."""
        B = x_pair.shape[0]
        dev = x_pair.device
        if self.config.get("lean_activations") == "auto" and self.memory_policy_log is None and x_pair.is_cuda and torch.is_grad_enabled():
            c = self._base_config()
            pt = c.patch_size_t or 1
            self.memory_policy(2 * B, prompt_emb.shape[1] + (x_pair.shape[2] // pt) * (x_pair.shape[4] // c.patch_size) * (x_pair.shape[5] // c.patch_size), dev)
        if timesteps is None:
            timesteps = torch.randint(0, self.scheduler.config.num_train_timesteps, (B,), device=dev, generator=self.rng(dev))
        if noise is None:
            noise = torch.randn(x_pair[:, 0].shape, dtype=x_pair.dtype, device=dev, generator=self.rng(dev))
        xt_pair, vt_pair = self.scheduler.noise_velocity_paired(x_pair.contiguous(), noise.contiguous(), timesteps)
        hs = xt_pair if cond_pair is None else torch.cat([xt_pair, cond_pair], dim=3)
        hs = hs.reshape(2 * B, *hs.shape[2:])
        prompt2 = prompt_emb.repeat_interleave(2, dim=0)
        t2 = timesteps.repeat_interleave(2)
        v_ref = self._ref_forward(hs, prompt2, t2)
        if self.after_reference is not None:       # DPOEngine: the previous optimizer step lands here, its all-reduce hidden under the pass above
            self.after_reference()
        v_pol = self.transformer(hs, encoder_hidden_states=prompt2, timestep=t2, return_dict=True).sample
        v_pol = v_pol.reshape(B, 2, *v_pol.shape[1:])
        v_ref = v_ref.reshape(B, 2, *v_ref.shape[1:])
        lf = self.loss_fn
        # bf16 predictions: the reference (bf16-mixed autocast) forms pred - target in bf16 before the fp32 square
        # (train/loss.py:73-77) -- same rule as the drop-in DPOLoss module (videogpa_amd/loss.py)
        loss, margin, wr, lr, acc, _ = ops.dpo_loss_paired(v_pol.contiguous(), v_ref.contiguous(), vt_pair, beta=lf.beta,
                                                            label_smoothing=lf.label_smoothing, loss_type=lf.loss_type,
                                                            round_diff=(v_pol.dtype == torch.bfloat16))
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

    # ------------------------------------------------------------------ model-variant rules of the reference's three step functions
    def _base_config(self):
        m = self.transformer
        for _ in range(4):                       # PeftModel -> LoraModel -> transformer at most
            cfg = getattr(m, "config", None)
            if cfg is not None and hasattr(cfg, "patch_size"):
                return cfg
            m = m.get_base_model() if hasattr(m, "get_base_model") else getattr(m, "base_model", getattr(m, "model", None))
            if m is None:
                break
        raise RuntimeError("the transformer exposes no CogVideoX config (patch_size / in_channels): cannot choose the step variant")

    def _paired_latents(self, batch):
        """-> x_pair [B,2,F,C,H,W] bf16.  T2V / I2V permute [B,C,F,H,W] -> [B,F,C,H,W] unconditionally
        (train/CogVideoX-5B/03_train.py:120-121, CogVideoX-I2V-5B/03_train.py:116-117); CogVideoX1.5 (patch_size_t set) casts
        to bf16, permutes only when dim 1 is the 16 latent channels and trims F, H, W to even sizes
        (train/CogVideoX1.5-5B/03_train.py:122-142)."""
        cfg = self._base_config()
        v15 = cfg.patch_size_t is not None
        if "x_pair" in batch:
            x_pair = batch["x_pair"]
        else:
            xw, xl = batch["x_win"], batch["x_lose"]
            if not v15 or xw.shape[1] == 16:
                xw, xl = xw.permute(0, 2, 1, 3, 4), xl.permute(0, 2, 1, 3, 4)
            x_pair = torch.stack([xw, xl], dim=1)
        if v15:
            x_pair = x_pair.to(torch.bfloat16)
            Fr, H, W = x_pair.shape[2], x_pair.shape[4], x_pair.shape[5]
            nF, nH, nW = Fr - Fr % 2, H - H % 2, W - W % 2
            if (nF, nH, nW) != (Fr, H, W):
                x_pair = x_pair[:, :, :nF, :, :nH, :nW]
        return x_pair.contiguous()

    def _i2v_condition(self, batch, x_pair):
        """Conditioning channels of the I2V model (train/CogVideoX-I2V-5B/03_train.py:119-130): batch['image_emb'] [B,3,h,w]
        (what DPODataset emits from 02_encode's 'image_embeds') is resized to the pixel size of the latents (nearest, the
        F.interpolate default, :123), encoded by the caller-supplied `image_encoder` (:124-125), zero-padded to F frames
        (:127-128); without an image the condition is zeros (:130).  A pre-encoded batch['image_latent'] is also accepted."""
        cfg = self._base_config()
        B, _, Fr, C, H, W = x_pair.shape
        if cfg.in_channels == C:
            if batch.get("image_emb") is not None or batch.get("image_latent") is not None:
                raise RuntimeError(f"batch carries an I2V image condition but the transformer takes {cfg.in_channels} input channels "
                                   f"(the I2V model takes {2 * C})")
            return None
        if cfg.in_channels != 2 * C:
            raise RuntimeError(f"transformer.in_channels = {cfg.in_channels}; expected {C} (T2V) or {2 * C} (I2V)")
        il = batch.get("image_latent")
        if il is None and batch.get("image_emb") is not None:
            if self.image_encoder is None:
                raise RuntimeError("batch['image_emb'] needs CogVideoXDPOTrainer(image_encoder=...): the callable standing for "
                                   "vae.encode(x).latent_dist.sample() * scaling_factor (train/CogVideoX-I2V-5B/03_train.py:124-125)")
            with torch.no_grad():
                img = torch.nn.functional.interpolate(batch["image_emb"].to(x_pair.device), size=(H * 8, W * 8))
                il = self.image_encoder(img.unsqueeze(2))                       # [B,C,1,h,w]
        if il is None:
            return x_pair.new_zeros(B, 2, Fr, C, H, W)
        if il.shape[1] != 1:
            il = il.permute(0, 2, 1, 3, 4)                                       # -> [B,1,C,h,w] (:126)
        if il.shape[2:] != (C, H, W):
            raise RuntimeError(f"image latent {tuple(il.shape)} does not match the video latents [B,1,{C},{H},{W}]")
        return self.i2v_condition_pair(il.to(x_pair.dtype), Fr)

    def _shared_step(self, batch, timesteps=None, noise=None) -> LossOutput:
        """Reference-shaped entry for all three CogVideoX step functions: batch['x_win'/'x_lose'] [B,C,F,H,W] (or the paired
        'x_pair' [B,2,F,C,H,W]), batch['prompt_emb'], optional batch['image_emb'] / ['image_latent'] (I2V)."""
        x_pair = self._paired_latents(batch)
        cond_pair = self._i2v_condition(batch, x_pair)
        return self.shared_step_paired(x_pair, batch["prompt_emb"].to(x_pair.dtype), timesteps, noise, cond_pair=cond_pair)

    def training_step(self, batch, batch_idx=0):
        if self.start_time is None:
            self.start_time = time.time()
        out = self._shared_step(batch)
        logs = {"train/loss": out.loss.detach(), "train/reward_margin": out.reward_margin,
                "train/reward_accuracy": (out.reward_margin > 0).float().mean()}   # as the reference logs it (:170)
        return out.loss, logs

    def validation_step(self, batch, batch_idx=0):
        """A validation pass never moves the adapters: the `after_reference` hook (DPOEngine's deferred optimizer step) is held off for its duration.  With
        a step still pending the policy pass below would otherwise see adapters updated in the MIDDLE of the validation batch; the caller is told instead
        (DPOEngine.flush() first -- fit.py does)."""
        hook, self.after_reference = self.after_reference, None
        try:
            pending = getattr(getattr(hook, "__self__", None), "_pending", None)
            if pending is not None:
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


class DPOEngine:
    """Micro-step / optimizer-step driver: accumulation, flat-gradient all-reduce, fused clip + AdamW.

    With `overlap` (default: whenever there is more than one rank) the optimizer step of an accumulation window is NOT applied where
    the window ends: the SUM all-reduce of [gradients | logged scalars] is issued on the side stream right after the last backward, and
    the update is applied inside the NEXT micro-step, between its frozen-reference pass (which does not read the adapters) and its
    policy pass (`trainer.after_reference`).  The collective is thereby hidden under a quarter of a step of compute; every policy pass
    still sees exactly the adapters it would see without the reordering, so results are bit-identical (tests/test_dp_gloo.py).
    `flush()` applies a step that is still pending (end of training, before a checkpoint or a validation pass, before reading
    `last`)."""

    _NO_WORK = object()

    def __init__(self, trainer: CogVideoXDPOTrainer, process_group=None, overlap=None):
        self.trainer = trainer
        self.opt = trainer.configure_optimizers(process_group)
        self.accum = int(trainer.config.get("accumulate_grad_batches", 1))
        self.micro = 0
        # every rank must start from rank 0's adapter values (DDP does this broadcast at construction)
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        if multi:
            dist.broadcast(self.opt.flat.flat, src=0, group=process_group)
        ops.bump_adapter_epoch()          # parameters were re-homed into the flat buffer (and possibly overwritten by rank 0's)
        self.opt.zero_grad()
        self.overlap = (multi or os.environ.get("VGPA_FORCE_DIST") == "1") if overlap is None else bool(overlap)
        self._pending = None              # all-reduce handle (or _NO_WORK) of an optimizer step that has not been applied yet
        self.last = {}                    # lr / rank-mean scalars of the most recently applied optimizer step
        self._fresh = False
        self._hooked = self.overlap and hasattr(trainer, "after_reference")
        self._hook_seen = False
        if self._hooked:
            trainer.after_reference = self._from_hook

    def _from_hook(self):
        self._hook_seen = True
        self._finish()

    def _finish(self):
        if self._pending is None:
            return
        work, self._pending = self._pending, None
        tail = self.opt.flat.tail
        lr = self.opt.step(None if work is DPOEngine._NO_WORK else work)
        # mean over ranks and micro-steps; a device tensor (read it only when logging)
        self.last = {"lr": lr, "sync": (tail[:3] / tail[3:4]).clone(), "sync_keys": ("train/loss", "train/reward_margin", "train/reward_accuracy")}
        self._fresh = True
        self.opt.zero_grad()
        self.trainer.global_step += 1

    def flush(self) -> Dict[str, Any]:
        """Apply the optimizer step that is still in flight (if any); returns the scalars of the last applied step."""
        self._finish()
        self._fresh = False
        return self.last

    def micro_step(self, batch) -> Dict[str, Any]:
        if not self._hooked:
            self._finish()                                               # a trainer without the hook: the step lands before its forward
        self._hook_seen = False
        loss, logs = self.trainer.training_step(batch, self.micro)      # hooked: a pending step is applied after the reference pass
        if self._hooked and not self._hook_seen:
            # The trainer carries `after_reference` but its training_step never called it.  The policy forward above has then run on the
            # adapters as they are; applying a pending update NOW (between that forward and its backward) would pair new A / B with old
            # activations -- silently wrong gradients.  So: never step here.  With nothing pending this is harmless and the engine simply
            # stops relying on the hook; with a step pending the forward already used stale adapters, which cannot be repaired.
            if self._pending is not None:
                raise RuntimeError("DPOEngine: an optimizer step is pending but trainer.training_step() did not call "
                                   "trainer.after_reference() between its reference and policy passes; construct the engine with "
                                   "overlap=False for this trainer (or call the hook)")
            self._hooked = False
            self.trainer.after_reference = None
        (loss / self.accum).backward()
        # the three scalars the reference logs with sync_dist=True (train/CogVideoX-5B/03_train.py:164-173) go into the tail
        # of the gradient buffer and are reduced by the SAME all-reduce: no extra collective, no host sync
        tail = self.opt.flat.tail
        tail[:3].add_(torch.stack([logs["train/loss"].float(), logs["train/reward_margin"].float(),
                                   logs["train/reward_accuracy"].float()]))
        tail[3:4].add_(1.0)
        self.micro += 1
        if self.micro % self.accum == 0:
            work = self.opt.all_reduce_grads()
            self._pending = DPOEngine._NO_WORK if work is None else work
            if not self.overlap:
                self._finish()
        if self._fresh:                   # an optimizer step was applied during this call: hand its scalars to the caller
            logs.update(self.last)
            self._fresh = False
        return logs
