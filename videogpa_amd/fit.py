"""`main_train` of the reference (train/CogVideoX-5B/03_train.py:219-288; train/Wan2.2-TI2V-5B/03_train.py:309-387 is the same loop around
WanDPOTrainer and the unpaired collate_fn) without Lightning / wandb: dataset -> 98/2 split (seed 42) -> per-rank shard -> prefetching
loader -> DPOEngine steps -> final_lora adapter.  One process per GPU (torchrun); rank 0 logs and saves."""
import os
import time
from typing import Any, Dict, Optional

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Subset

from .dataset import DPODataset, collate_fn, collate_paired, shard_indices
from .loader import PairedPrefetcher
from .trainer import DEFAULT_CONFIG, CogVideoXDPOTrainer, DPOEngine


def save_checkpoint(engine: DPOEngine, path: str, val: Optional[Dict[str, float]] = None, position: Optional[Dict[str, int]] = None) -> None:
    """Adapter (PEFT format) + optimizer state (flat Adam moments, step count) + trainer step: enough to resume.  The
    reference's ModelCheckpoint pickles the whole LightningModule incl. the two frozen 5B copies (SURVEY B-13); the
    frozen base is not state, so it is not written here."""
    engine.flush()          # a checkpoint never holds an optimizer step that is still in flight
    os.makedirs(path, exist_ok=True)
    engine.trainer.transformer.save_pretrained(path)
    sd = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in engine.opt.state_dict().items()}
    sd.update(global_step=engine.trainer.global_step, micro=engine.micro, val=val, position=dict(position or {}))
    # the (t, eps) stream of THIS rank: without it a resumed run would replay the timesteps and noise of steps 0..k.  Other ranks
    # write their own generator state next to it (rank 0 owns optimizer.pt).
    rng = engine.trainer._rng
    if rng is not None:
        torch.save(rng.get_state(), os.path.join(path, f"rng_rank{dist.get_rank() if dist.is_initialized() else 0}.pt"))
    tmp = os.path.join(path, "optimizer.pt.tmp")
    torch.save(sd, tmp)
    os.replace(tmp, os.path.join(path, "optimizer.pt"))


def save_rng_state(engine: DPOEngine, path: str) -> None:
    """What the ranks other than 0 contribute to a checkpoint: their own (t, eps) generator state."""
    rng = engine.trainer._rng
    if rng is not None:
        os.makedirs(path, exist_ok=True)
        torch.save(rng.get_state(), os.path.join(path, f"rng_rank{dist.get_rank() if dist.is_initialized() else 0}.pt"))


def load_checkpoint(engine: DPOEngine, path: str) -> Dict[str, int]:
    """-> the data position {"epoch", "batch"} the run had reached (zeros for checkpoints written before positions were saved)."""
    sd = torch.load(os.path.join(path, "optimizer.pt"), map_location="cpu")
    dev = engine.opt.flat.flat.device
    engine.opt.load_state_dict({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sd.items()
                                if k in ("exp_avg", "exp_avg_sq", "param", "step_count")})
    engine.trainer.global_step = int(sd["global_step"])
    engine.micro = int(sd["micro"])
    rank = dist.get_rank() if dist.is_initialized() else 0
    rp = os.path.join(path, f"rng_rank{rank}.pt")
    if os.path.exists(rp):
        engine.trainer.rng(dev).set_state(torch.load(rp, map_location="cpu"))
    pos = sd.get("position") or {}
    return {"epoch": int(pos.get("epoch", 0)), "batch": int(pos.get("batch", 0))}


def _prune_top_k(kept, k):
    """ModelCheckpoint(save_top_k=k, monitor='val/loss', mode='min'): keep the k best checkpoints (k < 0: all), delete the others' directories.
    The most recent one is never deleted by a tie; other ranks' rng files live in the same directory and go with it."""
    import shutil
    if k < 0 or len(kept) <= k:
        return
    kept.sort(key=lambda e: (e[0], -e[1]))
    for _, _, d in kept[k:]:
        shutil.rmtree(d, ignore_errors=True)
    del kept[k:]


@torch.no_grad()
def validate(trainer, dataset, val_idx, cfg, rank=0, world=1, collate=collate_paired) -> Dict[str, float]:
    """validation_step over the 2 % split, as the reference runs it (train/CogVideoX-5B/03_train.py:190-206,252,261): batch size 1, at most
    `limit_val_batches` = 50 batches per rank, mean over pairs and ranks.  The (t, eps) draws of validation come from their OWN generator, re-seeded
    identically at every call (seed 10 007 + rank): the training noise stream is untouched by when / how often validation runs, and the validation
    loss of successive checkpoints is measured on the same noise, so it is comparable (Lightning's validation consumes the global generator instead)."""
    local = val_idx[rank::world][: int(cfg.get("limit_val_batches", 50))]
    acc = torch.zeros(4, dtype=torch.float64, device="cuda")
    if local:
        loader = DataLoader(Subset(dataset, local), batch_size=1, shuffle=False, num_workers=0, collate_fn=collate)
        was = trainer.training
        trainer.eval()
        keep = getattr(trainer, "_rng", None)
        trainer._rng = torch.Generator(device=next(trainer.parameters()).device).manual_seed(10007 + rank)   # the device trainer.rng() compares with
        try:
            for batch in PairedPrefetcher(loader):
                out = trainer.validation_step(batch)
                b = (batch["x_pair"] if "x_pair" in batch else batch["x_win"]).shape[0]
                acc += torch.stack([out["val/loss"].double() * b, out["val/reward_margin"].double() * b,
                                    out["val/reward_accuracy"].double() * b, torch.tensor(float(b), dtype=torch.float64, device="cuda")])
        finally:
            trainer._rng = keep
        trainer.train(was)
    if dist.is_initialized() and world > 1:
        dist.all_reduce(acc)
    a = acc.tolist()
    n = max(a[3], 1.0)
    return {"val/loss": a[0] / n, "val/reward_margin": a[1] / n, "val/reward_accuracy": a[2] / n}


def fit(config: Dict[str, Any], transformer=None, dataset: Optional[DPODataset] = None, log=print, image_encoder=None, model: str = "cogvideox"):
    """model = "cogvideox" (CogVideoXDPOTrainer over the paired [B,2,F,C,H,W] layout) or "wan" (WanDPOTrainer over the reference's own batch keys
    x_win / x_lose [B,C,F,H,W], prompt_emb, image_latent: train/Wan2.2-TI2V-5B/03_train.py:309-387)."""
    if model not in ("cogvideox", "wan"):
        raise ValueError(f"fit: model must be 'cogvideox' or 'wan', got {model!r}")
    cfg = dict(DEFAULT_CONFIG)
    if model == "wan":
        from .wan import DEFAULT_CONFIG as WAN_DEFAULTS, WanDPOTrainer
        cfg.update(WAN_DEFAULTS)
    cfg.update(config)
    collate = collate_paired if model == "cogvideox" else collate_fn
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if dataset is None:
        dataset = DPODataset(base_path=cfg["base_path"], metadata_path=cfg["metadata_path"], metric_name=cfg["metric_name"],
                             metric_mode="min", min_gap=cfg.get("min_gap", 0.05), motion_threshold=cfg.get("motion_threshold", 1e-3))
    n = len(dataset)
    g = torch.Generator().manual_seed(42)
    perm = torch.randperm(n, generator=g).tolist()
    if n == 0:
        raise ValueError("DPODataset produced no preference pairs (check metric / min_gap / motion_threshold): nothing to train on")
    n_train = int(0.98 * n) if n > 1 else n
    train_idx = perm[:n_train] or perm
    val_idx = perm[n_train:]
    if model == "wan":
        if transformer is None:
            raise ValueError("fit(model='wan') needs the denoiser (videogpa_amd.wan_model.WanModel or a module with its call convention)")
        trainer = WanDPOTrainer(cfg, transformer).cuda()
    else:
        trainer = CogVideoXDPOTrainer(cfg, transformer=transformer, image_encoder=image_encoder).cuda()
    trainer.train()
    engine = DPOEngine(trainer)
    pos = {"epoch": 0, "batch": 0}
    if cfg.get("resume_from"):
        pos = load_checkpoint(engine, cfg["resume_from"])       # adapters, Adam moments, step counters, this rank's (t, eps) stream, data position
    step_target = cfg["max_steps"]
    every = int(cfg.get("checkpoint_every_n_steps", 1000))          # ModelCheckpoint(every_n_train_steps=1000), 03_train.py:268-275
    epoch, skip = pos["epoch"], pos["batch"]
    t0, start_step = time.time(), trainer.global_step
    last_ckpt = trainer.global_step
    kept = []           # (val/loss, step, dir) of the checkpoints on disk: ModelCheckpoint(monitor="val/loss", mode="min", save_top_k), 03_train.py:268-275
    while trainer.global_step < step_target:
        local = [train_idx[i] for i in shard_indices(len(train_idx), rank, world, epoch=epoch)]
        local = local[skip * cfg["batch_size"]:]                    # a resumed run continues inside the epoch it was in
        loader = DataLoader(Subset(dataset, local), batch_size=cfg["batch_size"], shuffle=False, num_workers=cfg.get("num_workers", 4),
                            collate_fn=collate, drop_last=False)
        batch_in_epoch = skip
        skip = 0
        for batch in PairedPrefetcher(loader):
            logs = engine.micro_step(batch)
            batch_in_epoch += 1
            # with the all-reduce overlapped the step lands one micro-step late: land it now when something waits for it
            due = trainer.global_step + (1 if engine._pending is not None else 0)
            if engine._pending is not None and (due >= step_target or (every > 0 and due % every == 0)):
                logs.update(engine.flush())
            if rank == 0 and "lr" in logs and trainer.global_step % cfg.get("log_every_n_steps", 10) == 0:
                sps = (trainer.global_step - start_step) * world * cfg["batch_size"] * cfg["accumulate_grad_batches"] / max(1e-9, time.time() - t0)
                sync = logs["sync"].tolist()          # rank-mean of the step's scalars (rode the gradient all-reduce)
                log(f"step {trainer.global_step}: loss {sync[0]:.6f} margin {sync[1]:.3e} acc {sync[2]:.2f} "
                    f"lr {logs['lr']:.3e} samples/s {sps:.3f} max_mem {torch.cuda.max_memory_reserved() / 2 ** 30:.1f} GB")
            if "lr" in logs and every > 0 and trainer.global_step % every == 0 and trainer.global_step != last_ckpt:
                last_ckpt = trainer.global_step
                val = validate(trainer, dataset, val_idx, cfg, rank, world, collate) if val_idx else None
                ckpt_dir = os.path.join(cfg["output_dir"], "checkpoints", f"step={trainer.global_step}") if cfg.get("output_dir") else None
                if rank == 0:
                    if val is not None:
                        log(f"step {trainer.global_step}: " + " ".join(f"{k} {v:.6f}" for k, v in val.items()))
                    if ckpt_dir:
                        save_checkpoint(engine, ckpt_dir, val, position={"epoch": epoch, "batch": batch_in_epoch})
                        kept.append((val["val/loss"] if val is not None else float("inf"), trainer.global_step, ckpt_dir))
                        _prune_top_k(kept, int(cfg.get("save_top_k", 10)))
                elif ckpt_dir:
                    save_rng_state(engine, ckpt_dir)
            if trainer.global_step >= step_target:
                break
        else:
            epoch += 1
            continue
        break
    engine.flush()
    if rank == 0 and cfg.get("output_dir"):
        trainer.transformer.save_pretrained(os.path.join(cfg["output_dir"], "final_lora"))
    return trainer
