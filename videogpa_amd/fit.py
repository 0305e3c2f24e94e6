"""`main_train` of the reference (train/CogVideoX-5B/03_train.py:219-288) without Lightning / wandb: dataset ->
98/2 split (seed 42) -> per-rank shard -> prefetching loader -> DPOEngine steps -> final_lora adapter.  One process per
GPU (torchrun); rank 0 logs and saves."""
import os
import time
from typing import Any, Dict, Optional

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Subset

from .dataset import DPODataset, collate_paired, shard_indices
from .loader import PairedPrefetcher
from .trainer import DEFAULT_CONFIG, CogVideoXDPOTrainer, DPOEngine


def fit(config: Dict[str, Any], transformer=None, dataset: Optional[DPODataset] = None, log=print) -> CogVideoXDPOTrainer:
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(config)
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if dataset is None:
        dataset = DPODataset(base_path=cfg["base_path"], metadata_path=cfg["metadata_path"], metric_name=cfg["metric_name"],
                             metric_mode="min", min_gap=cfg.get("min_gap", 0.05), motion_threshold=cfg.get("motion_threshold", 1e-3))
    n = len(dataset)
    g = torch.Generator().manual_seed(42)
    perm = torch.randperm(n, generator=g).tolist()
    n_train = int(0.98 * n) if n > 1 else n
    train_idx = perm[:n_train] or perm
    trainer = CogVideoXDPOTrainer(cfg, transformer=transformer).cuda()
    trainer.train()
    engine = DPOEngine(trainer)
    step_target = cfg["max_steps"]
    epoch = 0
    t0 = time.time()
    while trainer.global_step < step_target:
        local = [train_idx[i] for i in shard_indices(len(train_idx), rank, world, epoch=epoch)]
        loader = DataLoader(Subset(dataset, local), batch_size=cfg["batch_size"], shuffle=False, num_workers=cfg.get("num_workers", 4),
                            collate_fn=collate_paired, drop_last=False)
        for batch in PairedPrefetcher(loader):
            logs = engine.micro_step(batch)
            if rank == 0 and "lr" in logs and trainer.global_step % cfg.get("log_every_n_steps", 10) == 0:
                sps = trainer.global_step * world * cfg["batch_size"] * cfg["accumulate_grad_batches"] / max(1e-9, time.time() - t0)
                log(f"step {trainer.global_step}: loss {float(logs['train/loss']):.6f} margin {float(logs['train/reward_margin']):.3e} "
                    f"lr {logs['lr']:.3e} samples/s {sps:.3f} max_mem {torch.cuda.max_memory_reserved() / 2 ** 30:.1f} GB")
            if trainer.global_step >= step_target:
                break
        epoch += 1
    if rank == 0 and cfg.get("output_dir"):
        trainer.transformer.save_pretrained(os.path.join(cfg["output_dir"], "final_lora"))
    return trainer
