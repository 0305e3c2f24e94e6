"""Host-side preference-pair data path (stays Python, like the reference): drop-in for train/dataset.py's
`DPODataset` / `collate_fn` (same constructor arguments, same returned keys / shapes / dtypes), plus
`collate_paired`, which emits the paired latent layout [B,2,F,C,H,W] the MI355X trainer consumes directly, and
`shard_indices`, the per-rank strided split used for data-parallel training.

Selection rules (train/dataset.py:102-201): a group needs >= 2 usable videos; a video is usable when it carries
the ranking metric, `motion_norm`, `latent_path`, `condition_path`, both files exist, and motion_norm >=
motion_threshold; rank by the metric ('min': ascending, 'max': descending); winner = best, loser = worst; optional
winner threshold; keep the pair when |metric gap| >= min_gap.  The condition is always the WINNER's (:234).
"""
import json
from pathlib import Path
from typing import Any, Dict, List, Optional

import torch
from torch.utils.data import Dataset


def _load(path):
    try:
        return torch.load(path, weights_only=True)
    except TypeError:
        return torch.load(path)


class DPODataset(Dataset):
    def __init__(self, base_path: str, metadata_path: str, metric_name: str = "consistency_score", metric_mode: str = "min",
                 min_gap: float = 0.1, metric_threshold: Optional[float] = None, motion_threshold: float = 0.001,
                 max_samples: Optional[int] = None):
        super().__init__()
        self.base_path = Path(base_path)
        self.metadata_path = Path(metadata_path)
        self.metric_name, self.metric_mode = metric_name, metric_mode
        self.min_gap, self.metric_threshold, self.motion_threshold = min_gap, metric_threshold, motion_threshold
        with open(metadata_path, "r") as f:
            data = json.load(f)
        if "groups" not in data:
            raise ValueError("Invalid metadata format: missing 'groups' key")
        self.raw_groups = data["groups"]
        self.preference_pairs = self._create_preference_pairs()
        if max_samples is not None:
            self.preference_pairs = self.preference_pairs[:max_samples]

    def _usable(self, video) -> bool:
        need = (self.metric_name, "motion_norm", "latent_path", "condition_path")
        if any(k not in video for k in need):
            return False
        if not (self.base_path / video["latent_path"]).exists() or not (self.base_path / video["condition_path"]).exists():
            return False
        return video["motion_norm"] >= self.motion_threshold

    def _create_preference_pairs(self) -> List[Dict[str, Any]]:
        pairs = []
        descending = self.metric_mode == "max"
        for group in self.raw_groups:
            videos = group.get("videos", [])
            if len(videos) < 2:
                continue
            usable = [v for v in videos if self._usable(v)]
            if len(usable) < 2:
                continue
            ranked = sorted(usable, key=lambda v: v[self.metric_name], reverse=descending)
            winner, loser = ranked[0], ranked[-1]
            wm, lm = winner[self.metric_name], loser[self.metric_name]
            if self.metric_threshold is not None:
                fails = wm >= self.metric_threshold if self.metric_mode == "min" else wm <= self.metric_threshold
                if fails:
                    continue
            gap = abs(wm - lm)
            if gap < self.min_gap:
                continue
            pairs.append({
                "group_id": group.get("group_id", "unknown"),
                "prompt": group.get("text_prompt", group.get("prompt", "")),
                "input_image_path": group.get("image_path", group.get("input_image_path")),
                "original_video_path": group.get("original_video_path"),
                "winner": winner, "loser": loser, "metric_gap": gap,
            })
        return pairs

    def __len__(self) -> int:
        return len(self.preference_pairs)

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        pair = self.preference_pairs[idx]
        winner, loser = pair["winner"], pair["loser"]
        cond = _load(self.base_path / winner["condition_path"])
        item = {
            "x_win": _load(self.base_path / winner["latent_path"]),      # [C, F, H, W] as 02_encode.py stores it
            "x_lose": _load(self.base_path / loser["latent_path"]),
            "prompt_emb": cond.get("encoder_hidden_states"),
            "prompt": pair["prompt"],
            "m_win": winner[self.metric_name],
            "m_lose": loser[self.metric_name],
        }
        for src, dst in (("image_embeds", "image_emb"), ("image_latent", "image_latent")):
            if cond.get(src) is not None:
                item[dst] = cond[src]
        return item


def collate_fn(batch: List[Dict[str, Any]]) -> Dict[str, Any]:
    out: Dict[str, Any] = {}
    for key in ("x_win", "x_lose", "prompt_emb"):
        if key in batch[0]:
            out[key] = torch.stack([b[key] for b in batch])
    for key in ("image_emb", "image_latent"):
        if key in batch[0] and batch[0][key] is not None:
            out[key] = torch.stack([b[key] for b in batch])
    if "prompt" in batch[0]:
        out["prompt"] = [b["prompt"] for b in batch]
    for key in ("m_win", "m_lose"):
        if key in batch[0]:
            out[key] = torch.tensor([b[key] for b in batch])
    return out


def collate_paired(batch: List[Dict[str, Any]]) -> Dict[str, Any]:
    """Like collate_fn but win/lose latents arrive as ONE tensor x_pair [B,2,F,C,H,W] (frame-major, the layout
    the transformer and the fused noise/loss kernels read), built on the host in the DataLoader worker."""
    out = collate_fn(batch)
    xw = out.pop("x_win").permute(0, 2, 1, 3, 4)
    xl = out.pop("x_lose").permute(0, 2, 1, 3, 4)
    out["x_pair"] = torch.stack([xw, xl], dim=1).contiguous()
    return out


def shard_indices(n: int, rank: int, world_size: int, epoch: int = 0, seed: int = 42, shuffle: bool = True, drop_last: bool = False):
    """DistributedSampler-style split: seeded permutation, padded to a multiple of world_size, strided by rank."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if drop_last:
        idx = idx[: n - n % world_size]
    elif len(idx) % world_size:
        pad = world_size - len(idx) % world_size
        idx += (idx * ((pad + len(idx) - 1) // max(1, len(idx)) + 1))[:pad]
    return idx[rank::world_size]
