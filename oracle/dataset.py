"""Preference-pair mining, CPU oracle.  PINNED against train/dataset.py (tests/golden/dataset_pairs.json).

Follows train/dataset.py:102-201: per group keep videos having the metric, motion_norm, latent_path and
condition_path whose files exist and whose motion_norm >= threshold (:130-150); sort by metric (ascending
for 'min', descending for 'max') (:158-163); winner = first, loser = last (:165-170); optional winner
threshold (:177-183); keep when |metric gap| >= min_gap (:186-188)."""
from pathlib import Path


def mine_pairs(groups, base_path, metric_name="consistency_score", metric_mode="min", min_gap=0.1,
               metric_threshold=None, motion_threshold=0.001):
    base = Path(base_path)
    pairs = []
    for g in groups:
        vids = g.get("videos", [])
        if len(vids) < 2:
            continue
        ok = []
        for v in vids:
            if metric_name not in v or "motion_norm" not in v:
                continue
            if "latent_path" not in v or "condition_path" not in v:
                continue
            if not (base / v["latent_path"]).exists() or not (base / v["condition_path"]).exists():
                continue
            if v["motion_norm"] < motion_threshold:
                continue
            ok.append(v)
        if len(ok) < 2:
            continue
        s = sorted(ok, key=lambda x: x[metric_name], reverse=(metric_mode == "max"))
        win, lose = s[0], s[-1]
        if metric_threshold is not None:
            if metric_mode == "min" and win[metric_name] >= metric_threshold:
                continue
            if metric_mode != "min" and win[metric_name] <= metric_threshold:
                continue
        gap = abs(win[metric_name] - lose[metric_name])
        if gap < min_gap:
            continue
        pairs.append({
            "group_id": g.get("group_id", "unknown"),
            "prompt": g.get("text_prompt", g.get("prompt", "")),
            "winner": win, "loser": lose, "metric_gap": gap,
        })
    return pairs
