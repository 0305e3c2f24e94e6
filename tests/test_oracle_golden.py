"""Oracle vs the reference's own outputs (fixtures made by tests/golden/make_golden.py from
train/loss.py, train/dataset.py, utils/projection_utils.py, metrics/consistency_score.py, metrics/mse.py)."""
import json
import os

import numpy as np
import torch

from oracle import dataset as ods
from oracle import dpo, scorer


def test_dpo_loss_matches_reference(golden_dir):
    cases = torch.load(os.path.join(golden_dir, "dpo_loss.pt"), weights_only=False)
    assert len(cases) >= 19
    for c in cases:
        ins = [x.clone().requires_grad_(i < 2) for i, x in enumerate(c["inputs"])]
        out = dpo.dpo_loss(*ins, beta=c["beta"], label_smoothing=c["label_smoothing"], loss_type=c["loss_type"])
        for k in ("loss", "reward_margin", "winner_reward", "loser_reward", "accuracy"):
            assert torch.equal(out[k].detach(), c[k]), (k, c["beta"], c["loss_type"])
        if c["grad_v_win"] is not None:
            out["loss"].backward()
            assert torch.equal(ins[0].grad, c["grad_v_win"])
            assert torch.equal(ins[1].grad, c["grad_v_lose"])


def test_dpo_ln2_identity():
    g = torch.Generator().manual_seed(0)
    v = [torch.randn(2, 3, 4, 8, 8, generator=g) for _ in range(4)]
    out = dpo.dpo_loss(v[0], v[1], v[0].clone(), v[1].clone(), v[2], v[3], beta=1.0)
    assert abs(float(out["loss"]) - np.log(2.0)) < 1e-7


def test_pair_mining_matches_reference(golden_dir, tmp_path):
    with open(os.path.join(golden_dir, "dataset_pairs.json")) as f:
        gold = json.load(f)
    for name in gold["existing_files"]:
        (tmp_path / name).write_bytes(b"")
    for r in gold["results"]:
        kw = dict(r["kwargs"])
        max_samples = kw.pop("max_samples", None)
        pairs = ods.mine_pairs(gold["meta"]["groups"], tmp_path, metric_name="consistency_score", **kw)
        if max_samples is not None:
            pairs = pairs[:max_samples]
        got = [{"group_id": p["group_id"], "winner": p["winner"]["video_path"], "loser": p["loser"]["video_path"],
                "gap": p["metric_gap"]} for p in pairs]
        assert got == r["pairs"]


def test_project_points_matches_reference(golden_dir):
    gold = torch.load(os.path.join(golden_dir, "scorer.pt"), weights_only=False)
    for c in gold["project"]:
        canvas = scorer.project_points(c["pc"].numpy(), c["colors"].numpy(), c["K"].numpy(), c["E"].numpy(), c["H"], c["W"])
        ref = c["canvas"].numpy()
        assert canvas.shape == ref.shape and canvas.dtype == ref.dtype
        assert np.array_equal(canvas, ref), f"{(canvas != ref).any(-1).sum()} pixels differ"


def test_motion_and_mse_match_reference(golden_dir):
    gold = torch.load(os.path.join(golden_dir, "scorer.pt"), weights_only=False)
    for c in gold["motion"]:
        assert scorer.motion_score(c["E"].numpy()) == c["score"]
    for c in gold["mse"]:
        assert scorer.mse(c["gt"], c["rep"]) == c["val"]
