"""Generate golden fixtures by IMPORTING the reference (runs only in the build container, where
/root/reference exists).  Output: small data files under tests/golden/.  Nothing here is shipped as
source of the reference -- only inputs and the reference's outputs on them.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def golden_dpo_loss():
    sys.path.insert(0, os.path.join(REF, "train"))
    import loss as ref_loss
    cases = []
    g = torch.Generator().manual_seed(20260929)
    for (B, shape) in [(1, (3, 4, 8, 8)), (2, (5, 4, 6, 10)), (3, (13, 4, 4, 6))]:
        for beta in (1.0, 500.0):
            for ls, lt in ((0.0, "sigmoid"), (0.1, "sigmoid"), (0.0, "hinge")):
                t = [torch.randn(B, *shape, generator=g) for _ in range(6)]
                # make policy close to ref so logits are in an interesting range for beta=500
                t[2] = t[0] + 0.05 * torch.randn(B, *shape, generator=g)
                t[3] = t[1] + 0.05 * torch.randn(B, *shape, generator=g)
                ins = [x.clone().requires_grad_(i < 2) for i, x in enumerate(t)]
                fn = ref_loss.DPOLoss(beta=beta, label_smoothing=ls, loss_type=lt)
                out = fn(*ins)
                out.loss.backward()
                cases.append({
                    "beta": beta, "label_smoothing": ls, "loss_type": lt,
                    "inputs": [x.detach() for x in t],
                    "loss": out.loss.detach(), "reward_margin": out.reward_margin.detach(),
                    "winner_reward": out.winner_reward.detach(), "loser_reward": out.loser_reward.detach(),
                    "accuracy": out.accuracy.detach(),
                    "grad_v_win": ins[0].grad.clone(), "grad_v_lose": ins[1].grad.clone(),
                })
    # known answer: policy == ref -> ln 2
    v = [torch.randn(2, 3, 4, 8, 8, generator=g) for _ in range(4)]
    out = ref_loss.create_loss_strategy("dpo", beta=1.0)(v[0], v[1], v[0].clone(), v[1].clone(), v[2], v[3])
    cases.append({"beta": 1.0, "label_smoothing": 0.0, "loss_type": "sigmoid",
                  "inputs": [v[0], v[1], v[0].clone(), v[1].clone(), v[2], v[3]],
                  "loss": out.loss, "reward_margin": out.reward_margin, "winner_reward": out.winner_reward,
                  "loser_reward": out.loser_reward, "accuracy": out.accuracy, "grad_v_win": None, "grad_v_lose": None})
    torch.save(cases, os.path.join(HERE, "dpo_loss.pt"))
    print("dpo_loss.pt:", len(cases), "cases")


def golden_dataset():
    sys.path.insert(0, os.path.join(REF, "train"))
    import dataset as ref_ds
    rng = np.random.default_rng(7)
    with tempfile.TemporaryDirectory() as d:
        groups = []
        for gi in range(12):
            vids = []
            for vi in range(int(rng.integers(1, 5))):
                v = {"video_path": f"g{gi}_v{vi}.mp4", "generation_id": str(vi)}
                if rng.random() > 0.1:
                    v["consistency_score"] = round(float(rng.random() * 2), 4)
                if rng.random() > 0.1:
                    v["motion_norm"] = float(rng.choice([0.0, 0.0005, 0.5, 1.2]))
                if rng.random() > 0.1:
                    v["latent_path"] = f"lat_{gi}_{vi}.pt"
                    if rng.random() > 0.15:
                        torch.save(torch.full((16, 2, 4, 4), float(gi * 10 + vi)).to(torch.bfloat16),
                                   os.path.join(d, v["latent_path"]))
                v["condition_path"] = f"cond_{gi}_{vi}.pt"
                torch.save({"encoder_hidden_states": torch.full((6, 8), float(gi * 10 + vi)).to(torch.bfloat16)},
                           os.path.join(d, v["condition_path"]))
                vids.append(v)
            groups.append({"group_id": f"g{gi}", "prompt": f"prompt {gi}", "videos": vids})
        meta = {"groups": groups}
        with open(os.path.join(d, "meta_data.json"), "w") as f:
            json.dump(meta, f)
        # files that exist (so the test can recreate the directory without the reference)
        existing = sorted(os.listdir(d))
        results = []
        for kw in [dict(metric_mode="min", min_gap=0.05, motion_threshold=1e-3),
                   dict(metric_mode="max", min_gap=0.3, motion_threshold=1e-3),
                   dict(metric_mode="min", min_gap=0.0, motion_threshold=0.0, metric_threshold=0.8),
                   dict(metric_mode="min", min_gap=0.05, motion_threshold=1e-3, max_samples=2)]:
            ds = ref_ds.DPODataset(d, os.path.join(d, "meta_data.json"), metric_name="consistency_score", **kw)
            pairs = [{"group_id": p["group_id"], "winner": p["winner"]["video_path"], "loser": p["loser"]["video_path"],
                      "gap": p["metric_gap"]} for p in ds.preference_pairs]
            batch = None
            if len(ds) >= 2:
                b = ref_ds.collate_fn([ds[0], ds[1]])
                batch = {"x_win_shape": list(b["x_win"].shape), "x_win_dtype": str(b["x_win"].dtype),
                         "x_win_vals": [float(b["x_win"][i].flatten()[0]) for i in range(2)],
                         "x_lose_vals": [float(b["x_lose"][i].flatten()[0]) for i in range(2)],
                         "prompt_emb_shape": list(b["prompt_emb"].shape),
                         "prompt_emb_vals": [float(b["prompt_emb"][i].flatten()[0]) for i in range(2)],
                         "prompt": b["prompt"], "m_win": b["m_win"].tolist(), "m_lose": b["m_lose"].tolist(),
                         "keys": sorted(b.keys())}
            results.append({"kwargs": kw, "pairs": pairs, "batch": batch})
    with open(os.path.join(HERE, "dataset_pairs.json"), "w") as f:
        json.dump({"meta": meta, "existing_files": existing, "results": results}, f, indent=1)
    print("dataset_pairs.json:", [len(r["pairs"]) for r in results])


def golden_scorer():
    cv2 = _stub("cv2")
    piq = _stub("piq"); piq.ssim = None
    _stub("lpips")
    sys.path.insert(0, REF)
    import importlib.util

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    proj = load("ref_projection_utils", os.path.join(REF, "utils/projection_utils.py"))
    _stub("metrics")
    base = load("metrics.base", os.path.join(REF, "metrics/base.py"))
    mse = load("metrics.mse", os.path.join(REF, "metrics/mse.py"))
    lp = _stub("metrics.lpips"); lp.LPIPSMetric = object
    cs = load("metrics.consistency_score", os.path.join(REF, "metrics/consistency_score.py"))

    # The reference scatters duplicate pixel indices with `canvas[v, u] = c` after a z-descending sort
    # (utils/projection_utils.py:36-50).  torch's duplicate-index scatter is only ordered ("last write wins"
    # == nearest point wins, the evident intent of the sort) when it runs single-threaded; with >1 CPU thread
    # (and on CUDA) the winner of a collision is racy.  Goldens are therefore made with one thread.
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(99)
    out = {"project": [], "motion": [], "mse": []}
    H, W = 37, 53
    for case in range(4):
        N = [5000, 20000, 300, 0][case]
        pc = torch.randn(N, 3, generator=g) * torch.tensor([1.0, 0.8, 0.5]) + torch.tensor([0.0, 0.0, 3.0])
        if case == 1:
            colors = torch.rand(N, 3, generator=g) * 255.0
        elif case == 2:
            colors = torch.rand(N, 3, generator=g)          # <=1.0 branch (utils/projection_utils.py:45-46)
        else:
            colors = torch.rand(N, 3, generator=g) * 300.0 - 20.0  # exercises clamp
        fx = 40.0 + 5 * case
        K = torch.tensor([[fx, 0.0, W / 2], [0.0, fx * 1.1, H / 2], [0.0, 0.0, 1.0]])
        ang = 0.1 * (case + 1)
        R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
        E = torch.eye(4)
        E[:3, :3] = R
        E[:3, 3] = torch.tensor([0.1 * case, -0.05, 0.2])
        if case == 3:
            pc = torch.randn(50, 3, generator=g) - torch.tensor([0.0, 0.0, 10.0])  # everything behind camera
            colors = torch.rand(50, 3, generator=g)
        canvas = proj.project_points(pc, colors, K, E[:3] if case % 2 else E, H, W)
        out["project"].append({"pc": pc, "colors": colors, "K": K, "E": E[:3] if case % 2 else E,
                               "H": H, "W": W, "canvas": canvas})
    for case in range(3):
        T = 6
        Es = torch.eye(4).repeat(T, 1, 1)
        for i in range(T):
            a = 0.05 * i * (case + 1)
            Es[i, :3, :3] = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32)
            Es[i, :3, 3] = torch.randn(3, generator=g) * 0.1 * case
        out["motion"].append({"E": Es, "score": float(cs.compute_motion_score_vectorized(Es, device="cpu"))})
    m = mse.MSEMetric()
    gt_u8 = (torch.rand(3, 11, 13, 3, generator=g) * 255).to(torch.uint8)
    rep_pm1 = torch.rand(3, 3, 11, 13, generator=g) * 2 - 1
    rep_01 = torch.rand(3, 3, 11, 13, generator=g)
    out["mse"].append({"gt": gt_u8, "rep": rep_pm1, "val": m.compute(gt=gt_u8, rep=rep_pm1)})
    out["mse"].append({"gt": rep_01, "rep": rep_pm1, "val": m.compute(gt=rep_01, rep=rep_pm1)})
    out["mse"].append({"gt": gt_u8.numpy(), "rep": rep_01, "val": m.compute(gt=gt_u8.numpy(), rep=rep_01)})
    torch.save(out, os.path.join(HERE, "scorer.pt"))
    print("scorer.pt written")


def golden_adapter_config_keys():
    """Key set PEFT wrote for the released adapters (checkpoints/VideoGPA-T2V-lora/adapter_config.json)."""
    cfg = json.load(open(os.path.join(REF, "checkpoints/VideoGPA-T2V-lora/adapter_config.json")))
    json.dump(sorted(cfg.keys()), open(os.path.join(HERE, "adapter_config_keys.json"), "w"))


if __name__ == "__main__":
    golden_adapter_config_keys()
    golden_dpo_loss()
    golden_dataset()
    golden_scorer()
