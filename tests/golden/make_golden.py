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


def golden_scorer2():
    """Second scorer fixture (tests/golden/scorer2.pt): the confidence filter (utils/pointcloud_utils.py:10-80), PSNR and
    the size-mismatch branch of MSE (metrics/mse.py:24-25,56-80), MVCS (metrics/mvcs.py:12-114), VGGT's pose-encoding
    decoder (vggt/utils/pose_enc.py:62-124) and the DA3 unprojection (depth_anything_3/utils/geometry.py:54-59,434-497;
    pipelines/process_video.py:151-156) -- all run by importing the reference modules on CPU."""
    import importlib.util
    _stub("cv2")
    piq = _stub("piq"); piq.ssim = None
    _stub("lpips")
    ply = _stub("plyfile"); ply.PlyData = object; ply.PlyElement = object
    if REF not in sys.path:
        sys.path.insert(0, REF)

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    if "metrics" not in sys.modules:
        _stub("metrics")
    load("metrics.base", os.path.join(REF, "metrics/base.py"))
    mse = load("metrics.mse", os.path.join(REF, "metrics/mse.py"))
    mvcs = load("metrics.mvcs", os.path.join(REF, "metrics/mvcs.py"))
    pcu = load("ref_pointcloud_utils", os.path.join(REF, "utils/pointcloud_utils.py"))
    from vggt.utils.pose_enc import pose_encoding_to_extri_intri
    from depth_anything_3.utils.geometry import affine_inverse, unproject_depth

    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(4242)
    out = {"pointcloud": [], "psnr": [], "mse_resize": [], "mvcs": [], "pose_enc": [], "da3_unproject": []}

    # ---- confidence filter
    T, h, w = 3, 9, 11
    for mode, thres in (("depth", 0), ("depth", 30), ("depth", 90), ("pointmap", 50), ("depth", 100)):
        pts = torch.randn(T, h, w, 3, generator=g)
        conf = torch.rand(T, h, w, generator=g) * 3
        conf.view(-1)[::17] = float("nan")
        conf.view(-1)[5::23] = float("inf")
        conf.view(-1)[7::13] = 1e-6
        conf.view(-1)[3::29] = conf.view(-1)[2]          # ties
        imgs = torch.rand(T, 3, h, w, generator=g)
        preds = {"images": imgs}
        if mode == "pointmap":
            preds.update(world_points=pts, world_points_conf=conf)
        else:
            preds.update(world_points_from_depth=pts, depth_conf=conf)
        v, c = pcu.get_colored_pointcloud(preds, mode=mode, conf_thres=thres)
        out["pointcloud"].append({"mode": mode, "conf_thres": thres, "points": pts, "conf": conf, "images": imgs,
                                  "vertices": v.clone(), "colors": c.clone()})

    # ---- PSNR / resized MSE
    psnr = mse.PSNRMetric(device="cpu")
    m = mse.MSEMetric()
    a = torch.rand(2, 3, 12, 16, generator=g)
    b = (a + 0.05 * torch.randn(2, 3, 12, 16, generator=g)).clamp(0, 1)
    small = torch.rand(2, 3, 7, 9, generator=g) * 2 - 1
    big = torch.rand(2, 3, 20, 31, generator=g)
    u8 = (torch.rand(2, 12, 16, 3, generator=g) * 255).to(torch.uint8)
    for gt, rep in ((a, b), (a, a.clone()), (a, small), (u8, big), (u8.numpy(), small)):
        out["psnr"].append({"gt": gt, "rep": rep, "val": psnr.compute(gt=gt, rep=rep)})
    for gt, rep in ((a, small), (u8, big), (small, a)):
        out["mse_resize"].append({"gt": gt, "rep": rep, "val": m.compute(gt=gt, rep=rep)})

    # ---- MVCS: a smooth scene seen by slowly moving cameras (+ depth noise), three input layouts
    def scene(T, H, W, noise, seed):
        gg = torch.Generator().manual_seed(seed)
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        K = torch.tensor([[W * 0.9, 0.0, W / 2], [0.0, W * 0.95, H / 2], [0.0, 0.0, 1.0]]).repeat(T, 1, 1)
        K[:, 0, 0] += torch.arange(T) * 0.5
        E = torch.eye(4).repeat(T, 1, 1)
        depths = torch.empty(T, H, W)
        for i in range(T):
            ang = 0.02 * i
            E[i, :3, :3] = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
            E[i, :3, 3] = torch.tensor([0.03 * i, -0.01 * i, 0.02 * i])
            depths[i] = 2.0 + 0.3 * torch.sin(xs / W * 3 + 0.1 * i) + 0.2 * torch.cos(ys / H * 2) + noise * torch.randn(H, W, generator=gg)
        return depths, K, E

    mv = mvcs.MVCSMetric(device="cpu")
    for ci, (T_, H_, W_, noise) in enumerate(((4, 24, 32, 0.01), (3, 17, 23, 0.05), (5, 20, 20, 0.0), (1, 8, 8, 0.0))):
        d, K, E = scene(T_, H_, W_, noise, 100 + ci)
        if ci == 0:
            dd, KK, EE = d.unsqueeze(-1), K, E[:, :3]
        elif ci == 1:
            dd, KK, EE = d.unsqueeze(1), torch.eye(4).repeat(T_, 1, 1), E
            KK[:, :3, :3] = K
        else:
            dd, KK, EE = d.numpy(), K, E
        if ci == 2:
            dd = dd.copy()
            dd[1, :5] = -1.0                       # negative depths: projected z <= 0 -> masked out
        val = mv.compute(gt=None, rep=None, depths=dd, intrinsics=KK, extrinsics=EE)
        out["mvcs"].append({"depths": dd if torch.is_tensor(dd) else torch.from_numpy(dd), "depths_is_numpy": not torch.is_tensor(dd),
                            "intrinsics": KK, "extrinsics": EE, "val": float(val)})

    # ---- pose-encoding decoder
    pe = torch.randn(2, 5, 9, generator=g)
    pe[..., 7:] = 0.6 + 0.3 * torch.rand(2, 5, 2, generator=g)
    ex, intr = pose_encoding_to_extri_intri(pe, image_size_hw=(294, 518))
    out["pose_enc"].append({"pose_enc": pe, "image_size_hw": (294, 518), "extrinsics": ex, "intrinsics": intr})

    # ---- DA3: affine_inverse + unproject_depth as pipelines/process_video.py:151-156 calls them
    d, K, E = scene(3, 10, 14, 0.02, 7)
    c2w = affine_inverse(E)
    wp = unproject_depth(d.unsqueeze(0).unsqueeze(-1), K.unsqueeze(0), c2w.unsqueeze(0)).squeeze(0)
    out["da3_unproject"].append({"depths": d, "intrinsics": K, "extrinsics": E, "c2w": c2w, "world_points": wp})

    # ---- LPIPSMetric / Consistency_Score wrappers around a STAND-IN perceptual net (the LPIPS-VGG weights are third-party): pins the
    #      input normalisation, layout change, resize and the MSE + ratio * LPIPS combination (metrics/lpips.py:21-63,
    #      metrics/consistency_score.py:52-72)
    lp_mod = load("metrics.lpips", os.path.join(REF, "metrics/lpips.py"))
    cs_mod = load("metrics.consistency_score", os.path.join(REF, "metrics/consistency_score.py"))
    net = lambda a, b: (a - b).abs().mean(dim=(1, 2, 3), keepdim=True) + 0.01 * a.mean(dim=(1, 2, 3), keepdim=True)
    lpm = lp_mod.LPIPSMetric(device="cpu", lpips_net=net)
    csm = cs_mod.Consistency_Score(lpips_net=net, device="cpu")
    out["lpips"], out["consistency"] = [], []
    T_, H_, W_ = 3, 14, 18
    gt_u8 = (torch.rand(T_, H_, W_, 3, generator=g) * 255).to(torch.uint8)
    rep_pm1 = torch.rand(T_, 3, H_, W_, generator=g) * 2 - 1
    gt_01 = torch.rand(T_, 3, H_, W_, generator=g)
    rep_small = torch.rand(T_, 3, 9, 11, generator=g) * 2 - 1
    gt_255 = torch.rand(T_, 3, H_, W_, generator=g) * 255
    Es = torch.eye(4).repeat(T_, 1, 1)
    for i in range(T_):
        a_ = 0.07 * i
        Es[i, :3, :3] = torch.tensor([[np.cos(a_), -np.sin(a_), 0], [np.sin(a_), np.cos(a_), 0], [0, 0, 1]], dtype=torch.float32)
        Es[i, :3, 3] = torch.tensor([0.1 * i, 0.02 * i, -0.05 * i])
    for gt_, rep_ in ((gt_u8.numpy(), rep_pm1), (gt_01, rep_pm1), (gt_u8, rep_small), (gt_255, gt_01)):
        out["lpips"].append({"gt": gt_, "rep": rep_, "val": lpm.compute(gt=gt_, rep=rep_)})
        for ratio in (1, 0.1):
            sc_, mo_ = csm.compute(gt=gt_, rep=rep_, extrinsics=Es, ratio=ratio)
            out["consistency"].append({"gt": gt_, "rep": rep_, "extrinsics": Es, "ratio": ratio, "score": sc_, "motion": mo_})

    torch.save(out, os.path.join(HERE, "scorer2.pt"))
    print("scorer2.pt written:", {k: len(v) for k, v in out.items()})


def golden_vggt_attention():
    """tests/golden/vggt_attention.pt: the reference-HELD QK-norm attention (vggt/layers/attention.py:20-72: fused qkv, LayerNorm(64) on
    q and k, optional RotaryPositionEmbedding2D vggt/layers/rope.py:60-188, SDPA, proj) and the aggregator's block around it
    (vggt/layers/block.py:30-108: LayerNorm, attention, LayerScale, Mlp) run by IMPORTING the reference, fp32 on CPU, head_dim 64:
    forward outputs and the gradients of the input and of every parameter.  All values are bf16-representable on the input side,
    so a bf16 implementation differs by its arithmetic only.  This pins the attention kernel family against code the reference
    itself ships (the CogVideoX attention lives in un-vendored diffusers)."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from vggt.layers.attention import Attention
    from vggt.layers.block import Block
    from vggt.layers.rope import RotaryPositionEmbedding2D
    torch.manual_seed(20260930)
    g = torch.Generator().manual_seed(77)
    rb = lambda t: t.to(torch.bfloat16).float()
    out = {"attention": [], "block": []}
    for N, use_rope, dim, heads, Bx in ((37, False, 128, 2, 2), (261, False, 128, 2, 1), (261, True, 128, 2, 1), (1374, True, 64, 1, 1)):
        att = Attention(dim, num_heads=heads, qk_norm=True, rope=RotaryPositionEmbedding2D(frequency=100.0) if use_rope else None)
        with torch.no_grad():
            for p in att.parameters():
                p.copy_(rb(torch.randn(p.shape, generator=g) * (0.08 if p.ndim == 2 else 0.3) + (1.0 if p.ndim == 1 and p.shape[0] == 64 else 0.0)))
        x = rb(torch.randn(Bx, N, dim, generator=g)).requires_grad_(True)
        pos = None
        if use_rope:   # what the aggregator feeds (vggt/models/aggregator.py:215-224): 0 for the special tokens, grid position + 1 for patches
            side = int((N - 5) ** 0.5)
            yy, xx = torch.meshgrid(torch.arange(side), torch.arange(side), indexing="ij")
            grid = torch.stack([yy.flatten(), xx.flatten()], -1) + 1
            pos = torch.cat([torch.zeros(N - side * side, 2, dtype=torch.long), grid], 0)[None].expand(Bx, -1, -1).contiguous()
        y = att(x, pos=pos)
        gy = rb(torch.randn(y.shape, generator=g))
        y.backward(gy)
        out["attention"].append({"N": N, "dim": dim, "heads": heads, "rope": use_rope, "pos": None if pos is None else pos[0].to(torch.int16),
                                 "x": x.detach().to(torch.bfloat16), "grad_out": gy.to(torch.bfloat16),      # bf16-representable: stored as bf16
                                 "params": {k: v.detach().to(torch.bfloat16) for k, v in att.named_parameters()}, "y": y.detach(),
                                 "grad_x": x.grad.clone(), "grad_params": {k: v.grad.clone() for k, v in att.named_parameters()}})
    # one frame + one global block of the alternating-attention aggregator on [B=1, S=3 frames, P tokens]: frame attention sees
    # (B*S, P, C), global attention (B, S*P, C) (vggt/models/aggregator.py:260-306)
    dim, heads = 128, 2
    Bq, Sq, side = 1, 3, 6
    P = 5 + side * side
    rope = RotaryPositionEmbedding2D(frequency=100.0)
    blocks = [Block(dim, heads, mlp_ratio=2.0, qkv_bias=True, proj_bias=True, ffn_bias=True, init_values=0.01, qk_norm=True, rope=rope) for _ in range(2)]
    with torch.no_grad():
        for b in blocks:
            for n, p in b.named_parameters():
                if "gamma" in n:
                    p.copy_(rb(0.5 + 0.1 * torch.randn(p.shape, generator=g)))
                else:
                    p.copy_(rb(torch.randn(p.shape, generator=g) * (0.08 if p.ndim == 2 else 0.3) + (1.0 if "norm" in n and n.endswith("weight") else 0.0)))
    tokens = rb(torch.randn(Bq * Sq, P, dim, generator=g)).requires_grad_(True)
    yy, xx = torch.meshgrid(torch.arange(side), torch.arange(side), indexing="ij")
    pos = torch.cat([torch.zeros(5, 2, dtype=torch.long), torch.stack([yy.flatten(), xx.flatten()], -1) + 1], 0)[None].expand(Bq * Sq, -1, -1).contiguous()
    t1 = blocks[0](tokens, pos=pos)                                                   # frame attention
    t2 = blocks[1](t1.view(Bq, Sq * P, dim), pos=pos.view(Bq, Sq * P, 2))             # global attention
    gy = rb(torch.randn(t2.shape, generator=g))
    t2.backward(gy)
    out["block"] = {"B": Bq, "S": Sq, "P": P, "dim": dim, "heads": heads, "pos": pos[0].to(torch.int16), "tokens": tokens.detach().to(torch.bfloat16),
                    "grad_out": gy.to(torch.bfloat16), "params": [{k: v.detach().to(torch.bfloat16) for k, v in b.named_parameters()} for b in blocks],
                    "frame_out": t1.detach(), "global_out": t2.detach(), "grad_tokens": tokens.grad.clone(),
                    # gradients of the MLP matrices are plain GEMMs outside this path: their norms only
                    "grad_params": [{k: (v.grad.clone() if "mlp.fc" not in k or v.ndim == 1 else v.grad.norm()) for k, v in b.named_parameters()} for b in blocks]}
    torch.save(out, os.path.join(HERE, "vggt_attention.pt"))
    print("vggt_attention.pt:", len(out["attention"]), "attention cases + 1 frame/global block pair")


def golden_da3_attention():
    """tests/golden/da3_attention.pt: a SECOND reference-held witness of the attention kernel family -- Depth Anything 3's DINOv2 layers
    (depth_anything_3/model/dinov2/layers/attention.py:18-82: fused qkv, LayerNorm(head_dim) on q and k, RotaryPositionEmbedding2D rope.py, SDPA, proj;
    block.py:26-110: LayerNorm eps 1e-6, LayerScale, Mlp) IMPORTED from the reference and run in fp32 on the CPU the way the backbone calls them
    (vision_transformer.py:282-364): "local" attention on (B*S, N, C) with grid positions + 1 (0 for the special tokens), "global" attention on
    (B, S*N, C) with ALL-ZERO positions (pos_nodiff: the rotation is the identity).  The package __init__ chain above the layers needs addict /
    omegaconf (absent here), so the three parent packages are registered as plain namespaces and only the layer modules themselves execute."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for pkg in ("depth_anything_3", "depth_anything_3.model", "depth_anything_3.model.dinov2"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF, *pkg.split("."))]
            sys.modules[pkg] = m
    from depth_anything_3.model.dinov2.layers.attention import Attention
    from depth_anything_3.model.dinov2.layers.block import Block
    from depth_anything_3.model.dinov2.layers.rope import PositionGetter, RotaryPositionEmbedding2D
    torch.manual_seed(20261001)
    g = torch.Generator().manual_seed(991)
    rb = lambda t: t.to(torch.bfloat16).float()
    getter = PositionGetter()
    out = {"attention": []}
    for side_h, side_w, n_special, nodiff, dim, heads, Bx in ((14, 18, 1, False, 128, 2, 2), (9, 9, 5, True, 128, 2, 1), (37, 37, 1, False, 64, 1, 1)):
        att = Attention(dim, num_heads=heads, qkv_bias=True, qk_norm=True, rope=RotaryPositionEmbedding2D(frequency=100.0))
        with torch.no_grad():
            for p in att.parameters():
                p.copy_(rb(torch.randn(p.shape, generator=g) * (0.08 if p.ndim == 2 else 0.3) + (1.0 if p.ndim == 1 and p.shape[0] == 64 else 0.0)))
        N = n_special + side_h * side_w
        x = rb(torch.randn(Bx, N, dim, generator=g)).requires_grad_(True)
        pos = getter(Bx, side_h, side_w, device=torch.device("cpu")) + 1                       # vision_transformer.py:286-295
        pos = torch.cat([torch.zeros(Bx, n_special, 2, dtype=pos.dtype), pos], dim=1)
        if nodiff:
            pos = torch.zeros_like(pos)
        y = att(x, pos=pos)
        gy = rb(torch.randn(y.shape, generator=g))
        y.backward(gy)
        out["attention"].append({"N": N, "dim": dim, "heads": heads, "nodiff": nodiff, "pos": pos[0].to(torch.int16), "x": x.detach().to(torch.bfloat16),
                                 "grad_out": gy.to(torch.bfloat16), "params": {k: v.detach().to(torch.bfloat16) for k, v in att.named_parameters()},
                                 "y": y.detach(), "grad_x": x.grad.clone(), "grad_params": {k: v.grad.clone() for k, v in att.named_parameters()}})
    # one local + one global block on [B = 1, S = 3 views, N tokens] (process_attention, vision_transformer.py:351-364)
    dim, heads, Bq, Sq, sh, sw, nsp = 128, 2, 1, 3, 5, 7, 1
    N = nsp + sh * sw
    rope = RotaryPositionEmbedding2D(frequency=100.0)
    blocks = [Block(dim, heads, mlp_ratio=2.0, qkv_bias=True, proj_bias=True, ffn_bias=True, init_values=0.01, qk_norm=True, rope=rope, ln_eps=1e-6) for _ in range(2)]
    with torch.no_grad():
        for b in blocks:
            for n, p in b.named_parameters():
                if "gamma" in n:
                    p.copy_(rb(0.5 + 0.1 * torch.randn(p.shape, generator=g)))
                else:
                    p.copy_(rb(torch.randn(p.shape, generator=g) * (0.08 if p.ndim == 2 else 0.3) + (1.0 if "norm" in n and n.endswith("weight") else 0.0)))
    tokens = rb(torch.randn(Bq * Sq, N, dim, generator=g)).requires_grad_(True)
    pos = getter(Bq * Sq, sh, sw, device=torch.device("cpu")) + 1
    pos = torch.cat([torch.zeros(Bq * Sq, nsp, 2, dtype=pos.dtype), pos], dim=1)
    t1 = blocks[0](tokens, pos=pos)                                                              # local: (b s) n c, grid positions
    t2 = blocks[1](t1.view(Bq, Sq * N, dim), pos=torch.zeros(Bq, Sq * N, 2, dtype=pos.dtype))    # global: b (s n) c, pos_nodiff
    gy = rb(torch.randn(t2.shape, generator=g))
    t2.backward(gy)
    out["block"] = {"B": Bq, "S": Sq, "N": N, "dim": dim, "heads": heads, "ln_eps": 1e-6, "pos": pos[0].to(torch.int16), "tokens": tokens.detach().to(torch.bfloat16),
                    "grad_out": gy.to(torch.bfloat16), "params": [{k: v.detach().to(torch.bfloat16) for k, v in b.named_parameters()} for b in blocks],
                    "local_out": t1.detach(), "global_out": t2.detach(), "grad_tokens": tokens.grad.clone(),
                    "grad_params": [{k: (v.grad.clone() if "mlp.fc" not in k or v.ndim == 1 else v.grad.norm()) for k, v in b.named_parameters()} for b in blocks]}
    torch.save(out, os.path.join(HERE, "da3_attention.pt"))
    print("da3_attention.pt:", len(out["attention"]), "attention cases + 1 local/global block pair")


def golden_vggt_aggregator():
    """tests/golden/vggt_aggregator.pt: vggt/models/aggregator.py::Aggregator IMPORTED from the reference and run in fp32 on the CPU at small dimensions
    (patch_embed="conv", head_dim 64): the special-token assembly (slice_expand_and_flatten), PositionGetter positions, the aa_block_num x aa_order x
    aa_block_size loop and the list of concatenated intermediates, for aa_block_size 1 and 2.  Inputs and parameters are bf16-representable."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from vggt.models.aggregator import Aggregator
    g = torch.Generator().manual_seed(4242)
    rb = lambda t: t.to(torch.bfloat16).float()
    cases = []
    for depth, bs, B, S, H, W in ((2, 1, 2, 3, 28, 42), (2, 2, 1, 2, 42, 28)):
        torch.manual_seed(7 + bs)
        agg = Aggregator(img_size=H, patch_size=14, embed_dim=128, depth=depth, num_heads=2, mlp_ratio=2.0, num_register_tokens=4, patch_embed="conv",
                         aa_block_size=bs, qk_norm=True, rope_freq=100, init_values=0.01).eval()
        with torch.no_grad():
            for n, p in agg.named_parameters():
                if "gamma" in n:
                    p.copy_(rb(0.5 + 0.1 * torch.randn(p.shape, generator=g)))
                elif n in ("camera_token", "register_token"):
                    p.copy_(rb(0.5 * torch.randn(p.shape, generator=g)))
                elif n.startswith("patch_embed."):
                    p.copy_(rb(torch.randn(p.shape, generator=g) * 0.05))
                else:
                    p.copy_(rb(torch.randn(p.shape, generator=g) * (0.08 if p.ndim == 2 else 0.3) + (1.0 if "norm" in n and n.endswith("weight") else 0.0)))
        images = rb(torch.rand(B, S, 3, H, W, generator=g))
        with torch.no_grad():
            outs, start = agg(images)
        cases.append({"depth": depth, "aa_block_size": bs, "B": B, "S": S, "H": H, "W": W, "embed_dim": 128, "num_heads": 2, "mlp_ratio": 2.0,
                      "images": images.to(torch.bfloat16), "params": {k: v.detach().to(torch.bfloat16) for k, v in agg.state_dict().items()},
                      "outputs": [o.clone() for o in outs], "patch_start_idx": start})
    torch.save(cases, os.path.join(HERE, "vggt_aggregator.pt"))
    print("vggt_aggregator.pt:", [(c["aa_block_size"], len(c["outputs"]), tuple(c["outputs"][0].shape)) for c in cases])


def golden_preprocess():
    """tests/golden/preprocess.npz: the VGGT input preprocessing, utils/model_utils.py:16-85.  That module imports torchvision (not
    installed here), so it cannot be imported as a whole; its one non-trivial step is PIL's bicubic `Image.resize` (:51), which IS
    installed (Pillow 12.2.0) and is called here for real, exactly as :35,:51 call it.  Around it the fixture records the function's
    own arithmetic: sizes from :36-48, ToTensor's /255 (:52) left to the consumer (the arrays are stored as uint8 = the tensor
    times 255, exact for the white pad value 1.0 too), centre crop :54-56, white pad :58-71."""
    from PIL import Image
    rng = np.random.default_rng(518)
    out = {}
    cases = [("land_crop", 1, 90, 256, "crop"), ("port_crop", 1, 400, 300, "crop"), ("land_pad", 2, 90, 256, "pad"), ("port_pad", 1, 256, 90, "pad"),
             ("same_crop", 1, 98, 518, "crop"), ("up_crop", 2, 37, 53, "crop"), ("half_even_dn", 1, 175, 518, "crop"), ("half_even_up", 1, 189, 518, "pad")]
    for name, T, H, W, mode in cases:
        yy, xx = np.mgrid[0:H, 0:W]
        frames = []
        for t in range(T):   # smooth structure + hard edges + noise, so that ringing, clipping at 0 / 255 and rounding are all exercised
            base = 127 + 120 * np.sin(xx / (3.0 + t) + yy / 7.0)[..., None] * np.array([1.0, -1.0, 0.5])
            base = base + 90 * ((xx // 16 + yy // 16) % 2)[..., None] - 45
            base = base + rng.normal(0, 25, (H, W, 3))
            frames.append(np.clip(base, 0, 255).astype(np.uint8))
        frames = np.stack(frames)
        res = []
        for f in frames:
            img = Image.fromarray(f, "RGB")
            width, height = img.size
            if mode == "pad":
                if width >= height:
                    nw = 518; nh = round(height * (nw / width) / 14) * 14
                else:
                    nh = 518; nw = round(width * (nh / height) / 14) * 14
            else:
                nw = 518; nh = round(height * (nw / width) / 14) * 14
            r = np.asarray(img.resize((nw, nh), Image.Resampling.BICUBIC))
            if mode == "crop" and nh > 518:
                s0 = (nh - 518) // 2
                r = r[s0:s0 + 518]
            if mode == "pad":
                hp, wp = 518 - r.shape[0], 518 - r.shape[1]
                if hp > 0 or wp > 0:
                    t_ = torch.nn.functional.pad(torch.from_numpy(r.copy()).permute(2, 0, 1), (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2), mode="constant", value=255)
                    r = t_.permute(1, 2, 0).numpy()
            res.append(r.transpose(2, 0, 1))
        out[name + "__frames"] = frames
        out[name + "__expect_u8"] = np.stack(res)
        out[name + "__mode"] = np.array(mode)
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **out)
    print("preprocess.npz", {k: v.shape for k, v in out.items() if k.endswith("expect_u8")})


def golden_adapter_config_keys():
    """Key set PEFT wrote for the released adapters (checkpoints/VideoGPA-T2V-lora/adapter_config.json)."""
    cfg = json.load(open(os.path.join(REF, "checkpoints/VideoGPA-T2V-lora/adapter_config.json")))
    json.dump(sorted(cfg.keys()), open(os.path.join(HERE, "adapter_config_keys.json"), "w"))


if __name__ == "__main__":
    golden_adapter_config_keys()
    golden_dpo_loss()
    golden_dataset()
    golden_scorer()
    golden_scorer2()
    golden_vggt_attention()
    golden_da3_attention()
    golden_vggt_aggregator()
    golden_preprocess()
