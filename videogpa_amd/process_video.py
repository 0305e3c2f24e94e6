"""`VideoProcessor` of the reference (pipelines/process_video.py:17-201) with the geometry on the MI355X: the same
`process(video_path, thresholds, num_frames)` -> `{threshold: {metric name: value, "motion_norm": ...}, "_extrinsic": ...}`
contract and the same `compute_metrics` dispatch by metric name ("Consistency_Score" -> (score, motion_norm) tuple,
"MVCS" -> needs depths / intrinsics / extrinsics, anything else -> compute(gt=, rep=)).

What is NOT here, because it is a third-party network or host I/O outside the hot path: the VGGT /
DA3 backbones and video decoding.  They are passed in:
  * `frame_sampler(video_path, n_frames) -> uint8 [T,H,W,3]`      (utils/video_utils.py:19-44, decord + cv2 upstream)
  * backbone "vggt":  `backbone_fn(frames) -> dict` with the keys utils/model_utils.py:89-122 returns
      images [T,3,H,W] in [0,1], world_points_from_depth [T,H,W,3], depth_conf [T,H,W], depth, and either
      extrinsic [T,3,4] + intrinsic [T,3,3] or the raw `pose_enc` [T,9] (decoded here, vggt/utils/pose_enc.py:62-124)
  * backbone "da3":   `backbone_fn(frames) -> object` with .processed_images [T,H,W,3], .extrinsics, .intrinsics, .depth,
      .conf (depth_anything_3/api.py:133-273); world points are unprojected here (process_video.py:132-166).

Everything downstream of the backbone -- confidence cut, z-buffer reprojection of all views, MSE / motion / MVCS --
runs as HIP kernels (videogpa_amd/csrc/scorer.hip, scorer2.hip) with no host round trip until the final floats.
"""
import os

import numpy as np
import torch

from . import scorer


class VideoProcessor:
    def __init__(self, metrics, model_name=None, device=None, backbone=None, backbone_fn=None, frame_sampler=None, vggt_model=None):
        """First four arguments as the reference's (pipelines/process_video.py:17-29).  Where the reference loads VGGT-1B / DA3 weights
        itself, this class takes `backbone_fn` (and `frame_sampler` for paths): third-party networks and video decoding stay the caller's.
        `vggt_model(images [1,T,3,h,518]) -> predictions` is the bare VGGT network: with it the wrapper of utils/model_utils.py:89-122
        (device preprocessing in front, batch squeeze and pose decoding behind) runs here instead of inside `backbone_fn`."""
        if vggt_model is not None and backbone_fn is None:
            backbone_fn = lambda frames: self._run_vggt(vggt_model, frames)
        self.device = device or "cuda"
        self.metrics = metrics
        self.backbone = self._resolve_backbone(backbone, model_name)
        self.model_name = model_name
        self.backbone_fn = backbone_fn
        self.frame_sampler = frame_sampler

    @staticmethod
    def _resolve_backbone(backbone, model_name):
        """pipelines/process_video.py:32-43: explicit argument, then VIDEO_PROCESSOR_BACKBONE, then the model name."""
        if backbone:
            return backbone.lower()
        env = os.getenv("VIDEO_PROCESSOR_BACKBONE")
        if env:
            return env.lower()
        if model_name and "depth-anything" in model_name.lower():
            return "da3"
        return "vggt"

    # ------------------------------------------------------------------ entry points
    def process(self, video_path, thresholds, num_frames, save_visuals=False, out_dir=None):
        if save_visuals:
            raise NotImplementedError("save_visuals (PNG dumps through cv2) is host I/O outside the on-device path")
        if self.backbone_fn is None:
            raise RuntimeError("VideoProcessor needs backbone_fn (the VGGT / DA3 network is third-party: pass a callable frames -> predictions)")
        frames = video_path if not isinstance(video_path, (str, os.PathLike)) else self._sample(video_path, num_frames)
        if self.backbone == "da3":
            return self._process_da3(frames, thresholds)
        return self._process_vggt(frames, thresholds)

    def _sample(self, video_path, num_frames):
        if self.frame_sampler is None:
            raise RuntimeError("VideoProcessor needs frame_sampler(video_path, n_frames) -> uint8 [T,H,W,3] (video decoding is host I/O; "
                               "or pass the frame array itself instead of a path)")
        return self.frame_sampler(video_path, num_frames)

    def _run_vggt(self, model, frames):
        """utils/model_utils.py:86-122 around the external network, step for step: the model is moved to the device and put in eval mode (:92),
        frames are preprocessed on the device (:94), the forward runs under no_grad + bf16 autocast (:99-102), the camera head's `pose_enc` is decoded
        into extrinsic / intrinsic BEFORE the batch axis is dropped (:104-106, on the device: scorer.pose_encoding_to_extri_intri), then every
        [1, ...] tensor is squeezed (:108-111)"""
        from .model_utils import prepare_inputs
        dev = torch.device(self.device)
        if hasattr(model, "to") and hasattr(model, "eval"):
            model = model.to(dev).eval()
        images = prepare_inputs(np.asarray(frames) if not torch.is_tensor(frames) else frames, device=self.device)
        with torch.no_grad(), torch.autocast(dev.type, dtype=torch.bfloat16, enabled=dev.type == "cuda"):
            preds = dict(model(images))
        preds["images"] = images
        if "pose_enc" in preds and "extrinsic" not in preds:
            preds["extrinsic"], preds["intrinsic"] = scorer.pose_encoding_to_extri_intri(preds["pose_enc"].float(), tuple(images.shape[-2:]))
        preds = {k: (v.squeeze(0) if torch.is_tensor(v) and v.ndim > 0 and v.shape[0] == 1 else v) for k, v in preds.items()}
        if "world_points" in preds:       # :116-117
            preds["world_points_from_depth"] = preds["world_points"]
        return preds

    def _process_vggt(self, frames, thresholds):
        preds = dict(self.backbone_fn(frames))
        images = preds["images"]
        _, _, height, width = images.shape
        if "extrinsic" not in preds:      # raw camera head output: decode on device (utils/model_utils.py:108)
            ext, intr = scorer.pose_encoding_to_extri_intri(preds["pose_enc"], (height, width))
            preds["extrinsic"], preds["intrinsic"] = ext.reshape(-1, 3, 4), intr.reshape(-1, 3, 3)
        return self._score(preds, frames, thresholds, height, width)

    def _process_da3(self, frames, thresholds):
        preds, gt_frames = self._build_da3_predictions(frames)
        height, width = gt_frames.shape[-2:]
        return self._score(preds, gt_frames, thresholds, height, width)

    def _build_da3_predictions(self, frames):
        """pipelines/process_video.py:132-166."""
        pred = self.backbone_fn([frames[i] for i in range(len(frames))])
        images = scorer._dev_f32(pred.processed_images)
        if images.max() > 1.0:
            images = images / 255.0
        images = images.permute(0, 3, 1, 2).contiguous()
        extrinsics, intrinsics, depths = (scorer._dev_f32(x) for x in (pred.extrinsics, pred.intrinsics, pred.depth))
        conf = scorer._dev_f32(pred.conf) if getattr(pred, "conf", None) is not None else torch.ones_like(depths)
        world = scorer.unproject_depth_to_world(depths, intrinsics, extrinsics)
        return {"world_points_from_depth": world, "depth_conf": conf, "images": images, "extrinsic": extrinsics,
                "intrinsic": intrinsics, "depth": depths}, images

    def _score(self, preds, gt_frames, thresholds, height, width):
        extrinsics, intrinsics, depths = preds["extrinsic"], preds["intrinsic"], preds.get("depth")
        results = {}
        for th in thresholds:
            # get_colored_pointcloud(mode="depth", conf_thres=th) + batch_reproject (process_video.py:84-87,116-119), fused:
            # the confidence cut is a predicate inside the splat kernel, the filtered cloud is never materialised
            rep = scorer.reproject_predictions(preds["world_points_from_depth"], preds.get("depth_conf"), preds["images"], intrinsics,
                                               extrinsics, height, width, conf_thres=th)
            results[th] = self.compute_metrics(gt_frames, rep, extrinsics, intrinsics=intrinsics, depths=depths)
        results["_extrinsic"] = self._to_serializable(extrinsics)
        return results

    def compute_metrics(self, gt_frames, rep_frames, extrinsics, intrinsics=None, depths=None):
        """pipelines/process_video.py:168-196."""
        results = {}
        for name, metric_fn in self.metrics.items():
            if name == "Consistency_Score":
                final_score, motion = metric_fn.compute(gt=gt_frames, rep=rep_frames, extrinsics=extrinsics)
                results[name] = final_score
                results["motion_norm"] = motion
            elif name == "MVCS":
                results[name] = metric_fn.compute(gt=gt_frames, rep=rep_frames, depths=depths, intrinsics=intrinsics, extrinsics=extrinsics)
            else:
                results[name] = metric_fn.compute(gt=gt_frames, rep=rep_frames)
        return results

    @staticmethod
    def _to_serializable(value):
        if isinstance(value, torch.Tensor):
            return value.detach().cpu().tolist()
        return np.asarray(value).tolist()
