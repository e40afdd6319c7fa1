"""Evaluation caller of the hot path: iterate a scene dataset, run `PoseEstimator.run_inference_pipeline`, collect the predictions
of every stage and gather them across ranks.

Mirrors `PredictionRunner` (reference src/megapose/evaluation/prediction_runner.py:47-209): same constructor arguments, the same
`run_inference_pipeline(pose_estimator, obs_tensor, gt_detections, initial_estimates) -> Dict[str, PandasTensorCollection]` keys
("final", "refiner/iteration=N", "refiner/final", "coarse", "depth_refiner"; `scene_id` / `view_id` columns added, masks dropped)
and the same `get_predictions(pose_estimator)` result (one concatenated PandasTensorCollection per key).

What differs by design:
  * multi-image batching: `batch_size` frames go through ONE pipeline call (rows of all frames share the launches; `batch_im_id`
    indexes the frame) -- the reference is written for batch_size 1 (`np.unique(scene_id).item()`, :141-142); here scene_id / view_id
    are attached per row from the frame each row belongs to, so any batch size gives the same rows.
  * distributed: frames are dealt round-robin to the ranks (DistributedSceneSampler, reference evaluation/data_utils / :63) and the
    per-rank results are exchanged with `all_gather_object` over the process group (RCCL on the GPU box, gloo in the CPU tests)
    instead of through pickles on a shared filesystem (reference utils/tensor_collection.py:165-186 `gather_distributed`).
Dataset items are duck-typed: mappings (or objects) with `rgb` uint8 [H,W,3], optional `depth` float [H,W], `K` [3,3] (or
`cameras.K`), `gt_detections` (PandasTensorCollection with `label`, `bboxes`; optional `scene_id`/`view_id` columns or item-level
`scene_id`/`view_id`), optional `initial_data` (coarse estimates with `poses`).
"""
from __future__ import annotations

import time
from collections import defaultdict
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import pandas as pd
import torch

from . import distributed as mpdist
from .tcoll import PandasTensorCollection, concatenate
from .types import InferenceConfig, ObservationTensor


def _get(item: Any, name: str, default=None):
    if isinstance(item, dict):
        return item.get(name, default)
    return getattr(item, name, default)


def _item_K(item) -> np.ndarray:
    K = _get(item, "K")
    if K is None:
        K = _get(_get(item, "cameras"), "K")
    return np.asarray(K, dtype=np.float32).reshape(3, 3)


class PredictionRunner:
    def __init__(self, scene_ds: Sequence[Any], inference_cfg: InferenceConfig, batch_size: int = 1, n_workers: int = 4,
                 device: str = "cuda") -> None:
        self.inference_cfg = inference_cfg
        self.device = torch.device(device)  # where frames are staged for the estimator ("cpu" only for host-logic tests)
        self.rank = mpdist.rank()
        self.world_size = mpdist.world_size()
        self.scene_ds = scene_ds
        self.batch_size = int(batch_size)
        self.n_workers = n_workers  # kept for signature compatibility: frames are staged by the caller's dataset object
        self.load_depth = bool(getattr(scene_ds, "load_depth", False))
        self.frame_ids = list(range(self.rank, len(scene_ds), self.world_size))  # DistributedSceneSampler: rank::world
        self.sampler = self.frame_ids   # reference attribute names (prediction_runner.py:60-76)
        self.tmp_dir = None             # no filesystem hand-off: results travel through the process group
        self.timings: List[Dict[str, float]] = []

    # ------------------------------------------------------------------ batching
    def _collate(self, ids: Sequence[int]):
        """frames -> (ObservationTensor [B,C,H,W] on the GPU, detections with batch_im_id = position in the batch, frame table)"""
        items = [self.scene_ds[i] for i in ids]
        rgb = torch.stack([torch.as_tensor(np.asarray(_get(it, "rgb"))) for it in items]).permute(0, 3, 1, 2)
        depth = None
        if all(_get(it, "depth") is not None for it in items):
            depth = torch.stack([torch.as_tensor(np.asarray(_get(it, "depth"), dtype=np.float32)) for it in items])
        K = torch.from_numpy(np.stack([_item_K(it) for it in items]))
        obs = ObservationTensor.from_torch_batched(rgb, depth, K)
        dets, inits, frames = [], [], []
        for b, (i, it) in enumerate(zip(ids, items)):
            d = _get(it, "gt_detections")
            infos = d.infos.copy()
            infos["batch_im_id"] = b
            sid = infos["scene_id"].iloc[0] if "scene_id" in infos and len(infos) else _get(it, "scene_id", 0)
            vid = infos["view_id"].iloc[0] if "view_id" in infos and len(infos) else _get(it, "view_id", i)
            frames.append(dict(batch_im_id=b, scene_id=sid, view_id=vid))
            dets.append(PandasTensorCollection(infos, **{k: v for k, v in d.tensors.items()}))
            init = _get(it, "initial_data")
            if init is not None and len(init) > 0:
                ii = init.infos.copy()
                ii["batch_im_id"] = b
                inits.append(PandasTensorCollection(ii, **init.tensors))
        detections = concatenate(dets)
        initial = concatenate(inits) if inits else None
        return obs, detections, initial, pd.DataFrame(frames)

    # ------------------------------------------------------------------ one pipeline call
    def run_inference_pipeline(self, pose_estimator, obs_tensor: ObservationTensor, gt_detections: PandasTensorCollection,
                               initial_estimates: Optional[PandasTensorCollection] = None,
                               frames: Optional[pd.DataFrame] = None) -> Dict[str, PandasTensorCollection]:
        cfg = self.inference_cfg
        if cfg.detection_type == "gt":
            detections, run_detector = gt_detections, False
        elif cfg.detection_type == "detector":
            detections, run_detector = None, True
        else:
            raise ValueError(f"Unknown detection type {cfg.detection_type}")
        coarse_estimates = None
        if cfg.coarse_estimation_type == "external":
            from .pose_estimator import add_instance_id

            coarse_estimates = add_instance_id(initial_estimates)
            coarse_estimates.infos["instance_id"] = 0
            run_detector = False
        preds, extra_data = pose_estimator.run_inference_pipeline(
            obs_tensor, detections=detections, run_detector=run_detector, coarse_estimates=coarse_estimates,
            n_refiner_iterations=cfg.n_refiner_iterations, n_pose_hypotheses=cfg.n_pose_hypotheses,
            run_depth_refiner=cfg.run_depth_refiner, bsz_images=cfg.bsz_images, bsz_objects=cfg.bsz_objects)
        refiner_final = extra_data["refiner"]["preds"]
        all_preds = {"final": preds, f"refiner/iteration={cfg.n_refiner_iterations}": refiner_final, "refiner/final": refiner_final,
                     "coarse": extra_data["coarse"]["preds"]}
        if cfg.run_depth_refiner:
            all_preds["depth_refiner"] = extra_data["depth_refiner"]["preds"]
        if frames is None:  # reference behaviour: one frame per call
            frames = pd.DataFrame([dict(batch_im_id=0, scene_id=np.unique(gt_detections.infos["scene_id"]).item(),
                                        view_id=np.unique(gt_detections.infos["view_id"]).item())])
        sid = dict(zip(frames["batch_im_id"], frames["scene_id"]))
        vid = dict(zip(frames["batch_im_id"], frames["view_id"]))
        out = {}
        for k, v in all_preds.items():
            if v is None:  # coarse stage skipped (external coarse estimates)
                continue
            v = PandasTensorCollection(v.infos.copy(), **v.tensors)
            v.infos["scene_id"] = v.infos["batch_im_id"].map(sid).values
            v.infos["view_id"] = v.infos["batch_im_id"].map(vid).values
            if "mask" in v.tensors:
                v.delete_tensor("mask")
            out[k] = v
        return out

    @property
    def dataloader(self):
        """iterable of collated batches (observation, detections, initial estimates, frame table), like the reference's DataLoader"""
        return (self._collate(self.frame_ids[i : i + self.batch_size]) for i in range(0, len(self.frame_ids), self.batch_size))

    def _sync(self) -> None:
        if self.device.type == "cuda":
            torch.cuda.synchronize()

    # ------------------------------------------------------------------ the loop
    def get_predictions(self, pose_estimator, gather: bool = True) -> Dict[str, PandasTensorCollection]:
        predictions_list = defaultdict(list)
        self.timings = []
        for n, start in enumerate(range(0, len(self.frame_ids), self.batch_size)):
            ids = self.frame_ids[start : start + self.batch_size]
            obs, detections, initial, frames = self._collate(ids)
            obs = ObservationTensor(obs.images.to(self.device), obs.K.to(self.device))
            detections = detections.to(self.device)
            initial = initial.to(self.device) if initial is not None else None
            if n == 0:  # warm-up call, as the reference does for its timings (:182-187)
                with torch.no_grad():
                    self.run_inference_pipeline(pose_estimator, obs, detections, initial, frames)
            self._sync()
            t0 = time.perf_counter()
            with torch.no_grad():
                all_preds = self.run_inference_pipeline(pose_estimator, obs, detections, initial, frames)
            self._sync()
            self.timings.append({"n_frames": len(ids), "n_detections": len(detections), "seconds": time.perf_counter() - t0})
            for k, v in all_preds.items():
                predictions_list[k].append(v.cpu())
        predictions = {k: concatenate(v) for k, v in predictions_list.items()}
        if gather and self.world_size > 1:
            predictions = gather_predictions(predictions)
        return predictions


def gather_predictions(local: Dict[str, PandasTensorCollection]) -> Dict[str, PandasTensorCollection]:
    """Every rank receives the concatenation of all ranks' predictions (rank order), one collective for the whole dict."""
    if mpdist.world_size() <= 1:
        return local
    import torch.distributed as dist

    payload = {k: (v.infos, {n: t.cpu() for n, t in v.tensors.items()}) for k, v in local.items()}
    parts: List[Any] = [None] * mpdist.world_size()
    dist.all_gather_object(parts, payload)
    keys: List[str] = []
    for p in parts:
        keys += [k for k in p if k not in keys]
    out = {}
    for k in keys:
        out[k] = concatenate([PandasTensorCollection(p[k][0], **p[k][1]) for p in parts if k in p])
    return out
