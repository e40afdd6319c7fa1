"""PoseEstimator: SO(3)-grid hypotheses -> coarse scores -> top-K -> iterative refinement -> re-scoring -> arg-max.

Same public API, return types, DataFrame columns and `extra_data` keys as the reference
src/megapose/inference/pose_estimator.py:52-667, so it drops into
src/megapose/scripts/run_inference_on_example.py:126-148 unchanged.  What differs is the schedule: the reference walks
three Python loops of tiny batches (bsz_images / bsz_objects are 2022-era memory workarounds), gathers one full frame
per hypothesis row and syncs to the host in every batch; here every stage runs as a few large launches over rows that
share the single observation frame by index, and the host is touched once per stage (to fill the DataFrame).
`bsz_images` / `bsz_objects` are accepted and kept as attributes; the engine batches by `max_rows_per_launch`
(set `strict_batching=True` to honour the reference batch sizes exactly).

Multi-GPU: with torch.distributed initialised and `distributed=True`, rows are sharded `rank::world`,
coarse logits are all-gathered (RCCL), every rank computes the same top-K, refine+score runs on the local shard, and
`[rows,17]` (pose + logit) is all-gathered so that every rank returns the full result.
"""
from __future__ import annotations

import time
from collections import defaultdict
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import pandas as pd
import torch

from . import distributed as mpdist
from . import engine as eng
from . import tcoll as tc
from .tcoll import PandasTensorCollection
from .types import DetectionsType, ObservationTensor, PoseEstimatesType, assert_detections_valid

_DATA_DIR = Path(__file__).resolve().parent / "data"


def load_SO3_grid(resolution: int) -> torch.Tensor:
    """xyzw unit quaternions -> [N,3,3] (reference utils/transform_utils.py:27-50; roma.unitquat_to_rotmat formula,
    SURVEY.md App. A.1).  fp32 like the reference (torch.tensor of python floats)."""
    path = _DATA_DIR / f"so3_grid_{resolution}_xyzw.npy"
    assert path.is_file(), f"File {path} not found"
    q = torch.tensor(np.load(path).tolist())  # float32, same rounding as torch.tensor(list of python floats)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def add_instance_id(inputs):
    """reference inference/utils.py:151-171"""
    if "instance_id" in inputs.infos:
        return inputs
    df = inputs.infos.copy()
    df["instance_id"] = df.groupby(["batch_im_id", "label"]).cumcount()
    inputs.infos = df
    return inputs


def filter_detections(detections, labels: Optional[List[str]] = None, one_instance_per_class: bool = False):
    """reference inference/utils.py:174-194"""
    if labels is not None:
        df = detections.infos
        detections = detections[df[df.label.isin(labels)].index.tolist()]
    if one_instance_per_class:
        df = detections.infos.sort_values("score", ascending=False).groupby(["batch_im_id", "label"]).head(1)
        detections = detections[df.index.tolist()]
    return detections


class _Timer:
    def __init__(self):
        self.t0 = time.time()

    def elapsed(self) -> float:
        return time.time() - self.t0


class PoseEstimator(torch.nn.Module):
    """Performs inference for pose estimation."""

    def __init__(self, refiner_model: Optional[torch.nn.Module] = None, coarse_model: Optional[torch.nn.Module] = None,
                 detector_model: Optional[torch.nn.Module] = None, depth_refiner=None, bsz_objects: int = 8, bsz_images: int = 256,
                 SO3_grid_size: int = 576, max_rows_per_launch: int = 576, strict_batching: bool = False,
                 distributed: bool = False, n_streams: int = 1) -> None:
        super().__init__()
        self.coarse_model = coarse_model
        self.refiner_model = refiner_model
        self.detector_model = detector_model
        self.depth_refiner = depth_refiner
        self.bsz_objects = bsz_objects
        self.bsz_images = bsz_images
        self.max_rows_per_launch = max_rows_per_launch
        self.strict_batching = strict_batching
        self.distributed = distributed
        self.n_streams = max(1, int(n_streams))  # chunks are interleaved over this many HIP streams (tails / HBM-bound phases of
        self.min_rows_per_stream = 64             # one chunk overlap the MFMA work of the other)
        self._side_streams: List[torch.cuda.Stream] = []
        if self.refiner_model is not None:
            self.cfg = self.refiner_model.cfg
            self.mesh_db = self.refiner_model.mesh_db
        elif self.coarse_model is not None:
            self.cfg = self.coarse_model.cfg
            self.mesh_db = self.coarse_model.mesh_db
        else:
            raise ValueError("At least one of refiner_model or coarse_model must be specified.")
        self._extents: Optional[torch.Tensor] = None
        if SO3_grid_size is not None:
            self.load_SO3_grid(SO3_grid_size)
        self.eval()
        self.keep_all_outputs = False
        self.keep_all_coarse_outputs = False
        self.refiner_outputs = None
        self.coarse_outputs = None
        self.debug_dict: dict = dict()

    def load_SO3_grid(self, grid_size: int) -> None:
        self._SO3_grid = load_SO3_grid(grid_size).cuda()
        self._extents = None

    @property
    def render_dtype(self) -> torch.dtype:
        """torch.float32 (reference arithmetic, default) or torch.float16 = the "fp16 renders" mode of BASELINE.json configs[4]: both
        pose models store their CNN input (renders + observation crop) as binary16 (`PosePredictor.render_dtype`)."""
        m = self.refiner_model if self.refiner_model is not None else self.coarse_model
        return getattr(m, "render_dtype", torch.float32)

    @render_dtype.setter
    def render_dtype(self, dtype: torch.dtype) -> None:
        for m in (self.coarse_model, self.refiner_model):
            if m is not None:
                m.render_dtype = dtype

    # -- helpers -------------------------------------------------------------------------------------------------
    def _chunk(self, reference_bsz: int) -> int:
        return reference_bsz if self.strict_batching else max(self.max_rows_per_launch, 1)

    def _plan(self, n_rows: int, reference_bsz: int) -> Tuple[int, List[torch.cuda.Stream]]:
        """(rows per chunk, streams to interleave the chunks on).  With >= 2 streams a stage is cut into at least that many
        chunks; every stream has its own CNN-input / backbone / raster workspaces (`slot`)."""
        chunk = self._chunk(reference_bsz)
        ns = 1 if self.strict_batching else self.n_streams
        if ns > 1 and n_rows >= ns * self.min_rows_per_stream:
            chunk = min(chunk, -(-n_rows // ns))
        else:
            ns = 1
        cur = torch.cuda.current_stream()
        if ns == 1:
            return chunk, [cur]
        while len(self._side_streams) < ns:
            self._side_streams.append(torch.cuda.Stream())
        streams = self._side_streams[:ns]
        for st in streams:
            st.wait_stream(cur)
        return chunk, streams

    @staticmethod
    def _join(streams: List[torch.cuda.Stream], tensors) -> None:
        """make the current stream wait for the side streams and adopt the tensors they produced"""
        cur = torch.cuda.current_stream()
        for st in streams:
            if st is not cur:
                cur.wait_stream(st)
        if len(streams) > 1 or (streams and streams[0] is not cur):
            for t in tensors:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)

    def _grid_extents(self) -> torch.Tensor:
        if self._extents is None:  # [n_obj, M, 2]: depends only on (mesh, rotation) -- cosypose_ops.py:198-208
            self._extents = eng.init_extents(self.mesh_db.points, self._SO3_grid)
        return self._extents

    def _shard(self, n: int) -> np.ndarray:
        if self.distributed and mpdist.world_size() > 1:
            return mpdist.shard_indices(n, mpdist.rank(), mpdist.world_size())
        return np.arange(n)

    def _gather(self, local: torch.Tensor, n: int) -> torch.Tensor:
        if self.distributed and mpdist.world_size() > 1:
            return mpdist.gather_rows(local, n, mpdist.rank(), mpdist.world_size())
        return local

    # -- timing ---------------------------------------------------------------------------------------------------
    # Reference semantics (pose_estimator.py:274-275, 422-423; training/utils.py:224-264): `model_time` is a CUDA-event time that is
    # only measured with cuda_timer=True (0.0 otherwise), `render_time` and `time` are host wall clocks around SYNCHRONOUS work.
    # Here every launch is asynchronous, so: cuda_timer=True -> each stage synchronises at its start and end and reports DEVICE
    # times (render_time / model_time = sums of per-chunk HIP-event intervals around the raster+crop launch and the backbone,
    # time = wall clock of the fenced stage); cuda_timer=False -> nothing synchronises, model_time = 0.0 like the reference and
    # render_time / time are HOST ENQUEUE times (they say how long the host was busy, not how long the GPU took).
    @staticmethod
    def _sum_events(event_sets) -> Tuple[float, float]:
        from .pose_rigid import PosePredictor

        r = m = 0.0
        for ev in event_sets:
            dr, dm = PosePredictor.step_times(ev)
            r += dr
            m += dm
        return r, m

    def _is_sharded(self) -> bool:
        return self.distributed and mpdist.world_size() > 1

    # -- coarse ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_coarse_model(self, observation: ObservationTensor, detections: DetectionsType, cuda_timer: bool = False,
                             return_debug_data: bool = False) -> Tuple[PoseEstimatesType, dict]:
        if cuda_timer:
            torch.cuda.synchronize()
        start = time.time()
        assert_detections_valid(detections)
        if return_debug_data and self._is_sharded():
            raise NotImplementedError("return_debug_data with distributed=True: crops/renders are shard-local (rows rank::world); "
                                      "run the debug call on one rank with distributed=False")
        coarse = self.coarse_model
        device = observation.images.device
        B, M = len(detections), self._SO3_grid.shape[0]
        df = detections.infos
        # one row per (detection, grid rotation); detection-major, hypothesis-minor (pose_estimator.py:350-360)
        df_h = df.loc[df.index.repeat(M)].copy()
        df_h["hypothesis_id"] = np.tile(np.arange(M), B)
        df_h["bbox_id"] = np.repeat(df.index.values, M)
        n = B * M
        labels_det = df["label"].tolist()
        im_det = df["batch_im_id"].values.astype(np.int32)
        rows_h = self._shard(n)                       # host copies of every index table: no D2H read-back inside the chunk loop
        det_h = rows_h // M
        det_mesh = torch.tensor(self.mesh_db.ids(labels_det), dtype=torch.int32, device=device)
        det_of_row = torch.as_tensor(det_h, device=device, dtype=torch.long)
        rot_of_row = torch.as_tensor((rows_h % M).astype(np.int32), device=device)
        im_of_row = torch.as_tensor(im_det[det_h], device=device)
        bboxes_all = detections.bboxes.to(device=device, dtype=torch.float32)
        K_rows = observation.K[im_of_row.long()].float()
        TCO_local = eng.init_poses_from_boxes(bboxes_all[det_of_row], K_rows, det_mesh[det_of_row], rot_of_row, self._SO3_grid,
                                              self._grid_extents())
        logits_l, scores_l = [], []
        crops, renders, event_sets = [], [], []
        render_time = model_time = 0.0
        chunk, streams = self._plan(rows_h.size, self.bsz_images)   # after the inputs above are enqueued: side streams wait for them
        n_batches = 0
        for ci, s in enumerate(range(0, rows_h.size, chunk)):
            sl = slice(s, min(s + chunk, rows_h.size))
            labels_ = [labels_det[i] for i in det_h[sl]]
            slot = ci % len(streams)
            with torch.cuda.stream(streams[slot]):
                out_ = coarse.forward_coarse(images=observation.images, K=K_rows[sl], labels=labels_, TCO_input=TCO_local[sl],
                                             cuda_timer=cuda_timer, return_debug_data=return_debug_data, im_ids=im_of_row[sl], slot=slot,
                                             defer_timing=True)
            render_time += out_["render_time"]
            event_sets.append(out_["events"])
            logits_l.append(out_["logits"])
            scores_l.append(out_["scores"])
            if return_debug_data:
                crops.append(out_["images_crop"])
                renders.append(out_["renders"])
            n_batches += 1
        self._join(streams, logits_l + scores_l + crops + renders)
        packed = torch.cat([TCO_local.flatten(1), torch.cat(logits_l), torch.cat(scores_l)], dim=1)  # [rows, 18]
        packed = self._gather(packed, n)
        TCO = packed[:, :16].reshape(n, 4, 4).contiguous()
        logits = packed[:, 16].reshape(B, M)
        scores = packed[:, 17].reshape(B, M)
        bboxes = bboxes_all[torch.arange(n, device=device) // M]
        debug_data = dict()
        if return_debug_data:
            ic, rr = torch.cat(crops), torch.cat(renders)
            debug_data = {"images_crop": ic.reshape([B, M, -1, *ic.shape[-2:]]), "renders": rr.reshape([B, M, -1, *rr.shape[-2:]])}
        host = torch.stack([logits.flatten(), scores.flatten()]).cpu().numpy()  # the stage's single D2H sync
        df_h["coarse_logit"] = host[0]
        df_h["coarse_score"] = host[1]
        if cuda_timer:
            torch.cuda.synchronize()
            render_time, model_time = self._sum_events(event_sets)
        elapsed = time.time() - start
        timing_str = f"time: {elapsed:.2f}, model_time: {model_time:.2f}, render_time: {render_time:.2f}"
        extra_data = {"render_time": render_time, "model_time": model_time, "time": elapsed, "logits": logits, "scores": scores,
                      "TCO": TCO.reshape([B, M, 4, 4]), "debug": debug_data, "n_batches": n_batches, "timing_str": timing_str}
        return PandasTensorCollection(df_h, poses=TCO, bboxes=bboxes), extra_data

    # -- refiner --------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_refiner(self, observation: ObservationTensor, data_TCO_input: PoseEstimatesType, n_iterations: int = 5,
                        keep_all_outputs: bool = False, cuda_timer: bool = False, **refiner_kwargs) -> Tuple[dict, dict]:
        if cuda_timer:
            torch.cuda.synchronize()
        start = time.time()
        assert self.refiner_model is not None
        if keep_all_outputs and self._is_sharded():
            raise NotImplementedError("keep_all_outputs with distributed=True: the per-batch outputs are shard-local")
        device = observation.images.device
        R = data_TCO_input.poses.shape[0]
        rows_h = self._shard(R)
        df = data_TCO_input.infos.copy()  # the reference adds these two columns to its per-batch copies (:155-156)
        labels_all = df["label"].tolist()
        im_h = df["batch_im_id"].values.astype(np.int32)
        im_all = torch.as_tensor(im_h, device=device)
        rows = torch.as_tensor(rows_h, device=device, dtype=torch.long)
        poses_in = data_TCO_input.poses.to(device=device, dtype=torch.float32)
        K_all = observation.K[im_all.long()].float()
        K_rows, poses_rows, im_rows = K_all[rows], poses_in[rows], im_all[rows]
        chunk, streams = self._plan(rows_h.size, self.bsz_objects)   # after the inputs above are enqueued (side streams wait for them)
        df["refiner_batch_idx"] = np.arange(R) // chunk
        df["refiner_instance_idx"] = np.arange(R) % chunk
        keys = ("poses", "poses_input", "K_crop", "boxes_rend", "boxes_crop", "pose_out")
        acc: Dict[int, Dict[str, list]] = {n: {k: [] for k in keys} for n in range(1, n_iterations + 1)}
        all_outputs = []
        event_sets = []
        render_time = model_time = 0.0
        produced = []
        for ci, s in enumerate(range(0, rows_h.size, chunk)):
            sl = slice(s, min(s + chunk, rows_h.size))
            labels_ = [labels_all[i] for i in rows_h[sl]]
            slot = ci % len(streams)
            with torch.cuda.stream(streams[slot]):
                outputs_ = self.refiner_model(images=observation.images, K=K_rows[sl], TCO=poses_rows[sl], n_iterations=n_iterations,
                                              labels=labels_, im_ids=im_rows[sl], materialize=keep_all_outputs, slot=slot,
                                              cuda_timer=cuda_timer, **refiner_kwargs)
            for o in outputs_.values():
                produced += [o.TCO_output, o.TCO_input, o.KV_crop, o.boxes_rend, o.boxes_crop, o.renders, o.images_crop]
                produced += list(o.network_outputs.values())
            if keep_all_outputs:
                all_outputs.append(outputs_)
            for n in range(1, n_iterations + 1):
                o = outputs_[f"iteration={n}"]
                a = acc[n]
                a["poses"].append(o.TCO_output)
                a["poses_input"].append(o.TCO_input)
                a["K_crop"].append(o.K_crop)
                a["boxes_rend"].append(o.boxes_rend)
                a["boxes_crop"].append(o.boxes_crop)
                a["pose_out"].append(o.network_outputs["pose"] if "pose" in o.network_outputs else
                                     torch.zeros(o.TCO_output.shape[0], 9, device=device))
                render_time += o.timing_dict["render"]
                event_sets.append(o.timing_dict.get("events"))
        self._join(streams, produced)
        # ONE all-gather for the whole stage: the 58 floats of every iteration side by side, [rows, n_iterations * 58]
        W = 58
        blocks = []
        for n in range(1, n_iterations + 1):
            a = acc[n]
            if rows_h.size:
                blocks += [torch.cat(a["poses"]).flatten(1), torch.cat(a["poses_input"]).flatten(1), torch.cat(a["K_crop"]).flatten(1),
                           torch.cat(a["boxes_rend"]), torch.cat(a["boxes_crop"]), torch.cat(a["pose_out"])]
        packed_all = torch.cat(blocks, dim=1) if rows_h.size else torch.zeros(0, W * n_iterations, device=device)
        packed_all = self._gather(packed_all, R)
        preds = dict()
        pose_outputs = dict()
        for n in range(1, n_iterations + 1):
            packed = packed_all[:, (n - 1) * W:n * W]
            preds[f"iteration={n}"] = PandasTensorCollection(
                df, poses=packed[:, 0:16].reshape(R, 4, 4).contiguous(), poses_input=packed[:, 16:32].reshape(R, 4, 4).contiguous(),
                K_crop=packed[:, 32:41].reshape(R, 3, 3).contiguous(), K=K_all, boxes_rend=packed[:, 41:45].contiguous(),
                boxes_crop=packed[:, 45:49].contiguous())
            pose_outputs[f"iteration={n}"] = packed[:, 49:58].contiguous()
        if cuda_timer:
            torch.cuda.synchronize()
            render_time, model_time = self._sum_events(event_sets)
        elapsed = time.time() - start
        # `pose_outputs` (engine extension): the refiner network's raw 9-vector per row and iteration, [R, 9] -- what the parity
        # checks compare before the pose update damps it (PosePredictorOutput.network_outputs["pose"], pose_rigid.py:50-66)
        extra_data = {"n_iterations": n_iterations, "outputs": all_outputs, "model_time": model_time, "render_time": render_time,
                      "time": elapsed, "pose_outputs": pose_outputs}
        return preds, extra_data

    # -- scoring --------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_scoring_model(self, observation: ObservationTensor, data_TCO: PoseEstimatesType, cuda_timer: bool = False,
                              return_debug_data: bool = False) -> Tuple[PoseEstimatesType, dict]:
        """Adds 'pose_logit' / 'pose_score' to data_TCO.infos (modifies the collection in place, :217-322)."""
        if cuda_timer:
            torch.cuda.synchronize()
        start = time.time()
        assert self.coarse_model is not None
        if return_debug_data and self._is_sharded():
            raise NotImplementedError("return_debug_data with distributed=True: crops/renders are shard-local (rows rank::world)")
        device = observation.images.device
        df = data_TCO.infos
        R = len(df)
        labels_all = df["label"].tolist()
        im_h = df["batch_im_id"].values.astype(np.int32)
        rows_h = self._shard(R)
        im_all = torch.as_tensor(im_h, device=device)
        rows = torch.as_tensor(rows_h, device=device, dtype=torch.long)
        poses = data_TCO.poses.to(device=device, dtype=torch.float32)
        K_all = observation.K[im_all.long()].float()
        K_rows, poses_rows, im_rows = K_all[rows], poses[rows], im_all[rows]
        chunk, streams = self._plan(rows_h.size, self.bsz_images)
        logits_l, scores_l, crops, renders, event_sets = [], [], [], [], []
        render_time = model_time = 0.0
        n_batches = 0
        for ci, s in enumerate(range(0, rows_h.size, chunk)):
            sl = slice(s, min(s + chunk, rows_h.size))
            slot = ci % len(streams)
            with torch.cuda.stream(streams[slot]):
                out_ = self.coarse_model.forward_coarse(images=observation.images, K=K_rows[sl], labels=[labels_all[i] for i in rows_h[sl]],
                                                        TCO_input=poses_rows[sl], cuda_timer=cuda_timer, return_debug_data=return_debug_data,
                                                        im_ids=im_rows[sl], slot=slot, defer_timing=True)
            render_time += out_["render_time"]
            event_sets.append(out_["events"])
            logits_l.append(out_["logits"])
            scores_l.append(out_["scores"])
            if return_debug_data:
                crops.append(out_["images_crop"])
                renders.append(out_["renders"])
            n_batches += 1
        self._join(streams, logits_l + scores_l + crops + renders)
        packed = torch.cat([torch.cat(logits_l), torch.cat(scores_l)], dim=1) if rows_h.size else torch.zeros(0, 2, device=device)
        packed = self._gather(packed, R)
        logits, scores = packed[:, 0:1].contiguous(), packed[:, 1:2].contiguous()
        debug_data = dict()
        if return_debug_data:
            debug_data = {"images_crop": torch.cat(crops), "renders": torch.cat(renders)}
        host = packed.cpu().numpy()
        df["pose_logit"] = host[:, 0]
        df["pose_score"] = host[:, 1]
        if cuda_timer:
            torch.cuda.synchronize()
            render_time, model_time = self._sum_events(event_sets)
        elapsed = time.time() - start
        timing_str = f"time: {elapsed:.2f}, model_time: {model_time:.2f}, render_time: {render_time:.2f}"
        extra_data = {"render_time": render_time, "model_time": model_time, "time": elapsed, "logits": logits, "scores": scores,
                      "debug": debug_data, "n_batches": n_batches, "timing_str": timing_str}
        data_TCO.infos = df
        return data_TCO, extra_data

    @torch.no_grad()
    def forward_detection_model(self, observation: ObservationTensor, *args: Any, **kwargs: Any) -> DetectionsType:
        if self.detector_model is None:
            raise ValueError("no detector model: pass `detections` (the 2D detector is out of the hot-path scope)")
        return self.detector_model.get_detections(observation, *args, **kwargs)

    def run_depth_refiner(self, observation: ObservationTensor, predictions: PoseEstimatesType) -> Tuple[PoseEstimatesType, dict]:
        assert self.depth_refiner is not None, "You must specify a depth refiner"
        return self.depth_refiner.refine_poses(predictions, depth=observation.depth, K=observation.K)

    # -- pipeline -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def run_inference_pipeline(self, observation: ObservationTensor, detections: Optional[DetectionsType] = None,
                               run_detector: Optional[bool] = None, n_refiner_iterations: int = 5, n_pose_hypotheses: int = 1,
                               keep_all_refiner_outputs: bool = False, detection_filter_kwargs: Optional[dict] = None,
                               run_depth_refiner: bool = False, bsz_images: Optional[int] = None, bsz_objects: Optional[int] = None,
                               cuda_timer: bool = False, coarse_estimates: Optional[PoseEstimatesType] = None
                               ) -> Tuple[PoseEstimatesType, dict]:
        timing_str = ""
        timer = _Timer()
        if bsz_images is not None:
            self.bsz_images = bsz_images
        if bsz_objects is not None:
            self.bsz_objects = bsz_objects
        if coarse_estimates is None:
            assert detections is not None or run_detector, "You must either pass in `detections` or set run_detector=True"
            if detections is None and run_detector:
                t = time.time()
                detections = self.forward_detection_model(observation).cuda()
                timing_str += f"detection={time.time() - t:.2f}, "
            assert detections is not None
            detections = add_instance_id(detections)
            if detection_filter_kwargs is not None:
                detections = filter_detections(detections, **detection_filter_kwargs)
            data_TCO_coarse, coarse_extra_data = self.forward_coarse_model(observation=observation, detections=detections,
                                                                           cuda_timer=cuda_timer)
            timing_str += f"coarse={coarse_extra_data['time']:.2f}, "
            data_TCO_filtered = self.filter_pose_estimates(data_TCO_coarse, top_K=n_pose_hypotheses, filter_field="coarse_logit")
        else:
            data_TCO_coarse = coarse_estimates
            coarse_extra_data = None
            data_TCO_filtered = coarse_estimates
        preds, refiner_extra_data = self.forward_refiner(observation, data_TCO_filtered, n_iterations=n_refiner_iterations,
                                                         keep_all_outputs=keep_all_refiner_outputs, cuda_timer=cuda_timer)
        data_TCO_refined = preds[f"iteration={n_refiner_iterations}"]
        timing_str += f"refiner={refiner_extra_data['time']:.2f}, "
        data_TCO_scored, scoring_extra_data = self.forward_scoring_model(observation, data_TCO_refined, cuda_timer=cuda_timer)
        timing_str += f"scoring={scoring_extra_data['time']:.2f}, "
        data_TCO_final_scored = self.filter_pose_estimates(data_TCO_scored, top_K=1, filter_field="pose_logit")
        if run_depth_refiner:
            t = time.time()
            data_TCO_depth_refiner, _ = self.run_depth_refiner(observation, data_TCO_final_scored)
            data_TCO_final = data_TCO_depth_refiner
            timing_str += f"depth refiner={time.time() - t:.2f}"
        else:
            data_TCO_depth_refiner = None
            data_TCO_final = data_TCO_final_scored
        total = timer.elapsed()
        timing_str = f"total={total:.2f}, {timing_str}"
        extra_data: dict = dict()
        extra_data["coarse"] = {"preds": data_TCO_coarse, "data": coarse_extra_data}
        extra_data["coarse_filter"] = {"preds": data_TCO_filtered}
        extra_data["refiner_all_hypotheses"] = {"preds": preds, "data": refiner_extra_data}
        extra_data["scoring"] = {"preds": data_TCO_scored, "data": scoring_extra_data}
        extra_data["refiner"] = {"preds": data_TCO_final_scored, "data": refiner_extra_data}
        extra_data["timing_str"] = timing_str
        extra_data["time"] = total
        if run_depth_refiner:
            extra_data["depth_refiner"] = {"preds": data_TCO_depth_refiner}
        return data_TCO_final, extra_data

    def filter_pose_estimates(self, data_TCO: PoseEstimatesType, top_K: int, filter_field: str, ascending: bool = False) -> PoseEstimatesType:
        """Keep the top_K rows per (batch_im_id, label, instance_id) by `filter_field` (:643-667).  The sort is made
        stable so that exact ties resolve deterministically (lowest row first); the reference's default quicksort leaves
        tie order unspecified."""
        t0 = time.perf_counter()
        df = data_TCO.infos
        keep = _topk_rows_fast(df, top_K, filter_field, ascending)
        if keep is None:   # (scores with NaN, non-integer ids, ids beyond the packed key's range: the plain pandas form)
            group_cols = ["batch_im_id", "label", "instance_id"]
            keep = df.sort_values(filter_field, ascending=ascending, kind="stable").groupby(group_cols).head(top_K).index.to_numpy()
        out = data_TCO[keep.tolist()]
        mpdist.stats.host_s += time.perf_counter() - t0   # (replicated on every rank: bench.py reports its share of a step)
        return out


def _topk_rows_fast(df, top_K: int, field: str, ascending: bool):
    """Row positions `df.sort_values(field, kind="stable").groupby([batch_im_id, label, instance_id]).head(top_K)` keeps, in that order,
    without pandas' group-by (1.1 -> 0.2 ms on a 576-row table, 15 -> 9 ms on 36 864 rows; every rank of a multi-GPU run repeats it, so it
    is the part of a step that does not shrink with the world size).  Integer group key = batch_im_id | instance_id | factorised label;
    None when that key cannot be formed -- the caller then uses the pandas form, which this reproduces row for row (ties keep the lower
    row first, as the stable sort does: tests/test_host_cpu.py)."""
    import pandas as pd

    if len(df) == 0 or df.index.dtype.kind not in "iu" or not df.index.is_unique:
        return None
    v = df[field].to_numpy()
    c1, c3 = df["batch_im_id"].to_numpy(), df["instance_id"].to_numpy()
    if v.dtype.kind != "f" or c1.dtype.kind not in "iu" or c3.dtype.kind not in "iu" or np.isnan(v).any():
        return None
    c2 = pd.factorize(df["label"].to_numpy())[0]
    if c1.min() < 0 or c1.max() >= 1 << 22 or c3.min() < 0 or c3.max() >= 1 << 24 or c2.min() < 0 or c2.max() >= 1 << 16:
        return None
    n = len(v)
    codes = (c1.astype(np.int64) << 40) | (c3.astype(np.int64) << 16) | c2.astype(np.int64)
    order = np.argsort(v if ascending else -v, kind="stable")   # (negation keeps ties in row order, like ascending=False of a stable sort)
    cs = codes[order]
    o2 = np.argsort(cs, kind="stable")
    sc = cs[o2]
    start = np.r_[True, sc[1:] != sc[:-1]]
    first = np.maximum.accumulate(np.where(start, np.arange(n), 0))
    rank = np.empty(n, dtype=np.int64)
    rank[o2] = np.arange(n) - first
    return df.index.to_numpy()[order[rank < top_K]]


# name used by BASELINE.json's north_star; the reference snapshot only has PoseEstimator
CoarseRefinePoseEstimator = PoseEstimator
