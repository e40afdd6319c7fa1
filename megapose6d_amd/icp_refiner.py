"""Depth refiners: `DepthRefiner` interface + on-device `ICPRefiner`.

Same class names / constructor / `refine_poses` contract as the reference's src/megapose/inference/depth_refiner.py:29-51 and
src/megapose/inference/icp_refiner.py:178-262: render depth at the image resolution for every prediction, mask by
|measured - rendered| <= 0.1 m, refine with ICP, keep the input pose when the refinement is rejected.  The per-object CPU loop
(numpy back-projection, cv2.inpaint/gaussian normals, OpenCV ppf_match_3d_ICP) becomes one batched device call
(`mp_icp_refine_nn`, csrc/icp_nn.hip: the reference's algorithm step for step; OpenCV's ICP is third-party, restated in
oracle/icp_opencv.py, parity unpinned) -- or the cheaper projective-association variant `mp_icp_refine` (csrc/icp.hip).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional, Tuple

import torch

from . import engine as eng
from .types import Panda3dLightData, PoseEstimatesType


class DepthRefiner(ABC):
    @abstractmethod
    def refine_poses(self, predictions: PoseEstimatesType, masks: Optional[torch.Tensor] = None, depth: Optional[torch.Tensor] = None,
                     K: Optional[torch.Tensor] = None) -> Tuple[PoseEstimatesType, dict]:
        """predictions: N rows indexing depth / masks / K through `batch_im_id`; depth [B,H,W]; masks [B,H,W]; K [B,3,3].
        Returns (refined predictions with `poses` replaced, extra_data)."""


class ICPRefiner(DepthRefiner):
    def __init__(self, mesh_db, renderer, n_iterations: int = 100, n_levels: int = 4, tolerance: float = 0.05, n_min_points: int = 1000,
                 association: str = "nn"):
        # association (engine extension): "nn" = the reference's algorithm on the device (get_normal + OpenCV-style nearest-neighbour ICP,
        # csrc/icp_nn.hip); "projective" = the cheaper projective-association ICP of csrc/icp.hip
        self.association = association
        self.mesh_db = mesh_db
        self.renderer = renderer
        self.light_datas = [Panda3dLightData("ambient")]
        self.n_iterations, self.n_levels, self.tolerance, self.n_min_points = n_iterations, n_levels, tolerance, n_min_points

    @torch.no_grad()
    def refine_poses(self, predictions: PoseEstimatesType, masks: Optional[torch.Tensor] = None, depth: Optional[torch.Tensor] = None,
                     K: Optional[torch.Tensor] = None) -> Tuple[PoseEstimatesType, dict]:
        assert depth is not None
        assert K is not None
        predictions_refined = predictions.clone()
        df = predictions.infos
        labels = df.label.tolist()
        N = len(predictions)
        device = depth.device
        im_ids = torch.as_tensor(df.batch_im_id.values.astype("int32"), device=device)
        TCO_ = predictions.poses.to(device=device, dtype=torch.float32)
        K_ = K[im_ids.long()].float()
        resolution = tuple(depth.shape[-2:])
        # depth at the pixel centres (not the off-centre sample a multisample depth resolve returns): it is back-projected below
        depth_rendered = self.renderer.render_depth(labels, TCO_, K_, resolution).contiguous()
        depth_meas = depth.float()
        if self.association == "nn":  # the caller's masks REPLACE the threshold mask (icp_refiner.py:249-250); they only select points
            refined, retval, residual, iters = eng.icp_refine(depth_meas.contiguous(), im_ids, depth_rendered, K.float(), K_, TCO_,
                                                              self.n_iterations, self.n_levels, self.tolerance, self.n_min_points,
                                                              association="nn", return_iters=True,
                                                              masks=None if masks is None else masks.reshape(depth_meas.shape))
        else:
            iters = None
            if masks is not None:
                depth_meas = depth_meas * (masks.to(depth_meas.dtype) > 0)
            refined, retval, residual = eng.icp_refine(depth_meas.contiguous(), im_ids, depth_rendered, K.float(), K_, TCO_, self.n_iterations,
                                                       self.n_levels, self.tolerance, self.n_min_points, user_masks=masks is not None,
                                                       association="projective")
        if "poses_input" in predictions_refined.tensors:
            predictions_refined.poses_input = predictions.poses.clone()
        else:
            predictions_refined.register_tensor("poses_input", predictions.poses.clone())
        predictions_refined.poses = refined
        extra = {"retval": retval, "residual": residual}
        if iters is not None:
            extra["iterations_per_level"] = iters[:, : self.n_levels]   # (telemetry; index = pyramid level, coarsest = n_levels - 1)
        return predictions_refined, extra
