"""API data types of the hot path (same field names and meanings as the reference).

ObservationTensor / InferenceConfig / assert_detections_valid: src/megapose/inference/types.py:77-235
BatchRenderOutput: src/megapose/panda3d_renderer/panda3d_batch_renderer.py:61-71
Panda3dLightData: src/megapose/panda3d_renderer/types.py:104-114; make_scene_lights: panda3d_scene_renderer.py:104-136
PosePredictorOutput: src/megapose/models/pose_rigid.py:50-66
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from .tcoll import PandasTensorCollection

PoseEstimatesType = PandasTensorCollection
DetectionsType = PandasTensorCollection
Resolution = Tuple[int, int]
RgbaColor = Tuple[float, float, float, float]


def assert_detections_valid(detections: DetectionsType) -> None:
    df = detections.infos
    for f in ["batch_im_id", "label", "instance_id"]:
        assert f in df, f"detections.infos missing column {f}"
    assert "bboxes" in detections.tensors, "detections missing tensor bboxes."


@dataclass
class InferenceConfig:
    detection_type: str = "detector"
    coarse_estimation_type: str = "SO3_grid"
    SO3_grid_size: int = 576
    n_refiner_iterations: int = 5
    n_pose_hypotheses: int = 5
    run_depth_refiner: bool = False
    depth_refiner: Optional[str] = None
    bsz_objects: int = 16
    bsz_images: int = 576


@dataclass
class ObservationTensor:
    """images: [B,C,H,W], C=3 (rgb in [0,1]) or 4 (rgb + metric depth); K: [B,3,3]."""

    images: torch.Tensor
    K: Optional[torch.Tensor] = None

    def cuda(self) -> "ObservationTensor":
        self.images = self.images.cuda()
        if self.K is not None:
            self.K = self.K.cuda()
        return self

    @property
    def batch_size(self) -> int:
        return self.images.shape[0]

    @property
    def depth(self) -> torch.Tensor:
        assert self.channel_dim == 4
        return self.images[:, 3]

    @property
    def channel_dim(self) -> int:
        return self.images.shape[1]

    def is_valid(self) -> bool:
        if not self.images.ndim == 4:
            return False
        B, C = self.batch_size, self.channel_dim
        if C not in [3, 4]:
            return False
        if self.K is not None and not self.K.shape == torch.Size([B, 3, 3]):
            return False
        if not self.images.dtype == torch.float:
            return False
        return not bool(torch.max(self.images[:, :3]) > 1)

    @staticmethod
    def from_numpy(rgb: np.ndarray, depth: Optional[np.ndarray] = None, K: Optional[np.ndarray] = None) -> "ObservationTensor":
        assert rgb.dtype == np.uint8
        rgb_tensor = torch.as_tensor(rgb).float() / 255
        if rgb_tensor.shape[-1] == 3:
            rgb_tensor = rgb_tensor.permute(2, 0, 1)
        if depth is not None:
            img_tensor = torch.cat((rgb_tensor, torch.as_tensor(depth).unsqueeze(0)), dim=0)
        else:
            img_tensor = rgb_tensor
        return ObservationTensor(img_tensor.unsqueeze(0), torch.as_tensor(K).float().unsqueeze(0))

    @staticmethod
    def from_torch_batched(rgb: torch.Tensor, depth: torch.Tensor, K: torch.Tensor) -> "ObservationTensor":
        assert rgb.dtype == torch.uint8
        rgb = torch.as_tensor(rgb).float() / 255
        if depth is not None:
            if depth.ndim == 3:
                depth = depth.unsqueeze(1)
            img_tensor = torch.cat((rgb, depth), dim=1)
        else:
            img_tensor = rgb
        return ObservationTensor(img_tensor, torch.as_tensor(K).float())


@dataclass
class BatchRenderOutput:
    """rgbs: (bsz,3,h,w) in [0,1]; normals: (bsz,3,h,w) in [0,1]; depths: (bsz,1,h,w) metres."""

    rgbs: torch.Tensor
    normals: Optional[torch.Tensor]
    depths: Optional[torch.Tensor]


@dataclass
class Panda3dLightData:
    light_type: str
    color: RgbaColor = (1.0, 1.0, 1.0, 1.0)
    positioning_function: Optional[Callable] = None
    direction: Optional[Tuple[float, float, float]] = None  # engine extension: unit direction of a point light (pos = dir*10*radius)


_POINT_DIRS = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]


def make_scene_lights(ambient_light_color: RgbaColor = (0.1, 0.1, 0.1, 1.0),
                      point_lights_color: RgbaColor = (0.4, 0.4, 0.4, 1.0)) -> List[Panda3dLightData]:
    """1 ambient + 6 point lights on the +-axes at 10 x the bounding radius (panda3d_scene_renderer.py:104-136)."""
    lights = [Panda3dLightData(light_type="ambient", color=ambient_light_color)]
    for d in _POINT_DIRS:
        lights.append(Panda3dLightData(light_type="point", color=point_lights_color, direction=d))
    return lights


@dataclass
class PosePredictorOutput:
    TCO_output: torch.Tensor
    TCO_input: torch.Tensor
    renders: Optional[torch.Tensor]
    images_crop: Optional[torch.Tensor]
    TCV_O_input: torch.Tensor
    KV_crop: torch.Tensor
    tCR: torch.Tensor
    labels: List[str]
    K: torch.Tensor
    K_crop: torch.Tensor
    network_outputs: Dict[str, torch.Tensor]
    boxes_rend: torch.Tensor
    boxes_crop: torch.Tensor
    renderings_logits: torch.Tensor
    timing_dict: Dict[str, float] = field(default_factory=dict)
