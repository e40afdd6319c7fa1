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
    """reference panda3d_renderer/types.py:104-114.  `positioning_function(root_node, light_node)` places a point light exactly as
    in the reference (it reads `root_node.getBounds().radius` and calls `light_node.setPos(...)`); the engine evaluates it with
    recording stand-ins (`resolve_light_position`).  `direction` is an engine shortcut: position = direction * 10 * radius."""
    light_type: str
    color: RgbaColor = (1.0, 1.0, 1.0, 1.0)
    positioning_function: Optional[Callable] = None
    direction: Optional[Tuple[float, float, float]] = None


_POINT_DIRS = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]


def _scene_light_pos_fn(root_node, light_node, pos) -> None:
    """panda3d_scene_renderer.py:121-126: the light sits at pos * 10 * (bounding radius of the scene root)"""
    radius = root_node.getBounds().radius
    light_node.setPos(tuple(float(p) * radius * 10 for p in pos))


def make_scene_lights(ambient_light_color: RgbaColor = (0.1, 0.1, 0.1, 1.0),
                      point_lights_color: RgbaColor = (0.4, 0.4, 0.4, 1.0)) -> List[Panda3dLightData]:
    """1 ambient + 6 point lights on the +-axes at 10 x the bounding radius (panda3d_scene_renderer.py:104-136)."""
    from functools import partial

    lights = [Panda3dLightData(light_type="ambient", color=ambient_light_color)]
    for d in _POINT_DIRS:
        lights.append(Panda3dLightData(light_type="point", color=point_lights_color, positioning_function=partial(_scene_light_pos_fn, pos=d)))
    return lights


class _ProbeBounds:
    def __init__(self, radius: float):
        self.radius = radius

    def getRadius(self) -> float:
        return self.radius

    get_radius = getRadius


class _ProbeRoot:
    """what a positioning_function may ask the scene root: getBounds().radius"""

    def __init__(self, radius: float):
        self._bounds = _ProbeBounds(radius)

    def getBounds(self) -> _ProbeBounds:
        return self._bounds

    get_bounds = getBounds


class _ProbeLight:
    def __init__(self) -> None:
        self.pos = None

    def setPos(self, *a) -> None:
        self.pos = tuple(float(v) for v in (a[0] if len(a) == 1 else a))

    set_pos = setPos


def resolve_light_position(positioning_function: Callable) -> Tuple[Tuple[float, float, float], Tuple[float, float, float]]:
    """-> (a, b): the light's object-frame position is a * bounding_radius + b.  The function is called with stand-ins for the
    panda3d nodes at radius 1, 2 and 4; anything that is not affine in the radius (or touches other NodePath API) raises
    NotImplementedError instead of rendering a silently wrong light rig."""
    pts = []
    for r in (1.0, 2.0, 4.0):
        light = _ProbeLight()
        try:
            positioning_function(_ProbeRoot(r), light)
        except AttributeError as e:
            raise NotImplementedError(f"positioning_function uses panda3d API the engine does not emulate: {e}") from e
        if light.pos is None or len(light.pos) != 3:
            raise NotImplementedError("positioning_function did not call setPos(x, y, z) on the light node")
        pts.append(light.pos)
    a = tuple(p2 - p1 for p1, p2 in zip(pts[0], pts[1]))
    b = tuple(p1 - ai for p1, ai in zip(pts[0], a))
    for k in range(3):
        if abs(a[k] * 4.0 + b[k] - pts[2][k]) > 1e-9 * (1.0 + abs(pts[2][k])):
            raise NotImplementedError("positioning_function is not affine in the bounding radius")
    return a, b


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
