"""Scene annotation types of the inference caller: `Transform`, `ObjectData`, `CameraData` (+ `make_detections_from_object_data`).

Same constructors, JSON layout and accessors as the reference's src/megapose/lib3d/transform.py:27-119,
src/megapose/datasets/scene_dataset.py:67-165 and src/megapose/inference/utils.py:214-225 -- what its inference script
(src/megapose/scripts/run_inference_on_example.py) reads (`camera_data.json`, `inputs/object_data.json`) and writes
(`outputs/object_data.json`).  `Transform` is plain numpy here; the reference wraps pinocchio's SE3 / Eigen quaternions, whose
matrix -> quaternion conversion (Eigen `QuaternionBase::operator=(MatrixBase)`, the trace / largest-diagonal branches) is restated
below so that the written JSON carries the same xyzw coefficients.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

Resolution = Tuple[int, int]


class _Quaternion:
    """the two members of pinocchio.Quaternion the reference touches: `coeffs()` (xyzw) and `matrix()`"""

    def __init__(self, xyzw: np.ndarray):
        self._q = np.asarray(xyzw, dtype=np.float64).reshape(4)

    def coeffs(self) -> np.ndarray:
        return self._q.copy()

    def matrix(self) -> np.ndarray:
        return quaternion_xyzw_to_matrix(self._q)


def quaternion_xyzw_to_matrix(q) -> np.ndarray:
    x, y, z, w = (float(v) for v in q)
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def matrix_to_quaternion_xyzw(R) -> np.ndarray:
    """Eigen's rotation-matrix -> quaternion (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other, 3, 3>)"""
    m = np.asarray(R, dtype=np.float64)
    q = np.zeros(4)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0], q[1], q[2] = (m[2, 1] - m[1, 2]) * t, (m[0, 2] - m[2, 0]) * t, (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


class Transform:
    """SE(3) element.  Transform(T [4,4]) or Transform(rotation, translation) with rotation an xyzw quaternion (4 values, normalised
    on the way in) or a 3x3 matrix -- the reference's constructor forms (lib3d/transform.py:30-91)."""

    def __init__(self, *args: Any):
        if len(args) == 1:
            T = args[0]
            if isinstance(T, Transform):
                T = T.matrix
            if isinstance(T, torch.Tensor):
                T = T.detach().cpu().numpy()
            T = np.asarray(T, dtype=np.float64)
            if T.shape != (4, 4):
                raise ValueError
            R, t = T[:3, :3].copy(), T[:3, 3].copy()
        elif len(args) == 2:
            rot, t = args
            if isinstance(rot, _Quaternion):
                R = rot.matrix()
            else:
                if isinstance(rot, torch.Tensor):
                    rot = rot.detach().cpu().numpy()
                rot = np.asarray(rot, dtype=np.float64)
                if rot.size == 4:
                    q = rot.flatten()
                    R = quaternion_xyzw_to_matrix(q / np.linalg.norm(q))
                elif rot.size == 9:
                    assert rot.shape == (3, 3)
                    R = rot.copy()
                else:
                    raise ValueError
            if isinstance(t, torch.Tensor):
                t = t.detach().cpu().numpy()
            t = np.asarray(t, dtype=np.float64).reshape(3).copy()
        else:
            raise ValueError
        self._R, self._t = R, t

    def __mul__(self, other: "Transform") -> "Transform":
        return Transform(self._R @ other._R, self._R @ other._t + self._t)

    def inverse(self) -> "Transform":
        return Transform(self._R.T, -self._R.T @ self._t)

    def __str__(self) -> str:
        return f"  R =\n{self._R}\n  p = {self._t}\n"

    def toHomogeneousMatrix(self) -> np.ndarray:
        return self.matrix

    @property
    def translation(self) -> np.ndarray:
        return self._t.reshape(3)

    @property
    def quaternion(self) -> _Quaternion:
        return _Quaternion(matrix_to_quaternion_xyzw(self._R))

    @property
    def matrix(self) -> np.ndarray:
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = self._R, self._t
        return T


def transform_to_list(T: Transform) -> List[List[float]]:
    return [T.quaternion.coeffs().tolist(), T.translation.tolist()]


@dataclass
class ObjectData:
    """datasets/scene_dataset.py:72-127"""
    label: str
    TWO: Optional[Transform] = None
    unique_id: Optional[int] = None
    bbox_amodal: Optional[np.ndarray] = None   # (4,) [xmin, ymin, xmax, ymax]
    bbox_modal: Optional[np.ndarray] = None
    visib_fract: Optional[float] = None
    TWO_init: Optional[Transform] = None

    def to_json(self) -> Dict[str, Any]:
        d: Dict[str, Any] = dict(label=self.label)
        for k in ("TWO", "TWO_init"):
            if getattr(self, k) is not None:
                d[k] = transform_to_list(getattr(self, k))
        for k in ("bbox_amodal", "bbox_modal"):
            if getattr(self, k) is not None:
                d[k] = getattr(self, k).tolist()
        for k in ("visib_fract", "unique_id"):
            if getattr(self, k) is not None:
                d[k] = getattr(self, k)
        return d

    @staticmethod
    def from_json(d: Dict[str, Any]) -> "ObjectData":
        assert isinstance(d, dict)
        label = d["label"]
        assert isinstance(label, str)
        data = ObjectData(label=label)
        for k in ("TWO", "TWO_init"):
            if k in d:
                quat_list, trans_list = d[k]
                assert isinstance(quat_list, list) and isinstance(trans_list, list)
                setattr(data, k, Transform(tuple(quat_list), tuple(trans_list)))
        for k in ("unique_id", "visib_fract"):
            if k in d:
                setattr(data, k, d[k])
        for k in ("bbox_amodal", "bbox_modal"):
            if k in d:
                setattr(data, k, np.array(d[k]))
        return data


@dataclass
class CameraData:
    """datasets/scene_dataset.py:130-165"""
    K: Optional[np.ndarray] = None
    resolution: Optional[Resolution] = None
    TWC: Optional[Transform] = None
    camera_id: Optional[str] = None
    TWC_init: Optional[Transform] = None

    def to_json(self) -> str:
        d: Dict[str, Any] = dict()
        for k in ("TWC", "TWC_init"):
            if getattr(self, k) is not None:
                d[k] = transform_to_list(getattr(self, k))
        if self.K is not None:
            d["K"] = self.K.tolist()
        for k in ("camera_id", "resolution"):
            if getattr(self, k) is not None:
                d[k] = getattr(self, k)
        return json.dumps(d)

    @staticmethod
    def from_json(data_str: str) -> "CameraData":
        d = json.loads(data_str)
        assert isinstance(d, dict)
        data = CameraData()
        for k in ("TWC", "TWC_init"):
            if k in d:
                quat_list, trans_list = d[k]
                assert isinstance(quat_list, list) and isinstance(trans_list, list)
                setattr(data, k, Transform(tuple(quat_list), tuple(trans_list)))
        if "camera_id" in d:
            data.camera_id = d["camera_id"]
        if "K" in d:
            data.K = np.array(d["K"])
        if "resolution" in d:
            assert isinstance(d["resolution"], list)
            h, w = d["resolution"]
            assert isinstance(h, int) and isinstance(w, int)
            data.resolution = (h, w)
        return data


def make_detections_from_object_data(object_data: List[ObjectData]):
    """inference/utils.py:214-225"""
    import pandas as pd

    from .tcoll import PandasTensorCollection

    infos = pd.DataFrame(dict(label=[d.label for d in object_data], batch_im_id=0, instance_id=np.arange(len(object_data))))
    bboxes = torch.as_tensor(np.stack([d.bbox_modal for d in object_data]))
    return PandasTensorCollection(infos=infos, bboxes=bboxes)


def make_cameras(camera_data: List[CameraData]):
    """inference/utils.py:197-211: list of CameraData -> PandasTensorCollection(infos: batch_im_id, resolution; K [B,3,3])"""
    import pandas as pd

    from .tcoll import PandasTensorCollection

    infos, K = [], []
    for n, cam in enumerate(camera_data):
        K.append(torch.tensor(cam.K))
        infos.append(dict(batch_im_id=n, resolution=cam.resolution))
    return PandasTensorCollection(infos=pd.DataFrame(infos), K=torch.stack(K))
