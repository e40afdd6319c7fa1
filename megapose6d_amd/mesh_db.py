"""Mesh database for the pose math: points of every object, padded to a common count, and the deterministic
2000-/200-point subsets the crop logic projects.

Mirrors what the reference builds in src/megapose/lib3d/rigid_mesh_database.py (MeshDataBase :57-88, batched :90-130,
BatchedMeshes.select :146-153, Meshes.sample_points :168-169, pad_stack_tensors :172-200) and
src/megapose/lib3d/mesh_ops.py:77-87 (np.random.RandomState(0).choice without replacement), but samples ONCE per
database on the host (the reference re-draws the same permutation for every batch) and keeps everything resident.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import mesh_io
from .tcoll import TensorCollection


def _pad_points(point_sets: Sequence[torch.Tensor]) -> torch.Tensor:
    """Pad every point set to the longest one with randomly re-drawn points of the same object (seeded; duplicates
    do not change any min/max the pipeline computes)."""
    longest = max(p.shape[0] for p in point_sets)
    draw = np.random.RandomState(0)
    padded = []
    for p in point_sets:
        missing = longest - p.shape[0]
        if missing > 0:
            p = torch.cat((p, p[draw.choice(np.arange(p.shape[0]), size=missing)]), dim=0)
        padded.append(p)
    return torch.stack(padded)


def deterministic_point_ids(n_available: int, n_points: int) -> np.ndarray:
    if n_points > n_available:
        raise AssertionError(f"need at least {n_points} mesh points, got {n_available} (lib3d/mesh_ops.py:79)")
    return np.random.RandomState(0).choice(n_available, size=n_points, replace=False)


class Meshes(TensorCollection):
    def __init__(self, infos, labels, points, symmetries):
        super().__init__(points=points, symmetries=symmetries)
        self.infos = infos
        self.labels = np.asarray(labels)

    def sample_points(self, n_points: int, deterministic: bool = False) -> torch.Tensor:
        n_total = self.points.shape[1]
        if deterministic:
            ids = deterministic_point_ids(n_total, n_points)
        else:
            ids = np.random.choice(n_total, size=n_points, replace=False)
        return torch.index_select(self.points, 1, torch.as_tensor(ids, device=self.points.device))


class BatchedMeshes(TensorCollection):
    """points [n_obj, N_max, 3] (metres, float32) + label bookkeeping; `.select(labels)` gathers rows."""

    def __init__(self, infos, labels, points, symmetries):
        super().__init__(points=points, symmetries=symmetries)
        self.infos = infos
        self.labels = np.asarray(labels)
        self.label_to_id = {label: n for n, label in enumerate(labels)}
        self._sampled: Dict[int, torch.Tensor] = {}

    def ids(self, labels: Sequence[str]) -> List[int]:
        return [self.label_to_id[l] for l in labels]  # KeyError for unknown labels, like the reference

    def select(self, labels: Sequence[str]) -> Meshes:
        ids = self.ids(labels)
        return Meshes([self.infos[l] for l in labels], self.labels[ids], self.points[ids], self.symmetries[ids])

    def sampled_points(self, n_points: int = 2000) -> torch.Tensor:
        """[n_obj, n_points, 3]: the deterministic subset; its first 200 rows are the 200-point subset
        (RandomState.choice(replace=False) = permutation prefix)."""
        if n_points not in self._sampled:
            ids = torch.as_tensor(deterministic_point_ids(self.points.shape[1], n_points), device=self.points.device)
            self._sampled[n_points] = torch.index_select(self.points, 1, ids).contiguous()
        return self._sampled[n_points]

    def to(self, target):
        super().to(target)
        self._sampled = {}
        return self


class MeshDataBase:
    def __init__(self, obj_list):
        self.obj_list = list(obj_list)
        self.obj_dict = {o.label: o for o in self.obj_list}
        self.infos = {o.label: dict() for o in self.obj_list}
        self.engine_meshes = {o.label: mesh_io.load_rigid_object(o) for o in self.obj_list}
        for o in self.obj_list:
            if getattr(o, "diameter_meters", None) is None:
                pts = self.engine_meshes[o.label]["points"].astype(np.float64)
                o.diameter_meters = float(np.linalg.norm(pts.max(0) - pts.min(0)))

    @staticmethod
    def from_object_ds(object_ds) -> "MeshDataBase":
        return MeshDataBase([object_ds[n] for n in range(len(object_ds))])

    @property
    def labels(self) -> List[str]:
        return [o.label for o in self.obj_list]

    def batched(self, aabb: bool = False, resample_n_points: Optional[int] = None, n_sym: int = 64) -> BatchedMeshes:
        if aabb or resample_n_points:
            raise NotImplementedError("only the hot-path configuration (all vertices) is supported")
        pts = [torch.from_numpy(self.engine_meshes[l]["points"]) for l in self.labels]
        infos = {l: {"n_points": int(p.shape[0]), "n_sym": 1} for l, p in zip(self.labels, pts)}
        sym = torch.eye(4).repeat(len(pts), 1, 1, 1)
        return BatchedMeshes(infos, self.labels, _pad_points(pts).float(), sym)
