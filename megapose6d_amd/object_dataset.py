"""Object-dataset carriers the renderer / mesh database constructors take.

Duck types of the reference's `RigidObject` / `RigidObjectDataset` (src/megapose/datasets/object_dataset.py:35-166): same
constructor arguments and attributes, so a reference dataset object can be passed to `Panda3dBatchRenderer` / `MeshDataBase`
in their place and vice versa.
"""
from __future__ import annotations

from pathlib import Path


class RigidObject:
    """Duck type of the reference RigidObject (src/megapose/datasets/object_dataset.py:35-137): the same constructor, positional order
    included, and the same attributes."""

    def __init__(self, label, mesh_path, category=None, mesh_diameter=None, mesh_units="m", symmetries_discrete=[], symmetries_continuous=[],  # noqa: B006 (the reference's defaults)
                 ypr_offset_deg=(0.0, 0.0, 0.0), scaling_factor=1.0, scaling_factor_mesh_units_to_meters=None):
        self.label = label
        self.category = category
        self.mesh_path = Path(mesh_path)
        self.mesh_units = mesh_units
        self.scaling_factor_mesh_units_to_meters = (
            scaling_factor_mesh_units_to_meters if scaling_factor_mesh_units_to_meters is not None else {"m": 1.0, "mm": 0.001}[mesh_units]
        )
        self.scaling_factor = scaling_factor
        # object_dataset.py:105-110 tests `self._mesh_diameter` (always None) instead of the argument, so the reference never stores a
        # diameter: observable behaviour kept (diameter_meters is None), the argument is accepted and remembered
        self._mesh_diameter = None
        self.diameter_meters = None
        self._mesh_diameter_arg = mesh_diameter
        self.symmetries_discrete = list(symmetries_discrete)
        self.symmetries_continuous = list(symmetries_continuous)
        self.ypr_offset_deg = ypr_offset_deg

    @property
    def is_symmetric(self) -> bool:
        return len(self.symmetries_discrete) > 0 or len(self.symmetries_continuous) > 0

    @property
    def scale(self) -> float:
        return self.scaling_factor_mesh_units_to_meters * self.scaling_factor

    def make_symmetry_poses(self, n_symmetries_continuous: int = 64):
        """object_dataset.py:124-137.  Symmetry sets feed the evaluation metrics (out of scope); an object without
        symmetries has the identity alone, which is what the hot path's callers get."""
        import numpy as np

        if self.is_symmetric:
            raise NotImplementedError("symmetry pose sets (lib3d/symmetries.py) are part of the evaluation side, not of the pose engine")
        return np.eye(4, dtype=np.float32)[None]


class RigidObjectDataset:
    """Duck type of the reference RigidObjectDataset (object_dataset.py:140-166)."""

    def __init__(self, objects):
        self.list_objects = list(objects)
        self.label_to_objects = {o.label: o for o in self.list_objects}
        if len(self.label_to_objects) != len(self.list_objects):
            raise RuntimeError("There are objects with duplicate labels")

    def __getitem__(self, idx):
        return self.list_objects[idx]

    def __len__(self):
        return len(self.list_objects)

    def get_object_by_label(self, label):
        return self.label_to_objects[label]

    @property
    def objects(self):
        return self.list_objects

    def filter_objects(self, keep_labels):
        return RigidObjectDataset([o for o in self.list_objects if o.label in keep_labels])
