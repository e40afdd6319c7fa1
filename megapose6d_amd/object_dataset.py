"""Object-dataset carriers the renderer / mesh database constructors take.

Duck types of the reference's `RigidObject` / `RigidObjectDataset` (src/megapose/datasets/object_dataset.py:35-166): same
constructor arguments and attributes, so a reference dataset object can be passed to `Panda3dBatchRenderer` / `MeshDataBase`
in their place and vice versa.
"""
from __future__ import annotations

from pathlib import Path


class RigidObject:
    """Duck type of the reference RigidObject (src/megapose/datasets/object_dataset.py:35-137)."""

    def __init__(self, label, mesh_path, mesh_units="m", scaling_factor=1.0, ypr_offset_deg=(0.0, 0.0, 0.0),
                 scaling_factor_mesh_units_to_meters=None, **_):
        self.label = label
        self.mesh_path = Path(mesh_path)
        self.mesh_units = mesh_units
        self.scaling_factor_mesh_units_to_meters = (
            scaling_factor_mesh_units_to_meters if scaling_factor_mesh_units_to_meters is not None else {"m": 1.0, "mm": 0.001}[mesh_units]
        )
        self.scaling_factor = scaling_factor
        self.ypr_offset_deg = ypr_offset_deg
        self.symmetries_discrete, self.symmetries_continuous = [], []
        self.diameter_meters = None

    @property
    def scale(self) -> float:
        return self.scaling_factor_mesh_units_to_meters * self.scaling_factor


class RigidObjectDataset:
    """Duck type of the reference RigidObjectDataset (object_dataset.py:140-166)."""

    def __init__(self, objects):
        self.list_objects = list(objects)
        self.label_to_objects = {o.label: o for o in self.list_objects}
        if len(self.label_to_objects) != len(self.list_objects):
            raise RuntimeError("There are objects with duplicate labels")

    def __getitem__(self, i):
        return self.list_objects[i]

    def __len__(self):
        return len(self.list_objects)

    def get_object_by_label(self, label):
        return self.label_to_objects[label]

    @property
    def objects(self):
        return self.list_objects
