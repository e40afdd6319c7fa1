"""TEST INFRASTRUCTURE ONLY.  The oracle's own way from an object dataset to the arrays its renderer and pose math consume, so that
the parity harness does not build the oracle's inputs with product code (megapose6d_amd.mesh_io / mesh_db / pose_estimator).

What the reference does on this path (none of it runnable offline: trimesh and Panda3D are absent):
  * src/megapose/lib3d/rigid_mesh_database.py:49-75   trimesh.load(mesh_path, process=False, maintain_order=True); points = vertices * scale
  * src/megapose/lib3d/rigid_mesh_database.py:83-104  batched(): every object's vertices, padded to the longest set (pad_stack_tensors,
                                                      fill="select_random", deterministic=True -> np.random.RandomState(0))
  * src/megapose/panda3d_renderer/panda3d_scene_renderer.py:146-170  the same file loaded by Panda3D (loader.load_model), scaled by
                                                      obj.scale, vertex colours as stored, smooth normals
  * src/megapose/utils/transform_utils.py:27-50       the SO(3) grid: xyzw quaternions -> rotation matrices
Written from the PLY format definition (ascii / binary_little_endian, scalar vertex properties, one index list per face), not from the
product's reader; tests/test_oracle_loader_cpu.py holds the two against each other on the datasets the parity tests use.
"""
from __future__ import annotations

import struct
from pathlib import Path
from typing import Dict, List

import numpy as np
import torch

from . import geometry as og

_SCALAR = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h", "ushort": "H", "uint16": "H",
           "int": "i", "int32": "i", "uint": "I", "uint32": "I", "float": "f", "float32": "f", "double": "d", "float64": "d"}


def _parse_header(fh):
    if fh.readline().strip() != b"ply":
        raise ValueError("not a PLY file")
    fmt, elements = None, []
    for raw in iter(fh.readline, b""):
        words = raw.decode("ascii", "replace").split()
        if not words or words[0] in ("comment", "obj_info"):
            continue
        if words[0] == "format":
            fmt = words[1]
        elif words[0] == "element":
            elements.append((words[1], int(words[2]), []))
        elif words[0] == "property":
            elements[-1][2].append(words[1:])
        elif words[0] == "end_header":
            return fmt, elements
    raise ValueError("PLY header without end_header")


def read_ply_arrays(path) -> Dict[str, np.ndarray]:
    """-> vertices [V,3] float64, faces [T,3] int64 (polygons as fans around their first corner), optional normals [V,3], colors [V,3] in
    0..1.  One value at a time through `struct`: slow and plain on purpose (the test meshes have 5-10 k vertices)."""
    with open(path, "rb") as fh:
        fmt, elements = _parse_header(fh)
        if fmt not in ("ascii", "binary_little_endian"):
            raise ValueError(f"{path}: PLY format {fmt} not handled by the oracle loader")
        ascii_mode = fmt == "ascii"
        columns: Dict[str, List[float]] = {}
        int_columns = set()
        triangles: List[List[int]] = []

        def scalar(kind, tokens):
            if ascii_mode:
                return float(tokens.pop(0))
            code = _SCALAR[kind]
            return struct.unpack("<" + code, fh.read(struct.calcsize(code)))[0]

        for name, count, props in elements:
            if name == "vertex":
                for p in props:
                    columns[p[-1]] = []
                    if _SCALAR[p[0]] not in "fd":
                        int_columns.add(p[-1])
            for _ in range(count):
                tokens = fh.readline().decode("ascii").split() if ascii_mode else None
                for p in props:
                    if p[0] == "list":
                        n = int(scalar(p[1], tokens))
                        values = [scalar(p[2], tokens) for _ in range(n)]
                        if name == "face" and p[-1] in ("vertex_indices", "vertex_index"):
                            corners = [int(v) for v in values]
                            triangles += [[corners[0], corners[i], corners[i + 1]] for i in range(1, len(corners) - 1)]
                    else:
                        v = scalar(p[0], tokens)
                        if name == "vertex":
                            columns[p[-1]].append(v)
    out = {"vertices": np.array([columns["x"], columns["y"], columns["z"]], dtype=np.float64).T,
           "faces": np.array(triangles, dtype=np.int64).reshape(-1, 3)}
    if all(k in columns for k in ("nx", "ny", "nz")):
        out["normals"] = np.array([columns["nx"], columns["ny"], columns["nz"]], dtype=np.float64).T
    if all(k in columns for k in ("red", "green", "blue")):
        rgb = np.array([columns["red"], columns["green"], columns["blue"]], dtype=np.float64).T
        out["colors"] = rgb / 255.0 if "red" in int_columns or rgb.max() > 1.0 else rgb
    return out


def smooth_normals(vertices: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Unit vertex normals: sum of the (un-normalised, i.e. area-weighted) face normals of the faces around a vertex -- the contract of the
    engine for a file without normals.  Explicit loop over faces."""
    acc = np.zeros_like(vertices, dtype=np.float64)
    fns = [np.cross(vertices[b] - vertices[a], vertices[c] - vertices[a]) for a, b, c in faces]
    for corner in range(3):          # (corner-major accumulation: the summation order is part of the contract's bits)
        for t, face in enumerate(faces):
            acc[face[corner]] += fns[t]
    length = np.sqrt((acc * acc).sum(1))
    length[length == 0.0] = 1.0
    return acc / length[:, None]


def load_object(obj) -> Dict[str, np.ndarray]:
    """One RigidObject (label, mesh_path, scale; datasets/object_dataset.py:35-137) -> float32 vertices (metres), normals, colours 0..1,
    int32 faces, points (= the vertices the pose math uses).  PLY without textures and without a renderer-only orientation offset: what
    the parity datasets are made of; anything else is refused rather than guessed."""
    path = Path(obj.mesh_path)
    if path.suffix.lower() != ".ply":
        raise ValueError(f"oracle loader: {path.name}: only PLY meshes (the parity datasets); textured / OBJ inputs are tested at the raster level")
    if any(abs(float(a)) > 0 for a in getattr(obj, "ypr_offset_deg", (0.0, 0.0, 0.0))):
        raise ValueError("oracle loader: ypr_offset_deg is not handled")
    raw = read_ply_arrays(path)
    metres = raw["vertices"] * float(obj.scale)
    normals = raw["normals"] if "normals" in raw else smooth_normals(raw["vertices"], raw["faces"])
    colors = raw["colors"] if "colors" in raw else np.ones_like(metres)
    return {"vertices": metres.astype(np.float32), "normals": normals.astype(np.float32), "colors": colors.astype(np.float32),
            "faces": raw["faces"].astype(np.int32), "points": metres.astype(np.float32)}


class OraclePointSets:
    """labels + [n_obj, n_max, 3] padded vertex sets (rigid_mesh_database.py:83-104 with the deterministic random fill)."""

    def __init__(self, labels: List[str], points: torch.Tensor):
        self.labels = np.asarray(labels)
        self.points = points


def load_dataset(ds):
    """-> ({label: mesh dict}, OraclePointSets) for a RigidObjectDataset-like (`list_objects`)"""
    objects = list(ds.list_objects)
    meshes = {o.label: load_object(o) for o in objects}
    points = og.pad_stack_points([torch.from_numpy(meshes[o.label]["points"]) for o in objects]).float()
    return meshes, OraclePointSets([o.label for o in objects], points)


def load_so3_grid(path) -> torch.Tensor:
    """xyzw quaternion table (.npy, the converted form of the reference's data file) -> [N,3,3] fp32 (transform_utils.py:27-50)"""
    quats = torch.tensor(np.load(path).tolist())
    return og.load_SO3_grid_from_quats(quats)
