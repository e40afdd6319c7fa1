"""Mesh ingestion for the engine: RigidObject -> float32 arrays for mp_mesh_db_create.

Replaces, for the hot path, `trimesh.load(..., process=False, maintain_order=True)`
(reference src/megapose/lib3d/rigid_mesh_database.py:62-73) and Panda3D's assimp loader
(src/megapose/panda3d_renderer/panda3d_scene_renderer.py:192-207: scale =
scaling_factor_mesh_units_to_meters * scaling_factor, HPR offset applied in the renderer only).
Vertex ORDER is preserved (the deterministic point sampling indexes it, lib3d/mesh_ops.py:77-87).

Supported: PLY (ascii / binary_little_endian; x y z [nx ny nz] [red green blue [alpha]]; polygon faces
are fan-triangulated) and OBJ (v / vn / f with per-vertex colours `v x y z r g b`).  UV textures are a
"next" row: meshes without vertex colours render white, as Panda3D does for
untextured, uncoloured geometry.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional

import numpy as np

_PLY_T = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1",
          "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
          "uint": "u4", "uint32": "u4"}


def _fan(polys) -> np.ndarray:
    tris = []
    for p in polys:
        for i in range(1, len(p) - 1):
            tris.append((p[0], p[i], p[i + 1]))
    return np.asarray(tris, dtype=np.int32).reshape(-1, 3)


def read_ply(path) -> Dict[str, Optional[np.ndarray]]:
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt = None
        elems = []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elems.append({"name": tok[1], "count": int(tok[2]), "props": []})
            elif tok[0] == "property":
                elems[-1]["props"].append(tok[1:])
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        vp: Dict[str, np.ndarray] = {}
        faces = np.zeros((0, 3), np.int32)
        for el in elems:
            if el["name"] == "vertex":
                names = [p[-1] for p in el["props"]]
                if fmt == "ascii":
                    rows = [f.readline().split() for _ in range(el["count"])]
                    arr = np.asarray(rows, dtype=np.float64).reshape(el["count"], len(names))
                    vp = {n: arr[:, i] for i, n in enumerate(names)}
                else:
                    dt = np.dtype([(p[-1], "<" + _PLY_T[p[0]]) for p in el["props"]])
                    arr = np.frombuffer(f.read(dt.itemsize * el["count"]), dtype=dt)
                    vp = {n: arr[n] for n in names}
            elif el["name"] == "face":
                polys = []
                if fmt == "ascii":
                    for _ in range(el["count"]):
                        tok = f.readline().split()
                        n = int(tok[0])
                        polys.append([int(t) for t in tok[1 : 1 + n]])
                else:
                    lp = [p for p in el["props"] if p[0] == "list"][0]
                    ct, it = np.dtype("<" + _PLY_T[lp[1]]), np.dtype("<" + _PLY_T[lp[2]])
                    extra = [p for p in el["props"] if p[0] != "list"]
                    if extra:
                        raise ValueError(f"{path}: face elements with extra properties are not supported")
                    for _ in range(el["count"]):
                        n = int(np.frombuffer(f.read(ct.itemsize), dtype=ct)[0])
                        polys.append(np.frombuffer(f.read(it.itemsize * n), dtype=it).tolist())
                faces = _fan(polys)
            else:  # skip unknown elements (ascii only)
                if fmt == "ascii":
                    for _ in range(el["count"]):
                        f.readline()
                else:
                    raise ValueError(f"{path}: unsupported binary PLY element {el['name']}")
    verts = np.stack([vp["x"], vp["y"], vp["z"]], axis=1).astype(np.float64)
    normals = np.stack([vp["nx"], vp["ny"], vp["nz"]], axis=1).astype(np.float64) if "nx" in vp else None
    colors = None
    if "red" in vp:
        colors = np.stack([vp["red"], vp["green"], vp["blue"]], axis=1)
        colors = colors.astype(np.float64) / (255.0 if colors.dtype.kind in "ui" or colors.max() > 1.0 else 1.0)
    return {"vertices": verts, "faces": faces, "normals": normals, "colors": colors}


def read_obj(path) -> Dict[str, Optional[np.ndarray]]:
    verts, cols, polys = [], [], []
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append([float(t) for t in tok[1:4]])
                if len(tok) >= 7:
                    cols.append([float(t) for t in tok[4:7]])
            elif tok[0] == "f":
                polys.append([int(t.split("/")[0]) - 1 for t in tok[1:]])
    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    c = np.asarray(cols, dtype=np.float64) if len(cols) == len(verts) and cols else None
    return {"vertices": v, "faces": _fan(polys), "normals": None, "colors": c}


def vertex_normals(vertices: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Area-weighted vertex normals (what assimp's aiProcess_GenSmoothNormals-style loaders produce)."""
    v = vertices.astype(np.float64)
    fn = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, faces[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    return n / ln


def _ypr_matrix(ypr_deg) -> np.ndarray:
    """Panda3D setHpr(h, p, r): heading about Z, pitch about X, roll about Y (Z-up), applied as R = Rz(h) Rx(p) Ry(r)."""
    h, p, r = np.deg2rad(np.asarray(ypr_deg, dtype=np.float64))
    Rz = np.array([[np.cos(h), -np.sin(h), 0], [np.sin(h), np.cos(h), 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, np.cos(p), -np.sin(p)], [0, np.sin(p), np.cos(p)]])
    Ry = np.array([[np.cos(r), 0, np.sin(r)], [0, 1, 0], [-np.sin(r), 0, np.cos(r)]])
    return Rz @ Rx @ Ry


def load_mesh_file(path) -> Dict[str, Optional[np.ndarray]]:
    p = str(path).lower()
    if p.endswith(".ply"):
        return read_ply(path)
    if p.endswith(".obj"):
        return read_obj(path)
    raise ValueError(f"unsupported mesh format: {path} (PLY and OBJ are supported)")


def load_rigid_object(obj) -> Dict[str, np.ndarray]:
    """RigidObject (reference src/megapose/datasets/object_dataset.py:35-137 duck type) -> engine mesh dict.
    Returns float32 arrays: vertices (metres), normals, colors, int32 faces, plus 'points' = the metre-scaled
    vertices WITHOUT the renderer-only ypr offset (what MeshDataBase hands to the pose math)."""
    raw = load_mesh_file(Path(obj.mesh_path))
    scale = float(obj.scale)
    pts = raw["vertices"] * scale
    faces = raw["faces"].astype(np.int32)
    normals = raw["normals"] if raw["normals"] is not None else vertex_normals(raw["vertices"], faces)
    colors = raw["colors"] if raw["colors"] is not None else np.ones_like(pts)
    ypr = tuple(getattr(obj, "ypr_offset_deg", (0.0, 0.0, 0.0)))
    rv, rn = pts, normals
    if any(abs(a) > 0 for a in ypr):
        R = _ypr_matrix(ypr)
        rv = pts @ R.T
        rn = normals @ R.T
    return {
        "vertices": rv.astype(np.float32),
        "normals": rn.astype(np.float32),
        "colors": colors.astype(np.float32),
        "faces": faces,
        "points": pts.astype(np.float32),
    }
