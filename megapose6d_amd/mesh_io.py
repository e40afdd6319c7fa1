"""Mesh ingestion for the engine: RigidObject -> float32 arrays for mp_mesh_db_create.

Replaces, for the hot path, `trimesh.load(..., process=False, maintain_order=True)`
(reference src/megapose/lib3d/rigid_mesh_database.py:62-73) and Panda3D's assimp loader
(src/megapose/panda3d_renderer/panda3d_scene_renderer.py:192-207: scale =
scaling_factor_mesh_units_to_meters * scaling_factor, HPR offset applied in the renderer only).
Vertex ORDER is preserved (the deterministic point sampling indexes it, lib3d/mesh_ops.py:77-87).

Supported: PLY (ascii / binary_little_endian; x y z [nx ny nz] [red green blue [alpha]]; polygon faces
are fan-triangulated) and OBJ (v / vt / vn / f with optional per-vertex colours `v x y z r g b`).
UV textures: PLY `comment TextureFile <png>` with per-vertex `s t` / `u v` /
`texture_u texture_v` or a per-face `texcoord` list (the BOP / YCB-V convention); OBJ `vt` + `mtllib` ->
`map_Kd`.  UVs are kept PER CORNER ([T,3,2]) so the vertex order is untouched; the image is decoded with
PIL, flipped so that v grows with the row index, and expanded into an RGBA8 mip chain (`build_mip_chain`)
-- the rasteriser's input.  Meshes without vertex colours render white (times the texture, if any), as
Panda3D does.
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


def _fan_uv(poly_uvs) -> np.ndarray:
    """per-polygon corner uvs [(n,2)...] -> per-triangle corner uvs [T,3,2], same fan as `_fan`"""
    out = []
    for uv in poly_uvs:
        for i in range(1, len(uv) - 1):
            out.append((uv[0], uv[i], uv[i + 1]))
    return np.asarray(out, dtype=np.float64).reshape(-1, 3, 2)


def build_mip_chain(image_rgb: np.ndarray):
    """uint8 [H,W,3|4] (row 0 = v 0) -> list of uint32 [h_l,w_l] RGBA8 levels (R in the low byte), level l of size
    max(1,H>>l) x max(1,W>>l), each texel the rounded mean of its 2x2 parents (edge-clamped for odd sizes)."""
    img = np.asarray(image_rgb)
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] in (3, 4)
    cur = img[..., :3].astype(np.uint32)
    H, W = cur.shape[:2]
    levels = []
    l = 0
    while True:
        levels.append((cur[..., 0] | (cur[..., 1] << 8) | (cur[..., 2] << 16) | np.uint32(0xFF000000)).astype(np.uint32))
        if cur.shape[0] == 1 and cur.shape[1] == 1:
            break
        l += 1
        h2, w2 = max(1, H >> l), max(1, W >> l)
        ys = np.minimum(2 * np.arange(h2)[:, None] + np.array([0, 1])[None, :], cur.shape[0] - 1)  # [h2,2]
        xs = np.minimum(2 * np.arange(w2)[:, None] + np.array([0, 1])[None, :], cur.shape[1] - 1)
        acc = np.zeros((h2, w2, 3), np.uint32)
        for a in range(2):
            for b in range(2):
                acc += cur[ys[:, a]][:, xs[:, b]]
        cur = (acc + 2) >> 2
    return levels


def load_texture(path) -> np.ndarray:
    """PNG/JPEG -> uint8 [H,W,3], flipped vertically (image row 0 is the TOP of the picture = v 1 in the OBJ/PLY convention)."""
    from PIL import Image

    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(arr[::-1])


def read_ply(path) -> Dict[str, Optional[np.ndarray]]:
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt = None
        elems = []
        texture_file = None
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "comment":
                if len(tok) >= 3 and tok[1].lower() == "texturefile":
                    texture_file = line.decode("ascii", "replace").split(None, 2)[2].strip()
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
        face_uvs = None
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
                polys, puvs = [], []
                for _ in range(el["count"]):  # properties in declaration order: index list, optional texcoord list, scalars
                    rec = {}
                    tok = f.readline().split() if fmt == "ascii" else None
                    pos = 0
                    for pr in el["props"]:
                        if pr[0] == "list":
                            if fmt == "ascii":
                                n = int(tok[pos])
                                vals = tok[pos + 1 : pos + 1 + n]
                                pos += 1 + n
                            else:
                                ct, it = np.dtype("<" + _PLY_T[pr[1]]), np.dtype("<" + _PLY_T[pr[2]])
                                n = int(np.frombuffer(f.read(ct.itemsize), dtype=ct)[0])
                                vals = np.frombuffer(f.read(it.itemsize * n), dtype=it)
                            rec[pr[-1]] = vals
                        else:
                            if fmt == "ascii":
                                pos += 1
                            else:
                                f.read(np.dtype(_PLY_T[pr[0]]).itemsize)
                    idx = rec.get("vertex_indices", rec.get("vertex_index"))
                    if idx is None:
                        raise ValueError(f"{path}: face element without vertex_indices")
                    polys.append([int(t) for t in idx])
                    if "texcoord" in rec:
                        puvs.append(np.asarray(rec["texcoord"], dtype=np.float64).reshape(-1, 2))
                faces = _fan(polys)
                if puvs and len(puvs) == len(polys):
                    face_uvs = _fan_uv(puvs)
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
    for a, b in (("s", "t"), ("u", "v"), ("texture_u", "texture_v")):
        if face_uvs is None and a in vp and b in vp and len(faces):
            face_uvs = np.stack([vp[a], vp[b]], axis=1).astype(np.float64)[faces]
    tex = None
    if texture_file is not None and face_uvs is not None:
        tex = Path(path).parent / texture_file
    return {"vertices": verts, "faces": faces, "normals": normals, "colors": colors, "uvs": face_uvs if tex is not None else None,
            "texture_path": tex}


def read_obj(path) -> Dict[str, Optional[np.ndarray]]:
    verts, cols, polys, vts, puv_idx = [], [], [], [], []
    mtllib = None
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append([float(t) for t in tok[1:4]])
                if len(tok) >= 7:
                    cols.append([float(t) for t in tok[4:7]])
            elif tok[0] == "vt":
                vts.append([float(tok[1]), float(tok[2]) if len(tok) > 2 else 0.0])
            elif tok[0] == "f":
                corners = [t.split("/") for t in tok[1:]]
                nv, nt = len(verts), len(vts)
                polys.append([(int(c[0]) - 1) if int(c[0]) > 0 else nv + int(c[0]) for c in corners])
                if all(len(c) > 1 and c[1] for c in corners):
                    puv_idx.append([(int(c[1]) - 1) if int(c[1]) > 0 else nt + int(c[1]) for c in corners])
                else:
                    puv_idx.append(None)
            elif tok[0] == "mtllib":
                mtllib = line.split(None, 1)[1].strip()
    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    c = np.asarray(cols, dtype=np.float64) if len(cols) == len(verts) and cols else None
    uvs, tex = None, None
    if vts and polys and all(p is not None for p in puv_idx):
        vt = np.asarray(vts, dtype=np.float64)
        uvs = _fan_uv([vt[p] for p in puv_idx])
        if mtllib is not None and (Path(path).parent / mtllib).is_file():
            for line in (Path(path).parent / mtllib).read_text().splitlines():
                tok = line.split()
                if tok and tok[0] == "map_Kd":  # first diffuse map (one material per object on this path)
                    tex = Path(path).parent / tok[-1]
                    break
    return {"vertices": v, "faces": _fan(polys), "normals": None, "colors": c, "uvs": uvs if tex is not None else None,
            "texture_path": tex}


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
    out = {
        "vertices": rv.astype(np.float32),
        "normals": rn.astype(np.float32),
        "colors": colors.astype(np.float32),
        "faces": faces,
        "points": pts.astype(np.float32),
    }
    tex = getattr(obj, "texture_path", None) or raw.get("texture_path")
    if raw.get("uvs") is not None and tex is not None:
        if not Path(tex).is_file():
            raise FileNotFoundError(f"{obj.mesh_path}: texture {tex} not found")
        out["uvs"] = raw["uvs"].astype(np.float32)
        out["texture_mips"] = build_mip_chain(load_texture(tex))
    return out
