"""ctypes wrapper of oracle/raster.c + an oracle object implementing the Panda3dBatchRenderer.render contract
on CPU tensors.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "liboracle.so"


class _Lights(C.Structure):
    _fields_ = [("ambient", C.c_float * 3), ("n_point", C.c_int32), ("dir", (C.c_float * 3) * 8), ("color", (C.c_float * 3) * 8),
                ("offset", (C.c_float * 3) * 8)]


_lib = None


def build() -> None:
    subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if not _LIB.is_file():
            build()
        _lib = C.CDLL(str(_LIB))
        _lib.oracle_raster_render.restype = None
    return _lib


POINT_DIRS = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]  # panda3d_scene_renderer.py:109-118


def lights_struct(ambient=(1.0, 1.0, 1.0), point_dirs=(), point_colors=(), point_offsets=None):
    """point light i sits at dir_i * 10 * bounding radius + offset_i (object frame)"""
    L = _Lights()
    L.ambient[:] = ambient
    L.n_point = len(point_dirs)
    for i, (d, c) in enumerate(zip(point_dirs, point_colors)):
        L.dir[i][:] = d
        L.color[i][:] = c
        L.offset[i][:] = point_offsets[i] if point_offsets is not None else (0.0, 0.0, 0.0)
    return L


FLAG_NORMALS, FLAG_DEPTH, FLAG_GL_EYE, FLAG_MSAA4 = 1, 2, 4, 16


def mesh_radius(vertices: np.ndarray) -> float:
    v = vertices.astype(np.float32)
    lo, hi = v.min(0), v.max(0)
    c = (np.float32(0.5) * (lo + hi)).astype(np.float32)
    d = (v - c).astype(np.float32)
    s = np.zeros(len(v), np.float32)
    for k in range(3):
        s = (s + d[:, k] * d[:, k]).astype(np.float32)
    return float(np.sqrt(s.max(), dtype=np.float32))


def render(mesh: Dict[str, np.ndarray], TCO: np.ndarray, K: np.ndarray, h: int, w: int, flags: int, lights=None):
    """mesh: float32 vertices/normals/colors + int32 faces.  Returns rgb [n,h,w,3], normals [n,h,w,3], depth [n,h,w]."""
    v = np.ascontiguousarray(mesh["vertices"], np.float32)
    n = np.ascontiguousarray(mesh["normals"], np.float32)
    c = np.ascontiguousarray(mesh["colors"], np.float32)
    f = np.ascontiguousarray(mesh["faces"], np.int32)
    T = np.ascontiguousarray(TCO, np.float32).reshape(-1, 16)
    Kk = np.ascontiguousarray(K, np.float32).reshape(-1, 9)
    nv = T.shape[0]
    rgb = np.zeros((nv, h, w, 3), np.float32)
    nrm = np.zeros((nv, h, w, 3), np.float32)
    dep = np.zeros((nv, h, w), np.float32)
    L = lights if lights is not None else lights_struct()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    uv_p, tex_p, tw, th, nl = C.c_void_p(None), C.c_void_p(None), 0, 0, 0
    if mesh.get("uvs") is not None and mesh.get("texture_mips") is not None:
        uv = np.ascontiguousarray(mesh["uvs"], np.float32)
        mips = mesh["texture_mips"]
        flat = np.ascontiguousarray(np.concatenate([lv.reshape(-1) for lv in mips]).astype(np.uint32))
        th, tw = mips[0].shape[:2]
        uv_p, tex_p, nl = p(uv), p(flat), len(mips)
    lib().oracle_raster_render(p(v), p(n), p(c), p(f), C.c_int(v.shape[0]), C.c_int(f.shape[0]), C.c_float(mesh_radius(v)), p(T),
                               p(Kk), C.c_int(nv), C.c_int(h), C.c_int(w), C.c_uint32(flags), C.byref(L), p(rgb), p(nrm), p(dep),
                               uv_p, tex_p, C.c_int(tw), C.c_int(th), C.c_int(nl))
    return rgb, nrm, dep


class OracleBatchRenderer:
    """CPU object with the Panda3dBatchRenderer.render signature
    (/root/reference/src/megapose/panda3d_renderer/panda3d_batch_renderer.py:217-282)."""

    def __init__(self, meshes_by_label: Dict[str, Dict[str, np.ndarray]], gl_eye: bool = False, msaa: int = 4):
        assert msaa in (1, 4)
        self.meshes = meshes_by_label
        self.gl_eye = gl_eye
        self.msaa = msaa  # 4 = the reference's configuration (panda3d_scene_renderer.py:73-74)
        self.n_calls = 0
        self.n_views = 0

    def render(self, labels: List[str], TCO: torch.Tensor, K: torch.Tensor, light_datas=None, resolution=(240, 320),
               render_depth: bool = False, render_mask: bool = False, render_normals: bool = False):
        from types import SimpleNamespace

        if render_mask:
            raise NotImplementedError
        bsz = TCO.shape[0]
        assert TCO.shape == (bsz, 4, 4) and K.shape == (bsz, 3, 3)
        h, w = resolution
        T = TCO.detach().cpu().float().numpy()
        Kn = K.detach().cpu().float().numpy()
        rgbs = np.zeros((bsz, h, w, 3), np.float32)
        nrms = np.zeros((bsz, h, w, 3), np.float32)
        deps = np.zeros((bsz, h, w), np.float32)
        flags = (1 if render_normals else 0) | (2 if render_depth else 0) | (4 if self.gl_eye else 0) | (16 if self.msaa == 4 else 0)
        for i, lab in enumerate(labels):
            L = lights_from_datas(light_datas[i]) if light_datas is not None else lights_struct()
            r, n, d = render(self.meshes[lab], T[i : i + 1], Kn[i : i + 1], h, w, flags, L)
            rgbs[i], nrms[i], deps[i] = r[0], n[0], d[0]
        self.n_calls += 1
        self.n_views += bsz
        return SimpleNamespace(
            rgbs=torch.from_numpy(rgbs).permute(0, 3, 1, 2).contiguous(),
            normals=torch.from_numpy(nrms).permute(0, 3, 1, 2).contiguous() if render_normals else None,
            depths=torch.from_numpy(deps).unsqueeze(1).contiguous() if render_depth else None,
        )

    def stop(self):
        pass


class _ProbeBounds:
    def __init__(self, radius):
        self.radius = radius

    def getRadius(self):
        return self.radius

    get_radius = getRadius


class _ProbeRoot:
    """stands in for the panda3d root NodePath a positioning_function receives: only getBounds().radius is answered"""

    def __init__(self, radius):
        self._b = _ProbeBounds(radius)

    def getBounds(self):
        return self._b

    get_bounds = getBounds


class _ProbeLight:
    def __init__(self):
        self.pos = None

    def setPos(self, *a):
        self.pos = tuple(float(v) for v in (a[0] if len(a) == 1 else a))

    set_pos = setPos


def probe_light_position(fn):
    """positioning_function(root_node, light_node) (panda3d_scene_renderer.py:121-133, types.py:104-114) -> (a, b) with
    position = a * bounding_radius + b, found by calling it at radius 1, 2 and 4 with recording stand-ins."""
    pts = []
    for r in (1.0, 2.0, 4.0):
        light = _ProbeLight()
        fn(_ProbeRoot(r), light)
        if light.pos is None:
            raise NotImplementedError("positioning_function did not call setPos on the light node")
        pts.append(np.asarray(light.pos, np.float64))
    a = pts[1] - pts[0]
    b = pts[0] - a
    if np.abs(a * 4.0 + b - pts[2]).max() > 1e-9 * (1.0 + np.abs(pts[2]).max()):
        raise NotImplementedError("positioning_function is not affine in the bounding radius")
    return a, b


def lights_from_datas(datas) -> "_Lights":
    """List[Panda3dLightData-like] -> light struct.  A point light's position comes from its positioning_function (probed, see
    probe_light_position) or from an explicit `direction`; without either it is one of the 6 axis lights of make_scene_lights."""
    amb = np.zeros(3, np.float32)
    dirs, cols, offs = [], [], []
    k = 0
    for ld in datas:
        if ld.light_type == "ambient":
            amb += np.asarray(ld.color[:3], np.float32)
        elif ld.light_type == "point":
            fn = getattr(ld, "positioning_function", None)
            d = getattr(ld, "direction", None)
            if fn is not None:
                a, b = probe_light_position(fn)
                dirs.append(tuple(float(v) / 10.0 for v in a))
                offs.append(tuple(float(v) for v in b))
            else:
                dirs.append(tuple(float(v) for v in d) if d is not None else POINT_DIRS[k % 6])
                offs.append((0.0, 0.0, 0.0))
            cols.append(tuple(ld.color[:3]))
            k += 1
        else:
            raise NotImplementedError(ld.light_type)
    return lights_struct(tuple(float(a) for a in amb), dirs, cols, offs)
