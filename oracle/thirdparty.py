"""CPU restatements of the third-party calls on the hot path that are NOT vendored
under /root/reference.  TEST INFRASTRUCTURE ONLY.

Each function names the package + pinned version (conda/environment_full.yaml,
docker/Dockerfile.megapose) and the reference call site it serves.  None of these
can be checked against the real packages here ("parity unpinned"); they are
restated from the packages' published algorithms.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np
import torch


# --------------------------------------------------------------------------- #
# torchvision==0.12.0  torchvision.ops.roi_align  (aligned=False, spatial_scale=1)
# reference call sites: src/megapose/lib3d/cropping.py:125-127, :129, :137-139
# algorithm: torchvision/csrc/ops/cpu/roi_align_kernel.cpp (roi_align_forward_kernel_impl
#            + pre_calc_for_bilinear_interpolate)
# --------------------------------------------------------------------------- #
def roi_align(
    input: torch.Tensor,
    boxes: torch.Tensor,
    output_size: Tuple[int, int],
    spatial_scale: float = 1.0,
    sampling_ratio: int = -1,
    aligned: bool = False,
) -> torch.Tensor:
    """input [N,C,H,W] fp32, boxes [K,5] = (batch_idx, x1, y1, x2, y2) -> [K,C,ph,pw]."""
    assert input.dim() == 4 and boxes.dim() == 2 and boxes.shape[1] == 5
    assert sampling_ratio > 0, "hot path always passes sampling_ratio=4"
    N, C, H, W = input.shape
    K = boxes.shape[0]
    ph, pw = int(output_size[0]), int(output_size[1])
    dt = torch.float32
    inp = input.to(dt)
    b = boxes.to(dt)
    g = int(sampling_ratio)
    offset = 0.5 if aligned else 0.0
    out = torch.zeros(K, C, ph, pw, dtype=dt)
    if K == 0:
        return out
    bidx = b[:, 0].long()
    x1 = b[:, 1] * spatial_scale - offset
    y1 = b[:, 2] * spatial_scale - offset
    x2 = b[:, 3] * spatial_scale - offset
    y2 = b[:, 4] * spatial_scale - offset
    rw = x2 - x1
    rh = y2 - y1
    if not aligned:
        rw = torch.clamp(rw, min=1.0)
        rh = torch.clamp(rh, min=1.0)
    bin_h = rh / ph
    bin_w = rw / pw
    # sample coordinates, fp32 like the C++ kernel:  y = start + ph*bin + (iy+.5)*bin/grid
    iy = torch.arange(g, dtype=dt)
    py = torch.arange(ph, dtype=dt)
    px = torch.arange(pw, dtype=dt)
    # [K, ph, g]
    ys = y1[:, None, None] + py[None, :, None] * bin_h[:, None, None] + (
        (iy[None, None, :] + 0.5) * bin_h[:, None, None] / g
    )
    xs = x1[:, None, None] + px[None, :, None] * bin_w[:, None, None] + (
        (iy[None, None, :] + 0.5) * bin_w[:, None, None] / g
    )

    def prep(c, size):
        invalid = (c < -1.0) | (c > size)
        c = torch.where(c <= 0, torch.zeros_like(c), c)
        lo = c.floor().long()  # (int) cast of a non-negative float
        at_edge = lo >= size - 1
        lo = torch.where(at_edge, torch.full_like(lo, size - 1), lo)
        hi = torch.where(at_edge, lo, lo + 1)
        c = torch.where(at_edge, lo.to(dt), c)
        l = c - lo.to(dt)
        h = 1.0 - l
        return invalid, lo, hi, l, h

    inv_y, ylo, yhi, ly, hy = prep(ys, H)  # [K, ph, g]
    inv_x, xlo, xhi, lx, hx = prep(xs, W)  # [K, pw, g]
    count = float(max(g * g, 1))
    # accumulate in the kernel's order: iy outer, ix inner, sequential fp32 adds
    for k in range(K):
        img = inp[bidx[k]]  # [C,H,W]
        acc = torch.zeros(C, ph, pw, dtype=dt)
        for a in range(g):
            for c_ in range(g):
                yl = ylo[k, :, a]
                yh = yhi[k, :, a]
                xl = xlo[k, :, c_]
                xh = xhi[k, :, c_]
                w1 = hy[k, :, a][:, None] * hx[k, :, c_][None, :]
                w2 = hy[k, :, a][:, None] * lx[k, :, c_][None, :]
                w3 = ly[k, :, a][:, None] * hx[k, :, c_][None, :]
                w4 = ly[k, :, a][:, None] * lx[k, :, c_][None, :]
                bad = inv_y[k, :, a][:, None] | inv_x[k, :, c_][None, :]
                v1 = img[:, yl][:, :, xl]
                v2 = img[:, yl][:, :, xh]
                v3 = img[:, yh][:, :, xl]
                v4 = img[:, yh][:, :, xh]
                val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
                val = torch.where(bad[None], torch.zeros_like(val), val)
                acc = acc + val
        out[k] = acc / count
    return out


# --------------------------------------------------------------------------- #
# roma (unpinned)  roma.unitquat_to_rotmat  -- xyzw unit quaternion -> rotation matrix
# reference call site: src/megapose/utils/transform_utils.py:49
# --------------------------------------------------------------------------- #
def unitquat_to_rotmat(quat: torch.Tensor) -> torch.Tensor:
    x, y, z, w = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
            2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return R.reshape(*quat.shape[:-1], 3, 3)


# --------------------------------------------------------------------------- #
# pinocchio (unpinned)  pin.SE3 / pin.Quaternion -- only what lib3d/transform.py:27-119 uses
# --------------------------------------------------------------------------- #
class Quaternion:
    """pin.Quaternion(w, x, y, z) or pin.Quaternion(R[3,3])."""

    def __init__(self, *args):
        if len(args) == 4:
            self.w, self.x, self.y, self.z = [float(a) for a in args]
        elif len(args) == 1:
            R = np.asarray(args[0], dtype=np.float64)
            tr = np.trace(R)
            if tr > 0:
                s = math.sqrt(tr + 1.0) * 2
                self.w = 0.25 * s
                self.x = (R[2, 1] - R[1, 2]) / s
                self.y = (R[0, 2] - R[2, 0]) / s
                self.z = (R[1, 0] - R[0, 1]) / s
            else:
                i = int(np.argmax(np.diag(R)))
                j, k = (i + 1) % 3, (i + 2) % 3
                s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
                q = [0.0, 0.0, 0.0]
                q[i] = 0.25 * s
                q[j] = (R[j, i] + R[i, j]) / s
                q[k] = (R[k, i] + R[i, k]) / s
                self.w = (R[k, j] - R[j, k]) / s
                self.x, self.y, self.z = q
        else:
            raise ValueError

    def normalize(self):
        n = math.sqrt(self.w**2 + self.x**2 + self.y**2 + self.z**2)
        self.w, self.x, self.y, self.z = self.w / n, self.x / n, self.y / n, self.z / n
        return self

    def matrix(self) -> np.ndarray:
        q = torch.tensor([self.x, self.y, self.z, self.w], dtype=torch.float64)
        return unitquat_to_rotmat(q).numpy()

    def coeffs(self) -> np.ndarray:
        return np.array([self.x, self.y, self.z, self.w])


class SE3:
    def __init__(self, R, t):
        self.rotation = np.asarray(R, dtype=np.float64).reshape(3, 3).copy()
        self.translation = np.asarray(t, dtype=np.float64).reshape(3).copy()

    @property
    def homogeneous(self) -> np.ndarray:
        T = np.eye(4)
        T[:3, :3] = self.rotation
        T[:3, 3] = self.translation
        return T

    def __mul__(self, other: "SE3") -> "SE3":
        return SE3(self.rotation @ other.rotation, self.rotation @ other.translation + self.translation)

    def inverse(self) -> "SE3":
        Rt = self.rotation.T
        return SE3(Rt, -Rt @ self.translation)

    def __str__(self):
        return str(self.homogeneous)


# --------------------------------------------------------------------------- #
# panda3d NodePath.lookAt/setPos/getMat as used by
# src/megapose/lib3d/multiview.py:31-92 (_get_views_TCO_pos_sphere).
# Closed form: Panda look_at() = forward-exact:
#   y = normalize(target - pos), x = normalize(y x up), z = x x y   (x right, y fwd, z up)
# --------------------------------------------------------------------------- #
def _look_at_R(pos: np.ndarray, target: np.ndarray, up: np.ndarray) -> np.ndarray:
    y = target - pos
    y = y / np.linalg.norm(y)
    x = np.cross(y, up)
    x = x / np.linalg.norm(x)
    z = np.cross(x, y)
    return np.stack([x, y, z], axis=1)  # columns = node axes in world


_TCCGL3 = np.array([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])


def get_views_TCO_pos_sphere(TCO, tCR, cam_positions_wrt_cam0) -> List[np.ndarray]:
    """Drop-in for megapose.lib3d.multiview._get_views_TCO_pos_sphere: returns the
    list of TC0_CV (4x4, float64) for each requested camera offset."""
    TCO = np.array(np.asarray(TCO).tolist(), dtype=np.float64)
    tCR = np.array(np.asarray(tCR).tolist(), dtype=np.float64)
    Rco, tco = TCO[:3, :3], TCO[:3, 3]
    TOC = np.eye(4)
    TOC[:3, :3] = Rco.T
    TOC[:3, 3] = -Rco.T @ tco
    if not np.isfinite(TOC).all():  # multiview.py:44-46
        TOC = np.eye(4)
        tCR = np.zeros(3)
        TCO = np.eye(4)
    p0 = TOC[:3, 3]
    up = -TOC[:3, 1]
    ref = TOC[:3, :3] @ tCR + p0
    radius = np.linalg.norm(tCR)
    offsets = np.asarray(cam_positions_wrt_cam0, dtype=np.float64) * radius
    L = _look_at_R(p0, ref, up)
    out = []
    for o in offsets:
        pn = p0 + L @ o
        Rn = _look_at_R(pn, ref, up)
        TCV_O = np.eye(4)
        TCV_O[:3, :3] = _TCCGL3 @ Rn.T
        TCV_O[:3, 3] = -_TCCGL3 @ Rn.T @ pn
        # the reference returns TC0_CV with TCV_O = inv(TC0_CV) @ TCO  (multiview.py:219)
        TC0_CV = TCO @ np.linalg.inv(TCV_O)
        out.append(TC0_CV)
    return out


# --------------------------------------------------------------------------- #
# trimesh (unpinned)  trimesh.load(path, process=False, maintain_order=True)
# reference call site: src/megapose/lib3d/rigid_mesh_database.py:64-70.
# Only .vertices / .faces are consumed (vertex ORDER matters: RandomState(0).choice).
# --------------------------------------------------------------------------- #
class SimpleMesh:
    def __init__(self, vertices, faces, normals=None, colors=None):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)
        self.vertex_normals = normals
        self.vertex_colors = colors


def load_ply(path) -> SimpleMesh:
    """Minimal PLY reader (ascii or binary_little_endian; vertex x,y,z[,nx,ny,nz][,red,green,blue]
    and face vertex_indices lists)."""
    with open(path, "rb") as f:
        header = []
        while True:
            line = f.readline().decode("ascii").strip()
            header.append(line)
            if line == "end_header":
                break
        fmt = [l for l in header if l.startswith("format")][0].split()[1]
        elems = []
        cur = None
        for l in header:
            tok = l.split()
            if tok[0] == "element":
                cur = {"name": tok[1], "count": int(tok[2]), "props": []}
                elems.append(cur)
            elif tok[0] == "property":
                cur["props"].append(tok[1:])
        np_t = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1",
                "char": "i1", "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
                "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4"}
        verts = faces = None
        vprops = {}
        for el in elems:
            if el["name"] == "vertex":
                names = [p[-1] for p in el["props"]]
                if fmt == "ascii":
                    arr = np.loadtxt([f.readline().decode() for _ in range(el["count"])], ndmin=2)
                    vprops = {n: arr[:, i] for i, n in enumerate(names)}
                else:
                    dt = np.dtype([(p[-1], "<" + np_t[p[0]]) for p in el["props"]])
                    arr = np.frombuffer(f.read(dt.itemsize * el["count"]), dtype=dt)
                    vprops = {n: arr[n] for n in names}
                verts = np.stack([vprops["x"], vprops["y"], vprops["z"]], axis=1).astype(np.float64)
            elif el["name"] == "face":
                fl = []
                if fmt == "ascii":
                    for _ in range(el["count"]):
                        tok = f.readline().decode().split()
                        n = int(tok[0])
                        fl.append([int(t) for t in tok[1 : 1 + n]])
                else:
                    p = el["props"][0]  # list <count type> <index type> vertex_indices
                    ct, it = np.dtype("<" + np_t[p[1]]), np.dtype("<" + np_t[p[2]])
                    for _ in range(el["count"]):
                        n = int(np.frombuffer(f.read(ct.itemsize), dtype=ct)[0])
                        fl.append(np.frombuffer(f.read(it.itemsize * n), dtype=it).tolist())
                tris = []
                for poly in fl:
                    for i in range(1, len(poly) - 1):
                        tris.append([poly[0], poly[i], poly[i + 1]])
                faces = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
            else:
                raise ValueError(f"unsupported PLY element {el['name']}")
    normals = colors = None
    if "nx" in vprops:
        normals = np.stack([vprops["nx"], vprops["ny"], vprops["nz"]], axis=1).astype(np.float64)
    if "red" in vprops:
        colors = np.stack([vprops["red"], vprops["green"], vprops["blue"]], axis=1).astype(np.uint8)
    return SimpleMesh(verts, faces if faces is not None else np.zeros((0, 3), np.int64), normals, colors)


def trimesh_load(path, *args, **kwargs) -> SimpleMesh:
    path = str(path)
    if path.endswith(".ply"):
        return load_ply(path)
    raise ValueError(f"oracle trimesh stand-in only reads .ply, got {path}")
