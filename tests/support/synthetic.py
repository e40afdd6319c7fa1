"""TEST / BENCH SUPPORT (not product code): seeded synthetic assets (meshes, weights, scenes) -- there is no network for
datasets or checkpoints.

Shapes follow SURVEY.md section 8d: procedurally generated closed lathe surfaces (>= 2000 vertices, per-vertex
colours, mm units), K from the reference README.md:226, seeded random weights laid out exactly like the
reference checkpoints (state_dict keys of src/megapose/models/torchvision_resnet.py / wide_resnet.py /
pose_rigid.py:122-130) so that `load_state_dict(strict=True)` works on the reference modules.
"""
from __future__ import annotations

import math
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from megapose6d_amd.object_dataset import RigidObject, RigidObjectDataset  # noqa: F401

# K of the barbecue-sauce example (reference README.md:226), 640x480
K_EXAMPLE = np.array([[605.9547119140625, 0.0, 319.029052734375], [0.0, 605.006591796875, 249.67617797851562], [0.0, 0.0, 1.0]])


def make_lathe_mesh(seed: int = 0, n_theta: int = 72, n_z: int = 70, height_mm: float = 160.0, radius_mm: float = 40.0):
    """Closed bottle-like surface of revolution with angular bumps (so it has no symmetry).
    Returns vertices [V,3] float64 in mm, faces [T,3] int32, colors uint8 [V,3]."""
    rng = np.random.RandomState(seed)
    zs = np.linspace(-0.5, 0.5, n_z)
    prof = 1.0 + 0.25 * np.sin(2 * np.pi * (zs * rng.uniform(0.8, 1.6) + rng.uniform()))
    prof *= np.clip(1.4 - 1.2 * np.abs(zs) ** 2 * rng.uniform(1.0, 3.0), 0.35, None)
    th = np.linspace(0, 2 * np.pi, n_theta, endpoint=False)
    k1, k2 = rng.randint(2, 5), rng.randint(1, 4)
    a1, a2 = rng.uniform(0.05, 0.15), rng.uniform(0.03, 0.1)
    ph1, ph2 = rng.uniform(0, 2 * np.pi, 2)
    r = radius_mm * prof[:, None] * (1 + a1 * np.cos(k1 * th[None] + ph1) + a2 * np.sin(k2 * th[None] + ph2 + 3 * zs[:, None]))
    x = r * np.cos(th[None])
    y = r * np.sin(th[None])
    z = np.broadcast_to(zs[:, None] * height_mm, r.shape)
    verts = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    bot, top = len(verts), len(verts) + 1
    verts = np.concatenate([verts, [[0, 0, zs[0] * height_mm]], [[0, 0, zs[-1] * height_mm]]], axis=0)
    faces = []
    for i in range(n_z - 1):
        for j in range(n_theta):
            a = i * n_theta + j
            b = i * n_theta + (j + 1) % n_theta
            c = (i + 1) * n_theta + j
            d = (i + 1) * n_theta + (j + 1) % n_theta
            faces.append((a, b, d))
            faces.append((a, d, c))
    for j in range(n_theta):
        faces.append((bot, (j + 1) % n_theta, j))
        a = (n_z - 1) * n_theta
        faces.append((top, a + j, a + (j + 1) % n_theta))
    faces = np.asarray(faces, dtype=np.int32)
    # smooth random colour field
    f = rng.uniform(0.5, 2.5, size=(3, 3))
    p = rng.uniform(0, 2 * np.pi, size=(3, 3))
    vn = verts / np.array([radius_mm, radius_mm, height_mm / 2])
    col = np.stack([0.5 + 0.5 * np.sin(f[c, 0] * vn[:, 0] * 3 + p[c, 0]) * np.cos(f[c, 1] * vn[:, 1] * 3 + p[c, 1])
                    * np.sin(f[c, 2] * vn[:, 2] * 3 + p[c, 2]) for c in range(3)], axis=1)
    col = np.clip(col * 0.8 + 0.1 + rng.uniform(-0.05, 0.05, size=col.shape), 0, 1)
    colors = np.round(col * 255).astype(np.uint8)
    return verts, faces, colors


def write_ply(path, verts: np.ndarray, faces: np.ndarray, colors_u8: Optional[np.ndarray] = None,
              normals: Optional[np.ndarray] = None) -> None:
    """binary_little_endian PLY with float x,y,z [nx,ny,nz] [uchar red,green,blue]."""
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if normals is not None:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
    if colors_u8 is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    arr = np.zeros(len(verts), dtype=np.dtype(fields))
    arr["x"], arr["y"], arr["z"] = verts[:, 0], verts[:, 1], verts[:, 2]
    if normals is not None:
        arr["nx"], arr["ny"], arr["nz"] = normals[:, 0], normals[:, 1], normals[:, 2]
    if colors_u8 is not None:
        arr["red"], arr["green"], arr["blue"] = colors_u8[:, 0], colors_u8[:, 1], colors_u8[:, 2]
    names = {"<f4": "float", "u1": "uchar"}
    hdr = ["ply", "format binary_little_endian 1.0", f"element vertex {len(verts)}"]
    hdr += [f"property {names[t]} {n}" for n, t in fields]
    hdr += [f"element face {len(faces)}", "property list uchar int vertex_indices", "end_header"]
    fa = np.zeros(len(faces), dtype=np.dtype([("n", "u1"), ("i", "<i4", (3,))]))
    fa["n"] = 3
    fa["i"] = faces
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        f.write(arr.tobytes())
        f.write(fa.tobytes())


def make_object_dataset(out_dir, n_objects: int = 1, seed: int = 0, n_theta: int = 72, n_z: int = 70) -> RigidObjectDataset:
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    objs = []
    for i in range(n_objects):
        v, f, c = make_lathe_mesh(seed + i, n_theta=n_theta, n_z=n_z, height_mm=120.0 + 20.0 * (i % 4), radius_mm=30.0 + 4.0 * (i % 3))
        p = out_dir / f"obj_{i:06d}.ply"
        write_ply(p, v, f, c)
        objs.append(RigidObject(label=f"obj_{i:06d}", mesh_path=p, mesh_units="mm"))
    return RigidObjectDataset(objs)


def make_texture_image(seed: int = 0, h: int = 128, w: int = 256) -> np.ndarray:
    """procedural uint8 [h,w,3] picture (row 0 = top): colour checker + gradients + fine noise (so mip levels differ)"""
    rng = np.random.RandomState(seed)
    ys, xs = np.mgrid[0:h, 0:w]
    chk = ((xs // 16 + ys // 16) % 2).astype(np.float64)
    base = np.stack([0.25 + 0.6 * chk, 0.2 + 0.7 * xs / (w - 1), 0.9 - 0.7 * ys / (h - 1)], axis=-1)
    base += rng.uniform(-0.1, 0.1, size=base.shape)
    return np.round(np.clip(base, 0, 1) * 255).astype(np.uint8)


def lathe_corner_uvs(verts: np.ndarray, faces: np.ndarray, n_theta: int, n_z: int, radius_mm: float) -> np.ndarray:
    """cylindrical per-corner uvs [T,3,2] for make_lathe_mesh (seam-free: the wrap-around column gets u = 1)"""
    uv = np.zeros((len(faces), 3, 2))
    n_side = 2 * (n_z - 1) * n_theta
    for t, tri in enumerate(faces):
        if t < n_side:
            j0 = min(int(c) % n_theta for c in tri)
            for k, c in enumerate(tri):
                i, j = divmod(int(c), n_theta)
                if j == 0 and j0 == 0 and any(int(cc) % n_theta == n_theta - 1 for cc in tri):
                    j = n_theta
                uv[t, k] = (j / n_theta, i / (n_z - 1))
        else:  # caps: planar map
            for k, c in enumerate(tri):
                uv[t, k] = (0.5 + verts[c, 0] / (4 * radius_mm), 0.5 + verts[c, 1] / (4 * radius_mm))
    return uv


def make_textured_object(out_dir, label: str = "tex_000000", seed: int = 0, fmt: str = "obj", n_theta: int = 48, n_z: int = 40,
                         with_vertex_colors: bool = False) -> "RigidObject":
    """A UV-textured lathe object written as OBJ+MTL+PNG (`fmt="obj"`), ascii PLY with a per-face texcoord list + `comment
    TextureFile` (`"ply_face"`, the BOP/YCB-V layout) or binary PLY with per-vertex s,t (`"ply_vertex"`)."""
    from PIL import Image

    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    radius = 35.0
    v, f, c = make_lathe_mesh(seed, n_theta=n_theta, n_z=n_z, height_mm=140.0, radius_mm=radius)
    uv = lathe_corner_uvs(v, f, n_theta, n_z, radius)
    Image.fromarray(make_texture_image(seed)).save(out_dir / f"{label}.png")
    if fmt == "obj":
        lines = [f"mtllib {label}.mtl", "usemtl m0"]
        for i, p in enumerate(v):
            lines.append("v %.6f %.6f %.6f" % tuple(p) + (" %.6f %.6f %.6f" % tuple(c[i] / 255.0) if with_vertex_colors else ""))
        flat = uv.reshape(-1, 2)
        lines += ["vt %.6f %.6f" % tuple(t) for t in flat]
        for t, tri in enumerate(f):
            lines.append("f " + " ".join(f"{int(tri[k]) + 1}/{3 * t + k + 1}" for k in range(3)))
        (out_dir / f"{label}.obj").write_text("\n".join(lines) + "\n")
        (out_dir / f"{label}.mtl").write_text(f"newmtl m0\nKd 1 1 1\nmap_Kd {label}.png\n")
        path = out_dir / f"{label}.obj"
    elif fmt == "ply_face":
        hdr = ["ply", "format ascii 1.0", f"comment TextureFile {label}.png", f"element vertex {len(v)}", "property float x",
               "property float y", "property float z", f"element face {len(f)}", "property list uchar int vertex_indices",
               "property list uchar float texcoord", "end_header"]
        body = ["%.6f %.6f %.6f" % tuple(p) for p in v]
        body += ["3 %d %d %d 6 " % tuple(tri) + " ".join("%.6f" % x for x in uv[t].reshape(-1)) for t, tri in enumerate(f)]
        (out_dir / f"{label}.ply").write_text("\n".join(hdr + body) + "\n")
        path = out_dir / f"{label}.ply"
    elif fmt == "ply_vertex":
        th = np.arctan2(v[:, 1], v[:, 0]) / (2 * np.pi) % 1.0
        st = np.stack([th, (v[:, 2] - v[:, 2].min()) / np.ptp(v[:, 2])], axis=1)
        arr = np.zeros(len(v), dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("s", "<f4"), ("t", "<f4")]))
        arr["x"], arr["y"], arr["z"], arr["s"], arr["t"] = v[:, 0], v[:, 1], v[:, 2], st[:, 0], st[:, 1]
        hdr = ["ply", "format binary_little_endian 1.0", f"comment TextureFile {label}.png", f"element vertex {len(v)}"]
        hdr += [f"property float {n}" for n in ("x", "y", "z", "s", "t")]
        hdr += [f"element face {len(f)}", "property list uchar int vertex_indices", "end_header"]
        fa = np.zeros(len(f), dtype=np.dtype([("n", "u1"), ("i", "<i4", (3,))]))
        fa["n"], fa["i"] = 3, f
        path = out_dir / f"{label}.ply"
        with open(path, "wb") as fh:
            fh.write(("\n".join(hdr) + "\n").encode("ascii"))
            fh.write(arr.tobytes())
            fh.write(fa.tobytes())
    else:
        raise ValueError(fmt)
    return RigidObject(label=label, mesh_path=path, mesh_units="mm")


# --------------------------------------------------------------------------- #
# seeded weights in the reference checkpoint layout
# --------------------------------------------------------------------------- #
def _conv_w(g, cout, cin, k):
    std = math.sqrt(2.0 / (cout * k * k))  # kaiming_normal_(mode="fan_out", nonlinearity="relu")
    return torch.randn(cout, cin, k, k, generator=g) * std


def _bn(sd: Dict[str, torch.Tensor], g, prefix: str, c: int):
    sd[prefix + ".weight"] = torch.rand(c, generator=g) * 1.0 + 0.5
    sd[prefix + ".bias"] = torch.randn(c, generator=g) * 0.1
    sd[prefix + ".running_mean"] = torch.randn(c, generator=g) * 0.1
    sd[prefix + ".running_var"] = torch.rand(c, generator=g) * 1.0 + 0.5
    sd[prefix + ".num_batches_tracked"] = torch.tensor(1, dtype=torch.long)


def _linear(sd, g, prefix, out_f, in_f, scale=1.0):
    bound = 1.0 / math.sqrt(in_f)
    sd[prefix + ".weight"] = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound * scale
    sd[prefix + ".bias"] = (torch.rand(out_f, generator=g) * 2 - 1) * bound * scale


# Residual-branch gains that keep the 512-d features of the seeded nets O(1) (max |f| ~ 1-3 on rendered crops), as trained
# BN networks have: with plain kaiming weights + U[0.5,1.5] BN gains the residual stream of vanilla_resnet34 grows to 1e2-1e3,
# which forced every logit tolerance to be quoted relative to the feature scale.  With O(1) features the north-star
# tolerance (1e-4) is asserted ABSOLUTELY everywhere.
VANILLA_BN2_GAIN = 0.4     # gain on the last BN of every vanilla BasicBlock
WIDE_CONV2_GAIN = 0.35     # gain on the second conv of every pre-activation block
LOGIT_HEAD_SCALE = 2.0     # spreads the coarse / score logits of neighbouring hypotheses while |logit| stays O(1); a flipped silhouette sample
                           # (crop cameras agree to 1 ulp only) then moves a logit by <~5e-5
POSE_HEAD_SCALE = 0.05     # default pose-head weight scale: one refiner iteration moves a pose by ~1e-2 (rotation, rad; depth, rel.)


def make_state_dict(backbone_str: str, c_in: int, head: str, n_out: int, seed: int = 0,
                    pose_head_scale: float = POSE_HEAD_SCALE) -> Dict[str, torch.Tensor]:
    """state_dict of a PosePredictor (backbone.* + pose_fc.* | views_logits_head.*), seeded.
    BN running stats are non-trivial so folding is exercised; residual-branch gains keep the features O(1) (see above); the pose
    head is initialised around the identity update (bias = ortho6d identity, vx=vy=0, vz=1) with weights scaled by
    `pose_head_scale`: at the default 0.05 a conv-stack error reaches the pose essentially undamped (|d pose| ~ 0.03 |d f|) while
    chained refiner iterations still stay in the frustum."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    B = "backbone."
    stages = [64, 128, 256, 512]
    n_feat = 512
    if backbone_str == "vanilla_resnet34":
        sd[B + "conv1.weight"] = _conv_w(g, 64, c_in, 7)
        _bn(sd, g, B + "bn1", 64)
        inpl = 64
        for s, (planes, nb) in enumerate(zip(stages, [3, 4, 6, 3])):
            for i in range(nb):
                P = f"{B}layer{s + 1}.{i}."
                stride = 2 if (i == 0 and s > 0) else 1
                sd[P + "conv1.weight"] = _conv_w(g, planes, inpl, 3)
                _bn(sd, g, P + "bn1", planes)
                sd[P + "conv2.weight"] = _conv_w(g, planes, planes, 3)
                _bn(sd, g, P + "bn2", planes)
                sd[P + "bn2.weight"] = sd[P + "bn2.weight"] * VANILLA_BN2_GAIN
                if i == 0 and (stride != 1 or inpl != planes):
                    sd[P + "downsample.0.weight"] = _conv_w(g, planes, inpl, 1)
                    _bn(sd, g, P + "downsample.1", planes)
                inpl = planes
        _linear(sd, g, B + "fc", 512, 512)
    elif backbone_str in ("resnet34", "resnet18") or backbone_str.startswith("resnet34_width="):
        if backbone_str.startswith("resnet34_width="):   # WideResNet34(width=N), models/wide_resnet.py:62
            stages = [c * int(backbone_str.split("=")[1]) for c in stages]
        n_feat = stages[3]
        sd[B + "conv1.weight"] = _conv_w(g, stages[0], c_in, 5)
        _bn(sd, g, B + "bn1", stages[0])
        inpl = stages[0]
        for s, (planes, nb) in enumerate(zip(stages, [2, 2, 2, 2] if backbone_str == "resnet18" else [3, 4, 6, 3])):
            for i in range(nb):
                P = f"{B}layer{s + 1}.{i}."
                stride = 2 if (i == 0 and s > 0) else 1
                _bn(sd, g, P + "bn1", inpl)
                sd[P + "conv1.weight"] = _conv_w(g, planes, inpl, 3)
                _bn(sd, g, P + "bn2", planes)
                sd[P + "conv2.weight"] = _conv_w(g, planes, planes, 3) * WIDE_CONV2_GAIN
                if i == 0 and (stride != 1 or inpl != planes):
                    sd[P + "downsample.weight"] = _conv_w(g, planes, inpl, 1)
                inpl = planes
    else:
        raise ValueError(backbone_str)
    if head == "pose":
        _linear(sd, g, "pose_fc", 9, n_feat, scale=pose_head_scale)
        sd["pose_fc.bias"] = sd["pose_fc.bias"] + torch.tensor([1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0])
    else:
        _linear(sd, g, "views_logits_head", n_out, n_feat, scale=LOGIT_HEAD_SCALE)
    return sd


def make_cfg(role: str, backbone_str: str = "vanilla_resnet34", rgbd: bool = False) -> SimpleNamespace:
    """The cfg fields create_model_pose reads (src/megapose/training/pose_models_cfg.py:95-137) for the released
    recipes (scripts/run_megapose_training.py:120-153): refiner = 4 views TCO+front_3views + normals;
    coarse = 1 view, logits head, no pose head."""
    if role == "coarse":
        # (make_coarse_cfg, scripts/run_megapose_training.py:131-142: one view, "1view_TCO", remove_TCO_rendering = True)
        return SimpleNamespace(backbone_str=backbone_str, n_rendered_views=1, multiview_type="1view_TCO", views_inplane_rotations=False,
                               render_normals=True, render_depth=False, input_depth=False, predict_rendered_views_logits=True,
                               remove_TCO_rendering=True, predict_pose_update=False, depth_normalization_type="tCR_scale_clamp_center",
                               depth_augmentation=False, renderer="panda3d")
    return SimpleNamespace(backbone_str=backbone_str, n_rendered_views=4, multiview_type="TCO+front_3views", views_inplane_rotations=False,
                           render_normals=True, render_depth=rgbd, input_depth=rgbd, predict_rendered_views_logits=False,
                           remove_TCO_rendering=False, predict_pose_update=True, depth_normalization_type="tCR_scale_clamp_center",
                           depth_augmentation=False, renderer="panda3d")


def n_inputs_for(cfg) -> int:
    """pose_models_cfg.py:96-103"""
    n = 3 + (1 if cfg.input_depth else 0)
    per_view = 3 + (3 if cfg.render_normals else 0) + (1 if cfg.render_depth else 0)
    return n + per_view * cfg.n_rendered_views


def random_pose(rng: np.random.RandomState, z_range=(0.35, 0.7), xy_frac=0.15) -> np.ndarray:
    """A random TCO with the object comfortably inside a 640x480 frame."""
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    zc = rng.uniform(*z_range)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = [rng.uniform(-xy_frac, xy_frac) * zc, rng.uniform(-xy_frac, xy_frac) * zc, zc]
    return T.astype(np.float32)
