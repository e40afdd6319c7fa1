"""TEST / BENCH SUPPORT (not product code): synthetic scenes + model assembly shared by bench.py, smoke() and the tests
(no oracle imports).
The observation is rendered by the engine's own rasteriser over a uniform-noise background."""
from __future__ import annotations

import tempfile
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from megapose6d_amd.load_model import build_pose_model, make_detections
from megapose6d_amd.mesh_db import MeshDataBase
from megapose6d_amd.pose_estimator import PoseEstimator
from megapose6d_amd.renderer import Panda3dBatchRenderer
from megapose6d_amd.types import ObservationTensor, Panda3dLightData

from . import synthetic as syn


def build_estimator(object_dataset, backbone: str = "vanilla_resnet34", rgbd: bool = False, SO3_grid_size: int = 576,
                    seeds=(11, 12), pose_head_scale: float = syn.POSE_HEAD_SCALE, **est_kwargs) -> PoseEstimator:
    """Seeded random-weight coarse + refiner models in the released recipes' structure, on the HIP engine."""
    renderer = Panda3dBatchRenderer(object_dataset, n_workers=1, preload_cache=True)
    mesh_db = MeshDataBase.from_object_ds(object_dataset).batched().cuda()
    models = {}
    for role, seed in zip(("coarse", "refiner"), seeds):
        cfg = syn.make_cfg(role, backbone, rgbd=(rgbd and role == "refiner"))
        head, n_out = ("pose", 9) if role == "refiner" else ("logits", 1)
        sd = syn.make_state_dict(backbone, syn.n_inputs_for(cfg), head, n_out, seed=seed, pose_head_scale=pose_head_scale)
        models[role] = build_pose_model(cfg, sd, renderer, mesh_db)
    return PoseEstimator(refiner_model=models["refiner"], coarse_model=models["coarse"], SO3_grid_size=SO3_grid_size, **est_kwargs)


def render_observation(renderer: Panda3dBatchRenderer, labels: List[str], poses: np.ndarray, K: np.ndarray, seed: int = 0,
                       with_depth: bool = False, hw=(480, 640)) -> Tuple[torch.Tensor, np.ndarray]:
    """One 640x480 frame containing all objects (painter's order by depth buffer) -> images [1,C,H,W] on the GPU, bboxes."""
    h, w = hw
    dev = torch.device("cuda")
    n = len(labels)
    Kt = torch.from_numpy(np.repeat(K[None].astype(np.float32), n, 0)).to(dev)
    out = renderer.render(labels, torch.from_numpy(poses.astype(np.float32)).to(dev), Kt,
                          [[Panda3dLightData("ambient", (1.0, 1.0, 1.0, 1.0))]] * n, (h, w), render_depth=True, render_normals=False)
    rgb, dep = out.rgbs, out.depths[:, 0]
    g = torch.Generator(device="cpu").manual_seed(seed)
    img = (torch.rand(3, h, w, generator=g) * 0.3).to(dev)
    zbuf = torch.zeros(h, w, device=dev)
    bboxes = []
    for i in range(n):
        m = dep[i] > 0
        closer = m & ((zbuf == 0) | (dep[i] < zbuf))
        img = torch.where(closer[None], rgb[i], img)
        zbuf = torch.where(closer, dep[i], zbuf)
        ys, xs = torch.nonzero(m, as_tuple=True)
        bboxes.append([xs.min().item(), ys.min().item(), xs.max().item(), ys.max().item()])
    img = torch.round(img * 255) / 255
    if with_depth:
        gd = torch.Generator(device="cpu").manual_seed(seed + 1)
        noise = (torch.randn(h, w, generator=gd) * 0.002).to(dev)
        d = torch.where(zbuf > 0, zbuf + noise, zbuf)
        drop = (torch.rand(h, w, generator=gd) < 0.05).to(dev)
        d = torch.where(drop, torch.zeros_like(d), d)
        img = torch.cat([img, d[None]], 0)
    return img[None].contiguous(), np.asarray(bboxes, np.float32)


def make_scene(n_objects: int = 1, seed: int = 0, backbone: str = "vanilla_resnet34", rgbd: bool = False, SO3_grid_size: int = 576,
               tmp_dir: Optional[str] = None, **est_kwargs):
    """-> (estimator, observation, detections, gt_poses)"""
    tmp = Path(tmp_dir or tempfile.mkdtemp(prefix="mp_scene_"))
    ds = syn.make_object_dataset(tmp, n_objects=n_objects, seed=seed)
    est = build_estimator(ds, backbone, rgbd, SO3_grid_size, **est_kwargs)
    rng = np.random.RandomState(seed + 100)
    labels = [o.label for o in ds.list_objects]
    poses = np.stack([syn.random_pose(rng, z_range=(0.45, 0.7), xy_frac=0.12 if n_objects == 1 else 0.3) for _ in labels])
    K = syn.K_EXAMPLE.astype(np.float32)
    images, bboxes = render_observation(est.coarse_model.renderer, labels, poses, K, seed=seed, with_depth=rgbd)
    obs = ObservationTensor(images, torch.from_numpy(K)[None].cuda())
    det = make_detections(labels, bboxes).cuda()
    return est, obs, det, poses


def make_multi_frame_scene(n_frames: int = 8, n_per_frame: int = 8, n_meshes: int = 16, seed: int = 40, backbone: str = "vanilla_resnet34",
                           rgbd: bool = False, SO3_grid_size: int = 576, tmp_dir: Optional[str] = None, depth_obs: bool = False,
                           **est_kwargs):
    """BASELINE.json configs[3]/[4] shape: `n_frames` 640x480 frames with `n_per_frame` detections each over `n_meshes` distinct
    meshes (every mesh appears n_frames * n_per_frame / n_meshes times).  -> (estimator, observation, detections, object dataset).
    rgbd: RGBD refiner + RGBD frames; depth_obs: RGBD frames for an RGB model pair (the "...-icp" recipes: the depth image only feeds
    the depth refiner, reference utils/load_model.py NAMED_MODELS)."""
    import pandas as pd

    from megapose6d_amd.tcoll import PandasTensorCollection

    tmp = Path(tmp_dir or tempfile.mkdtemp(prefix="mp_scene_mf_"))
    ds = syn.make_object_dataset(tmp, n_objects=n_meshes, seed=seed, n_theta=48, n_z=50)
    est = build_estimator(ds, backbone, rgbd, SO3_grid_size, **est_kwargs)
    r = est.coarse_model.renderer
    rng = np.random.RandomState(seed - 31)
    K = syn.K_EXAMPLE.astype(np.float32)
    labels_all = [o.label for o in ds.list_objects]
    frames, rows, boxes = [], [], []
    for f in range(n_frames):
        labs = [labels_all[(2 * f + j) % n_meshes] for j in range(n_per_frame)]
        poses = np.stack([syn.random_pose(rng, (0.5, 0.8), 0.3) for _ in labs])
        im, bb = render_observation(r, labs, poses, K, seed=f, with_depth=rgbd or depth_obs)
        frames.append(im)
        rows += [dict(label=l, batch_im_id=f) for l in labs]
        boxes.append(bb)
    obs = ObservationTensor(torch.cat(frames), torch.from_numpy(np.repeat(K[None], n_frames, 0)).cuda())
    det = PandasTensorCollection(pd.DataFrame(rows), bboxes=torch.from_numpy(np.concatenate(boxes)).cuda())
    return est, obs, det, ds
