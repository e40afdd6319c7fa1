"""Generate tests/golden/*.npz by running the REFERENCE's own Python (imported read-only from /root/reference/src via
oracle/ref_import.py).  CONTAINER-ONLY; the vectors are committed so that the oracle can be pinned anywhere.

    python -m oracle.make_golden

Files
  geometry.npz   reference outputs of the pure-torch pose/geometry functions on seeded inputs
  backbones.npz  reference module outputs (torchvision_resnet.resnet34 / WideResNet34 / WideResNet18) with the seeded
                 reference-layout state_dicts of tests.support.synthetic.make_state_dict loaded strict=True
  pipeline.npz   the reference's unmodified PoseEstimator.run_inference_pipeline (+ create_model_pose / PosePredictor)
                 driven with the oracle renderer on a synthetic scene: coarse logits, top-K, per-iteration refined poses,
                 scores, final pose
"""
from __future__ import annotations

import sys
import tempfile
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pandas as pd
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"


def geometry_vectors(r) -> dict:
    g = torch.Generator().manual_seed(123)
    b, n = 5, 300
    out = {}
    T = torch.eye(4).repeat(b, 1, 1)
    T[:, :3, :3] = torch.linalg.qr(torch.randn(b, 3, 3, generator=g))[0] + 0.02 * torch.randn(b, 3, 3, generator=g)
    T[:, :3, 3] = torch.tensor([0.0, 0.0, 0.6]) + 0.1 * torch.randn(b, 3, generator=g)
    K = torch.tensor([[600.0, 0, 320], [0, 590.0, 240], [0, 0, 1]]).repeat(b, 1, 1) + torch.randn(b, 3, 3, generator=g) * torch.tensor(
        [[5.0, 0, 5], [0, 5.0, 5], [0, 0, 0]])
    pts = torch.randn(b, n, 3, generator=g) * 0.05
    out["T"], out["K"], out["pts"] = T, K, pts
    Tn = r.to.normalize_T(T)
    out["normalize_T"] = Tn
    p6 = torch.randn(b, 6, generator=g)
    out["p6"], out["ortho6d"] = p6, r.rot.compute_rotation_matrix_from_ortho6d(p6)
    out["invert"] = r.to.invert_transform_matrices(Tn)
    out["uv"] = r.cg.project_points_robust(pts, K, Tn)
    out["boxes_uv"] = r.cg.boxes_from_uv(out["uv"])
    imgs = torch.zeros(b, 3, 480, 640)
    tCR = Tn[:, :3, 3] + 0.01 * torch.randn(b, 3, generator=g)
    out["tCR"] = tCR
    out["boxes_crop"], _ = r.cr.deepim_crops_robust(images=imgs, obs_boxes=out["boxes_uv"], K=K, TCO_pred=Tn, tCR_in=tCR, O_vertices=pts,
                                                   output_size=(240, 320), lamb=1.4, return_crops=False)
    out["K_crop"] = r.cg.get_K_crop_resize(K=K.clone(), boxes=out["boxes_crop"], orig_size=(480, 640), crop_resize=(240, 320))
    boxes2d = torch.tensor([[200.0, 150, 330, 300]]).repeat(b, 1) + torch.rand(b, 4, generator=g) * 30
    R = r.rot.compute_rotation_matrix_from_ortho6d(torch.randn(b, 6, generator=g))
    out["boxes2d"], out["R"] = boxes2d, R
    out["TCO_init"] = r.co.TCO_init_from_boxes_autodepth_with_R(boxes2d, pts, K, R)
    v = torch.randn(b, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 1.0])
    dR = r.rot.compute_rotation_matrix_from_ortho6d(torch.randn(b, 6, generator=g) * 0.1 + torch.tensor([1.0, 0, 0, 0, 1.0, 0]))
    out["vxvyvz"], out["dR"] = v, dR
    out["pose_update"] = r.co.pose_update_with_reference_point(Tn, out["K_crop"], v, dR, tCR)
    out["sample_ids_5042_2000"] = np.random.RandomState(0).choice(5042, size=2000, replace=False)  # mesh_ops.py:82-84
    sp = r.mo.sample_points(torch.arange(5042 * 3).float().view(1, 5042, 3), 2000, deterministic=True)
    out["sample_points_first"] = sp[0, :, 0] / 3
    padded = r.rmd.pad_stack_tensors([torch.arange(30).float().view(10, 3), torch.arange(60).float().view(20, 3)], fill="select_random",
                                     deterministic=True)
    out["pad_stack"] = padded
    out["so3_72"] = r.tu.load_SO3_grid(72)
    out["so3_576"] = r.tu.load_SO3_grid(576)
    out["mv_TCV_O"] = r.mv.make_TCO_multiview(Tn, Tn[:, :3, 3], multiview_type="TCO+front_3views", n_views=4)
    # the other view lists of lib3d/multiview.py:197-246 (transforms3d is absent: its euler2mat(0, 0, angle) = Rz(angle) is stubbed)
    import transforms3d

    transforms3d.euler.euler2mat = lambda ai, aj, ak: np.array([[np.cos(ak), -np.sin(ak), 0], [np.sin(ak), np.cos(ak), 0], [0, 0, 1.0]])
    out["mv_TCV_O_front1"] = r.mv.make_TCO_multiview(Tn, Tn[:, :3, 3], multiview_type="TCO+front_1view", n_views=2)
    out["mv_TCV_O_sphere26"] = r.mv.make_TCO_multiview(Tn, Tn[:, :3, 3], multiview_type="sphere_26views", n_views=27)
    out["mv_TCV_O_front3_noTCO"] = r.mv.make_TCO_multiview(Tn, Tn[:, :3, 3], multiview_type="TCO+front_3views", n_views=3, remove_TCO_rendering=True)
    out["mv_TCV_O_front3_noTCO_inplane"] = r.mv.make_TCO_multiview(Tn, Tn[:, :3, 3], multiview_type="TCO+front_3views", n_views=12,
                                                                    remove_TCO_rendering=True, views_inplane_rotations=True)
    from oracle import thirdparty as tp

    img = torch.rand(2, 4, 60, 80, generator=g)
    rois = torch.tensor([[0, 5.5, 3.2, 50.1, 40.7], [1, -10.0, -5.0, 70.0, 50.0], [1, 30.0, 20.0, 95.0, 75.0]])
    out["roi_img"], out["roi_rois"] = img, rois
    out["roi_out_unpinned"] = tp.roi_align(img, rois, (12, 16), sampling_ratio=4)  # restated third party: NOT a reference output
    depth = torch.rand(b, 2, 1, 4, 4, generator=g)
    out["depth_in"] = depth
    return {k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}


def synthetic_input(n, c, h, w) -> torch.Tensor:
    """Closed-form pseudo-image (not stored in the golden file; tests rebuild it)."""
    i = np.arange(n * c * h * w, dtype=np.float64).reshape(n, c, h, w)
    return torch.from_numpy((0.5 + 0.5 * np.sin(i * 0.7310585 + (i % 97) * 0.11)).astype(np.float32))


def backbone_vectors(r) -> dict:
    from tests.support import synthetic as syn

    out = {}
    g = torch.Generator().manual_seed(7)
    for kind, c_in, head, n_out in (("vanilla_resnet34", 9, "logits", 1), ("vanilla_resnet34", 27, "pose", 9), ("resnet34", 27, "pose", 9),
                                    ("resnet18", 9, "logits", 1), ("resnet34", 32, "pose", 9), ("resnet34_width=2", 9, "logits", 1)):
        sd = syn.make_state_dict(kind, c_in, head, n_out, seed=1)
        if kind == "vanilla_resnet34":
            m = r.tvr.resnet34(num_classes=512, n_input_channels=c_in)
        elif kind == "resnet34":
            m = r.wr.WideResNet34(n_inputs=c_in)
        elif kind.startswith("resnet34_width="):   # training/pose_models_cfg.py:114-116
            m = r.wr.WideResNet34(n_inputs=c_in, width=int(kind.split("resnet34_width=")[1]))
        else:
            m = r.wr.WideResNet18(n_inputs=c_in)
        bsd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
        m.load_state_dict(bsd, strict=True)
        m.eval()
        x = synthetic_input(2, c_in, 96, 128)
        with torch.no_grad():
            f = m(x)
            if f.dim() == 4:
                f = f.flatten(2).mean(dim=-1)
        out[f"{kind}_{c_in}_feat"] = f.numpy()
    return out


def make_scene(tmp, n_objects=1, seed=0):
    """Synthetic scene shared by the golden generator and the tests (tests/scene_util.py re-implements nothing: it calls this
    through the saved npz inputs)."""
    from tests.support import synthetic as syn
    from oracle import mesh_loader
    from oracle import raster as orr

    ds = syn.make_object_dataset(tmp, n_objects=n_objects, seed=seed)
    meshes = {o.label: mesh_loader.load_object(o) for o in ds.list_objects}   # (the oracle's own reader, not the product's)
    rng = np.random.RandomState(seed + 100)
    K = syn.K_EXAMPLE.astype(np.float32)
    img = rng.uniform(0, 1, size=(480, 640, 3)).astype(np.float32) * 0.3
    depth_all = np.zeros((480, 640), np.float32)
    bboxes, poses = [], []
    for i, o in enumerate(ds.list_objects):
        T = syn.random_pose(rng, z_range=(0.45, 0.7), xy_frac=0.12 if n_objects == 1 else 0.3)
        rgb, _, dep = orr.render(meshes[o.label], T[None], K[None], 480, 640, 2)
        m = dep[0] > 0
        closer = m & ((depth_all == 0) | (dep[0] < depth_all))
        img[closer] = rgb[0][closer]
        depth_all[closer] = dep[0][closer]
        ys, xs = np.nonzero(m)
        bboxes.append([xs.min(), ys.min(), xs.max(), ys.max()])
        poses.append(T)
    img_u8 = np.round(img * 255).astype(np.uint8)
    return ds, meshes, img_u8, depth_all, K, np.asarray(bboxes, np.float32), np.stack(poses)


def pipeline_vectors(r) -> dict:
    from tests.support import synthetic as syn
    from oracle import raster as orr

    tmp = Path(tempfile.mkdtemp(prefix="mp_golden_"))
    ds, meshes, img_u8, depth, K, bboxes, gt = make_scene(tmp, n_objects=1, seed=0)
    ref_objs = [r.RigidObject(label=o.label, mesh_path=o.mesh_path, mesh_units="mm") for o in ds.list_objects]
    ref_ds = r.RigidObjectDataset(ref_objs)
    mesh_db = r.rmd.MeshDataBase.from_object_ds(ref_ds).batched()
    renderer = orr.OracleBatchRenderer(meshes)
    models = {}
    for role in ("coarse", "refiner"):
        cfg = syn.make_cfg(role, "vanilla_resnet34")
        head, n_out = ("pose", 9) if role == "refiner" else ("logits", 1)
        sd = syn.make_state_dict("vanilla_resnet34", syn.n_inputs_for(cfg), head, n_out, seed={"coarse": 11, "refiner": 12}[role])
        import megapose.models.pose_rigid as pr

        pr.Panda3dBatchRenderer = orr.OracleBatchRenderer  # the isinstance assert in render_images_multiview (:378)
        from types import SimpleNamespace

        class _Cfg(SimpleNamespace):   # attribute access + `"x" in cfg`, like the OmegaConf node the reference passes around
            def __contains__(self, key):
                return hasattr(self, key)

        cfg = _Cfg(**cfg) if isinstance(cfg, dict) else _Cfg(**{k: getattr(cfg, k) for k in vars(cfg)})
        model = r.pmc.create_model_pose(r.pmc.check_update_config(cfg), renderer=renderer, mesh_db=mesh_db)
        model.load_state_dict(sd, strict=True)
        model.eval()
        model.cfg = cfg
        models[role] = model
    est = r.pe.PoseEstimator(refiner_model=models["refiner"], coarse_model=models["coarse"], bsz_objects=2, bsz_images=24, SO3_grid_size=72)
    obs = r.ty.ObservationTensor.from_numpy(img_u8, None, K)
    det = r.tc.PandasTensorCollection(infos=pd.DataFrame(dict(label=[o.label for o in ref_objs], batch_im_id=0, instance_id=np.arange(len(ref_objs)))),
                                      bboxes=torch.as_tensor(bboxes))
    torch.manual_seed(0)
    final, extra = est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=3, n_pose_hypotheses=2)
    out = dict(img_u8=img_u8, K=K, bboxes=bboxes, gt_TCO=gt, n_render_views=np.int64(renderer.n_views))
    cd = extra["coarse"]
    out["coarse_TCO"] = cd["preds"].poses.numpy()
    out["coarse_logits"] = cd["data"]["logits"].numpy()
    out["filtered_hyp_ids"] = extra["coarse_filter"]["preds"].infos["hypothesis_id"].values.astype(np.int64)
    for n in range(1, 4):
        p = extra["refiner_all_hypotheses"]["preds"][f"iteration={n}"]
        out[f"refiner_poses_{n}"] = p.poses.numpy()
        out[f"refiner_K_crop_{n}"] = p.K_crop.numpy()
        out[f"refiner_boxes_crop_{n}"] = p.boxes_crop.numpy()
    out["scoring_logits"] = extra["scoring"]["data"]["logits"].numpy()
    out["final_TCO"] = final.poses.numpy()
    out["final_columns"] = np.array(list(final.infos.columns))
    out["extra_keys"] = np.array(sorted(extra.keys()))
    return out


def main():
    from oracle import ref_import

    r = ref_import.ref()
    GOLD.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)
    with torch.no_grad():
        np.savez_compressed(GOLD / "geometry.npz", **geometry_vectors(r))
        print("geometry.npz done")
        np.savez_compressed(GOLD / "backbones.npz", **backbone_vectors(r))
        print("backbones.npz done")
        np.savez_compressed(GOLD / "pipeline.npz", **pipeline_vectors(r))
        print("pipeline.npz done")


if __name__ == "__main__":
    main()
