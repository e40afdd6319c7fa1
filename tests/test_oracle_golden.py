"""CPU: pin the oracle (oracle/*.py restatements) against golden vectors produced by the REFERENCE's own code
(oracle/make_golden.py, run in the build container where /root/reference exists).  Tolerances: the restatements use
the same fp32 formulas, so 1e-6 abs on O(1) quantities / 1e-6 relative on pixel-scale ones; CNN features 1e-4 relative."""
import re
import sys
import tempfile
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
import torch

GOLD = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def geo():
    return {k: v for k, v in np.load(GOLD / "geometry.npz").items()}


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_geometry_against_reference_outputs(geo):
    from oracle import geometry as og

    T, K, pts = t(geo["T"]), t(geo["K"]), t(geo["pts"])
    Tn = og.normalize_T(T)
    assert (Tn - t(geo["normalize_T"])).abs().max() < 1e-6
    assert (og.ortho6d_to_R(t(geo["p6"])) - t(geo["ortho6d"])).abs().max() < 1e-6
    assert (og.invert_T(Tn) - t(geo["invert"])).abs().max() < 1e-6
    uv = og.project_points_robust(pts, K, Tn)
    assert (uv - t(geo["uv"])).abs().max() < 1e-3  # pixels ~ 1e2..1e3
    assert (og.boxes_from_uv(uv) - t(geo["boxes_uv"])).abs().max() < 1e-3
    bc = og.crop_boxes_robust(t(geo["boxes_uv"]), K, Tn, t(geo["tCR"]), pts, (480, 640))
    assert (bc - t(geo["boxes_crop"])).abs().max() < 1e-3
    Kc = og.get_K_crop_resize(K, t(geo["boxes_crop"]), (240, 320))
    assert ((Kc - t(geo["K_crop"])).abs() / t(geo["K_crop"]).abs().clamp(min=1)).max() < 1e-6
    Ti = og.TCO_init_from_boxes_autodepth_with_R(t(geo["boxes2d"]), pts, K, t(geo["R"]))
    assert (Ti - t(geo["TCO_init"])).abs().max() < 1e-6
    pu = og.pose_update_with_reference_point(Tn, t(geo["K_crop"]), t(geo["vxvyvz"]), t(geo["dR"]), t(geo["tCR"]))
    assert (pu - t(geo["pose_update"])).abs().max() < 1e-6
    assert np.array_equal(og.sample_point_ids(5042, 2000), geo["sample_ids_5042_2000"])
    assert np.array_equal(geo["sample_points_first"].astype(np.int64), geo["sample_ids_5042_2000"])
    ps = og.pad_stack_points([torch.arange(30).float().view(10, 3), torch.arange(60).float().view(20, 3)])
    assert torch.equal(ps, t(geo["pad_stack"]))
    mv = og.make_TCO_multiview(Tn, Tn[:, :3, 3], "TCO+front_3views", 4)
    assert (mv - t(geo["mv_TCV_O"])).abs().max() < 1e-6
    # the other view lists of lib3d/multiview.py:197-246, as the reference function returned them
    for key, args in {"mv_TCV_O_front1": ("TCO+front_1view", 2, False, False), "mv_TCV_O_sphere26": ("sphere_26views", 27, False, False),
                      "mv_TCV_O_front3_noTCO": ("TCO+front_3views", 3, True, False),
                      "mv_TCV_O_front3_noTCO_inplane": ("TCO+front_3views", 12, True, True)}.items():
        mv = og.make_TCO_multiview(Tn, Tn[:, :3, 3], *args)
        assert mv.shape == t(geo[key]).shape and (mv - t(geo[key])).abs().max() < 1e-6, key


def test_so3_grid_matches_reference(geo):
    """product loader vs the reference's load_SO3_grid outputs (roma formula restated)."""
    from megapose6d_amd.pose_estimator import load_SO3_grid

    for n in (72, 576):
        R = load_SO3_grid(n)
        assert R.shape == (n, 3, 3) and R.dtype == torch.float32
        assert (R - t(geo[f"so3_{n}"])).abs().max() < 1e-6
        assert (R @ R.transpose(1, 2) - torch.eye(3)).abs().max() < 1e-5


def test_host_side_mesh_db_matches_reference_sampling(geo, object_dataset):
    from megapose6d_amd.mesh_db import MeshDataBase, deterministic_point_ids
    from oracle import geometry as og

    assert np.array_equal(deterministic_point_ids(5042, 2000), geo["sample_ids_5042_2000"])
    assert np.array_equal(deterministic_point_ids(5042, 200), geo["sample_ids_5042_2000"][:200])  # permutation prefix
    db = MeshDataBase.from_object_ds(object_dataset).batched()
    assert db.points.shape[0] == 3 and db.points.dtype == torch.float32
    sel = db.select(["obj_000002", "obj_000000"])
    assert torch.equal(sel.points[1], db.points[0])
    assert torch.equal(sel.sample_points(2000, deterministic=True), db.sampled_points(2000)[[2, 0]])
    ref = og.pad_stack_points([db.points[i] for i in range(3)])
    assert torch.equal(ref, db.points)
    with pytest.raises(KeyError):
        db.select(["nope"])


def test_roi_align_restatement_is_self_consistent(geo):
    """roi_align is third-party (torchvision 0.12) -> 'parity unpinned'; this guards the restatement against drift."""
    from oracle import thirdparty as tp

    out = tp.roi_align(t(geo["roi_img"]), t(geo["roi_rois"]), (12, 16), sampling_ratio=4)
    assert (out - t(geo["roi_out_unpinned"])).abs().max() < 1e-7
    # analytic check: constant image -> constant crop; fully outside -> zeros
    img = torch.full((1, 1, 20, 30), 0.7)
    o = tp.roi_align(img, torch.tensor([[0, 2.0, 2.0, 20.0, 15.0], [0, 100.0, 100.0, 120.0, 110.0]]), (4, 6), sampling_ratio=4)
    assert (o[0] - 0.7).abs().max() < 1e-6 and o[1].abs().max() == 0


def test_backbone_restatement_against_reference_modules():
    from tests.support import synthetic as syn
    from oracle import backbones as ob
    from oracle.make_golden import synthetic_input

    gold = np.load(GOLD / "backbones.npz")
    for key in gold.files:
        m = re.match(r"(.+)_(\d+)_feat", key)
        kind, c_in = m.group(1), int(m.group(2))
        head, n_out = ("logits", 1) if c_in == 9 else ("pose", 9)
        sd = syn.make_state_dict(kind, c_in, head, n_out, seed=1)
        with torch.no_grad():
            f = ob.net_forward(sd, kind, synthetic_input(2, c_in, 96, 128))["features"]
        ref = t(gold[key])
        assert (f - ref).abs().max() < 1e-4 * max(1.0, ref.abs().max().item()), key


def test_state_dict_keys_match_product_module():
    """the product's parameter-hosting module must load reference-layout checkpoints strict=True."""
    from tests.support import synthetic as syn
    from megapose6d_amd.pose_rigid import HipBackbone

    for kind, c_in in (("vanilla_resnet34", 27), ("resnet34", 9), ("resnet18", 32)):
        sd = syn.make_state_dict(kind, c_in, "pose", 9, seed=0)
        bsd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
        HipBackbone(kind, c_in).load_state_dict(bsd, strict=True)


@pytest.fixture(scope="module")
def pipeline_gold():
    return {k: v for k, v in np.load(GOLD / "pipeline.npz", allow_pickle=False).items()}


def build_oracle_estimator(tmp, backbone="vanilla_resnet34", grid=72):
    from tests.support import synthetic as syn
    from oracle import mesh_loader
    from oracle import pipeline as op
    from oracle import raster as orr

    ds = syn.make_object_dataset(tmp, n_objects=1, seed=0)
    meshes, db = mesh_loader.load_dataset(ds)          # the oracle's own reader, not the product's (megapose6d_amd.mesh_io)
    load_SO3_grid = lambda n: mesh_loader.load_so3_grid(Path(__file__).resolve().parent.parent / "megapose6d_amd" / "data" / f"so3_grid_{n}_xyzw.npy")
    renderer = orr.OracleBatchRenderer(meshes)
    preds = {}
    for role, seed in (("coarse", 11), ("refiner", 12)):
        cfg = syn.make_cfg(role, backbone)
        head, n_out = ("pose", 9) if role == "refiner" else ("logits", 1)
        sd = syn.make_state_dict(backbone, syn.n_inputs_for(cfg), head, n_out, seed=seed)
        preds[role] = op.OraclePosePredictor(cfg, sd, db.labels.tolist(), db.points, renderer)
    return ds, op.OraclePoseEstimator(preds["coarse"], preds["refiner"], load_SO3_grid(grid), bsz=24, bsz_refiner=2), renderer


def test_oracle_pipeline_against_reference_orchestration(pipeline_gold, tmp_path):
    """The reference's unmodified PoseEstimator/PosePredictor (run in the container with the oracle renderer) vs the
    oracle's own restatement of the orchestration: coarse logits, top-K ids, every refiner iteration, scores, final pose."""
    g = pipeline_gold
    torch.set_num_threads(8)
    ds, est, renderer = build_oracle_estimator(tmp_path)
    images = torch.from_numpy(g["img_u8"]).float().permute(2, 0, 1)[None] / 255
    K = torch.from_numpy(g["K"])[None].float()
    infos = pd.DataFrame(dict(label=[o.label for o in ds.list_objects], batch_im_id=0, instance_id=[0]))
    res = est.run(images, K, infos, torch.from_numpy(g["bboxes"]), n_refiner_iterations=3, n_pose_hypotheses=2)
    assert (res["coarse_TCO"].numpy() - g["coarse_TCO"]).max() < 1e-6
    lscale = max(1.0, float(np.abs(g["coarse_logits"]).max()))  # = 1 with the O(1) seeded networks
    assert lscale < 5
    assert np.abs(res["coarse_logits"].numpy() - g["coarse_logits"].flatten()).max() < 1e-4 * lscale
    assert sorted(res["filtered_infos"]["hypothesis_id"].tolist()) == sorted(g["filtered_hyp_ids"].tolist())
    order = [res["filtered_infos"]["hypothesis_id"].tolist().index(h) for h in g["filtered_hyp_ids"].tolist()]
    for n in range(1, 4):
        assert np.abs(res["refiner_poses"][n - 1].numpy()[order] - g[f"refiner_poses_{n}"]).max() < 1e-5
        rel = np.abs(res["refiner_K_crop"][n - 1].numpy()[order] - g[f"refiner_K_crop_{n}"]) / np.maximum(np.abs(g[f"refiner_K_crop_{n}"]), 1)
        assert rel.max() < 1e-5
    assert np.abs(res["scoring_logits"].numpy()[order] - g["scoring_logits"].flatten()).max() < 1e-4 * lscale
    assert np.abs(res["final_TCO"].numpy() - g["final_TCO"]).max() < 1e-5
    # render-request count predicted by SURVEY.md section 3: M + K*n_iter*4 + K
    assert renderer.n_views == int(g["n_render_views"]) == 72 + 2 * 3 * 4 + 2
    assert set(["coarse", "coarse_filter", "refiner", "refiner_all_hypotheses", "scoring", "time", "timing_str"]) <= set(g["extra_keys"].tolist())


def test_icp_oracle_recovers_known_offset():
    """oracle/icp.py on a closed-form depth map: a 4 mm depth offset + 2 px shift is pulled back; too few points -> rejected"""
    import numpy as np

    from oracle import icp

    H, W = 120, 160
    K = np.array([[150, 0, 80], [0, 150, 60], [0, 0, 1.0]])
    ys, xs = np.mgrid[0:H, 0:W]
    surf = lambda x, y: (0.6 + 0.05 * np.sin(x / 9.0) * np.cos(y / 7.0)).astype(np.float32)
    meas = surf(xs, ys)
    meas[:20] = 0
    rend = surf(xs + 2.0, ys) + 0.004
    T = np.eye(4, dtype=np.float32)
    T_ref, rv, res = icp.icp_refine(meas, rend, K, T, n_min_points=500)
    assert rv == 0 and 0 <= res < 2e-3
    assert -0.006 < T_ref[2, 3] < -0.002          # moves the model back towards the measurement
    _, rv2, _ = icp.icp_refine(meas, rend, K, T, n_min_points=10 ** 6)
    assert rv2 == -1
    n = icp.target_normals(meas, K)
    assert np.allclose(np.linalg.norm(n[40:100, 20:140], axis=-1), 1.0, atol=1e-6) and (n[..., 2] <= 0).all()
