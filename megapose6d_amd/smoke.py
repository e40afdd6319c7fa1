"""smoke(): one tiny invocation of the hot path on cuda:0 (1 object, 72-rotation grid, top-2, 1 refiner iteration),
checked against the CPU oracle.  The oracle is imported here ONLY as the checker."""
from __future__ import annotations

import tempfile
from pathlib import Path

import numpy as np
import pandas as pd
import torch


def run_smoke() -> None:
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    torch.cuda.set_device(0)
    from . import engine as eng
    from . import synthetic as syn
    from .scene import make_scene

    n_cu, lds, arch = eng.device_info()
    tmp = tempfile.mkdtemp(prefix="mp_smoke_")
    est, obs, det, gt = make_scene(n_objects=1, seed=0, SO3_grid_size=72, tmp_dir=tmp)
    final, extra = est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=1, n_pose_hypotheses=2)
    torch.cuda.synchronize()
    assert final.poses.shape == (1, 4, 4) and torch.isfinite(final.poses).all()

    # checker: CPU oracle on the same inputs
    from megapose6d_amd import mesh_io
    from megapose6d_amd.mesh_db import MeshDataBase
    from megapose6d_amd.pose_estimator import load_SO3_grid
    from oracle import pipeline as op
    from oracle import raster as orr

    ds = syn.make_object_dataset(tmp, n_objects=1, seed=0)
    meshes = {o.label: mesh_io.load_rigid_object(o) for o in ds.list_objects}
    db = MeshDataBase.from_object_ds(ds).batched()
    renderer = orr.OracleBatchRenderer(meshes)
    preds = {}
    for role, seed in (("coarse", 11), ("refiner", 12)):
        cfg = syn.make_cfg(role)
        head, n_out = ("pose", 9) if role == "refiner" else ("logits", 1)
        preds[role] = op.OraclePosePredictor(cfg, syn.make_state_dict("vanilla_resnet34", syn.n_inputs_for(cfg), head, n_out, seed=seed),
                                             db.labels.tolist(), db.points, renderer)
    oest = op.OraclePoseEstimator(preds["coarse"], preds["refiner"], load_SO3_grid(72), bsz=24)
    infos = pd.DataFrame(dict(label=[o.label for o in ds.list_objects], batch_im_id=0, instance_id=[0]))
    res = oest.run(obs.images.cpu(), obs.K.cpu(), infos, det.bboxes.cpu(), n_refiner_iterations=1, n_pose_hypotheses=2)
    lg = extra["coarse"]["data"]["logits"].flatten().cpu()
    scale = max(1.0, res["coarse_logits"].abs().max().item())
    err_l = (lg - res["coarse_logits"]).abs().max().item()
    err_p = (final.poses.cpu() - res["final_TCO"]).abs().max().item()
    print(f"[smoke] {arch} ({n_cu} CUs): coarse logit err {err_l:.2e} (scale {scale:.1f}), final pose err {err_p:.2e}, {extra['timing_str']}")
    assert err_l < 1e-4 * scale, err_l
    assert err_p < 1e-4, err_p
