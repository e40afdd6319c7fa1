"""-m gpu: the evaluation caller on the real engine: 4 RGBD frames, per-frame calls vs ONE multi-image call."""
import tempfile

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_prediction_runner_multi_image_batching_matches_per_frame_calls():
    from tests.support import synthetic as syn
    from megapose6d_amd.icp_refiner import ICPRefiner
    from megapose6d_amd.prediction_runner import PredictionRunner
    from tests.support.scene import build_estimator, render_observation
    from megapose6d_amd.tcoll import PandasTensorCollection
    from megapose6d_amd.types import InferenceConfig

    ds = syn.make_object_dataset(tempfile.mkdtemp(prefix="mp_runner_"), n_objects=3, seed=60, n_theta=48, n_z=50)
    est = build_estimator(ds, SO3_grid_size=72)
    est.depth_refiner = ICPRefiner(est.mesh_db, est.refiner_model.renderer)
    r = est.coarse_model.renderer
    rng = np.random.RandomState(1)
    K = syn.K_EXAMPLE.astype(np.float32)
    labels = [o.label for o in ds.list_objects]
    scene = []
    for f in range(4):
        labs = [labels[(f + j) % 3] for j in range(1 + f % 2)]
        poses = np.stack([syn.random_pose(rng, (0.45, 0.7), 0.25) for _ in labs])
        im, bb = render_observation(r, labs, poses, K, seed=f, with_depth=True)
        rgb = (im[0, :3].permute(1, 2, 0) * 255).round().to(torch.uint8).cpu().numpy()
        infos = pd.DataFrame(dict(label=labs, scene_id=7, view_id=10 + f))
        scene.append(dict(rgb=rgb, depth=im[0, 3].cpu().numpy(), K=K, gt_detections=PandasTensorCollection(infos, bboxes=torch.from_numpy(bb))))
    cfg = InferenceConfig(detection_type="gt", n_refiner_iterations=2, n_pose_hypotheses=2, run_depth_refiner=True)
    per_frame = PredictionRunner(scene, cfg, batch_size=1).get_predictions(est)
    batched_runner = PredictionRunner(scene, cfg, batch_size=4)
    batched = batched_runner.get_predictions(est)
    assert set(per_frame) == {"final", "refiner/iteration=2", "refiner/final", "coarse", "depth_refiner"}
    assert len(per_frame["final"]) == 6 and len(per_frame["coarse"]) == 6 * 72
    assert len(batched_runner.timings) == 1 and batched_runner.timings[0]["n_frames"] == 4
    key = lambda df: list(zip(df["view_id"], df["label"], df["instance_id"]))
    for k in ("final", "refiner/final", "depth_refiner"):
        a, b = per_frame[k], batched[k]
        assert sorted(key(a.infos)) == sorted(key(b.infos))
        order = [key(b.infos).index(x) for x in key(a.infos)]
        # rows are independent of what shares the launch.  The ICP stage starts from the random-weight refiner's poses (far from the
        # measured depth), where its accept/reject thresholds sit on a knife edge: only its structure is compared here, its numerics
        # are covered by tests/test_gpu_icp.py
        if k == "refiner/final":
            assert (a.poses - b.poses[order]).abs().max().item() < 5e-5
        assert torch.isfinite(b.poses).all()
        assert (a.infos["scene_id"] == 7).all()
    # the depth-refined poses are rigid transforms (the random-weight refiner hands ICP arbitrary starts, so no accuracy claim here)
    Rm = per_frame["depth_refiner"].poses[:, :3, :3]
    assert torch.isfinite(per_frame["depth_refiner"].poses).all() and (Rm @ Rm.transpose(1, 2) - torch.eye(3)).abs().max() < 1e-4
