"""GPU box: chained refiner parity on the HEADLINE config (BASELINE configs[1]: RGB, vanilla ResNet-34, 1 object x 576 hypotheses x 5 iterations)
as a function of how strongly the pose head passes a conv-stack error on to the pose, and of the kernel family that computes the 3x3 / stride-1
convolutions (round-5 verdict, item 4).

    python scripts/parity_undamped_config2.py <pose_head_scale> [rows]        (kernel family from the environment: MP_CONV_WINO = 2 | 1 | 0)

pose_head_scale multiplies pose_fc's weights (tests/support/synthetic.py: 0.05 = the seeded default of the tests and of bench.py -- one iteration
moves a pose by ~1e-2; 1.0 = "weights x 1", bias = identity update).  The HIP pipeline runs the whole 576-row call; the oracle recomputes the
sampled refiner chains (oracle.harness.sampled_rows_parity) from the oracle's own initial poses: every figure is CHAINED over the 5 iterations.
Prints one JSON line: per-iteration pose error, the size of the pose updates themselves (how un-damped the head is), and whether the chain stayed
in front of the camera."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import harness  # noqa: E402
from tests.support import synthetic as syn  # noqa: E402
from tests.support.scene import make_scene  # noqa: E402

scale = float(sys.argv[1])
rows = [int(r) for r in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 191, 320, 575]
tmp = tempfile.mkdtemp(prefix="mp_und_")
est, obs, det, _ = make_scene(n_objects=1, seed=0, SO3_grid_size=576, tmp_dir=tmp, pose_head_scale=scale)
final, extra = est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=5, n_pose_hypotheses=576)
torch.cuda.synchronize()
preds = extra["refiner_all_hypotheses"]["preds"]
P = [preds[f"iteration={n}"].poses.cpu() for n in range(1, 6)]
P0 = preds["iteration=1"].poses_input.cpu()
step = [(P[0] - P0).abs().flatten(1).max(1).values] + [(P[n] - P[n - 1]).abs().flatten(1).max(1).values for n in range(1, 5)]
z = torch.stack([p[:, 2, 3] for p in P])
ds = syn.make_object_dataset(tmp, n_objects=1, seed=0)
oest, db = harness.make_oracle_estimator(ds, 576, bsz=16, pose_head_scale=scale)
res = harness.sampled_rows_parity(oest, db, obs.images.cpu(), obs.K.cpu(), det.bboxes.cpu(), extra, coarse_rows=[0, 97, 383, 575], refine_rows=rows,
                                  n_iterations=5)
out = {"pose_head_scale": scale, "MP_CONV_WINO": os.environ.get("MP_CONV_WINO", "2 (default: bf16x9 exact pieces)"), "rows": rows,
       "pose_update_size_per_iter_median_max": [[round(float(s.median()), 5), round(float(s.max()), 5)] for s in step],
       "z_min_max_over_chains": [round(float(z.min()), 4), round(float(z.max()), 4)],
       "all_finite": bool(all(torch.isfinite(p).all() for p in P)),
       "pose_max_err_per_iter": res["pose_max_err_per_iter"], "pose_out_max_err_per_iter": res["pose_out_max_err_per_iter"],
       "final_pose_max_err": res["final_pose_max_err"], "score_logit_max_err_chained": res["score_logit_max_err"],
       "score_logit_max_err_teacher_forced": max(res["score_logit_errs_teacher_forced"]), "coarse_logit_max_err": res.get("coarse_logit_max_err"),
       "logit_scale": res["logit_scale"], "feature_max": res.get("feature_max")}
print("UNDAMPED", json.dumps(out))
