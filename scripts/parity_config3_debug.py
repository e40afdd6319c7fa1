"""GPU box: the sampled-rows parity of BASELINE configs[2] (RGBD, WideResNet-34 refiner, 8 x 576 x 5) with the stem-record path on and off,
every figure printed (tests/test_gpu_parity_full_size.py asserts the same call).  python scripts/parity_config3_debug.py [backbone]"""
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

backbone = sys.argv[1] if len(sys.argv) > 1 else "resnet34"
for records in (True, False):
    os.environ["MP_STEM_RECORDS"] = "1" if records else "0"
    tmp = tempfile.mkdtemp(prefix="mp_p3_")
    est, obs, det, _ = make_scene(n_objects=8, seed=7, backbone=backbone, rgbd=True, SO3_grid_size=576, tmp_dir=tmp)
    est.refiner_model.stem_records = records
    est.coarse_model.stem_records = records
    final, extra = est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=5, n_pose_hypotheses=576)
    ds = syn.make_object_dataset(tmp, n_objects=8, seed=7)
    oest, db = harness.make_oracle_estimator(ds, 576, backbone=backbone, rgbd=True, bsz=16)
    res = harness.sampled_rows_parity(oest, db, obs.images.cpu(), obs.K.cpu(), det.bboxes.cpu(), extra,
                                      coarse_rows=[5, 576 + 200, 3 * 576 + 575, 7 * 576 + 1],
                                      refine_rows=[3, 2 * 576 + 17, 5 * 576 + 300, 8 * 576 - 1, 100, 576 + 9, 4 * 576 + 400, 6 * 576 + 77], n_iterations=5)
    print("RECORDS" if records else "FP32 TENSOR", backbone, json.dumps({k: v for k, v in res.items()}))
    del est
    torch.cuda.empty_cache()
