"""-m gpu: raw frames -> detector -> pose pipeline through the reference-shaped API (`run_inference_pipeline(run_detector=True)`,
reference inference/pose_estimator.py:553-562): the HIP Mask R-CNN behind `Detector` feeds the coarse / refine / score stages without
leaving the device.  The detector carries random weights (no checkpoint is available offline), so only the plumbing is asserted:
labels map onto the mesh database, every detection of the filtered set gets exactly one pose estimate, shapes and columns are the
reference's.  (Sorted last: written without GPU access.)"""
import tempfile
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_run_inference_pipeline_with_the_hip_detector():
    from tests.support import synthetic as syn
    from megapose6d_amd.detector import Detector
    from megapose6d_amd.mask_rcnn import DetectorMaskRCNN
    from tests.support.scene import make_scene
    from oracle import mask_rcnn as om

    tmp = tempfile.mkdtemp(prefix="mp_tdp_")
    est, obs, det_gt, gt = make_scene(n_objects=2, seed=4, SO3_grid_size=72, tmp_dir=tmp)
    labels = det_gt.infos["label"].tolist()
    C = len(labels) + 1
    m = DetectorMaskRCNN(input_resize=(480, 640), n_classes=C)
    m.load_state_dict(om.synthetic_state_dict(C))
    m.engine_overrides = {"box_detections_per_img": 16}
    m = m.cuda().eval()
    m.config = SimpleNamespace(label_to_category_id={l: i + 1 for i, l in enumerate(labels)})
    est.detector_model = Detector(m)
    dets = est.forward_detection_model(obs, one_instance_per_class=True)
    assert len(dets) >= 1 and set(dets.infos["label"]) <= set(labels) and dets.bboxes.shape == (len(dets), 4) and dets.bboxes.is_cuda
    assert (dets.bboxes[:, 2] >= dets.bboxes[:, 0]).all() and float(dets.bboxes.min()) >= 0 and float(dets.bboxes[:, 2].max()) <= 640
    final, extra = est.run_inference_pipeline(obs, run_detector=True, n_refiner_iterations=1, n_pose_hypotheses=1)
    torch.cuda.synchronize()
    assert "detection=" in extra["timing_str"]
    n_det = len(extra["coarse"]["preds"]) // 72
    assert 1 <= n_det <= 16 and len(final) == n_det and final.poses.shape == (n_det, 4, 4)
    assert set(final.infos.columns) >= {"batch_im_id", "label", "instance_id", "pose_score", "pose_logit"}
    assert set(final.infos["label"]) <= set(labels)
