"""-m gpu, sorted last: the two end-to-end comparisons of this round's new modes that run a full CPU oracle beside the GPU
(tens of seconds of host work each) and could not be executed on hardware in the session that wrote them (its GPU budget was
spent; every other new GPU test of the round did run -- profiles/r02_new_gpu_tests.txt):
  * the detector network against a FRESH oracle run (not the committed goldens) + the reference-shaped Detector wrapper on top;
  * the whole pose pipeline in the "fp16 renders" mode against the oracle with the same binary16 rounding of its CNN input."""
import tempfile
from types import SimpleNamespace

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import assert_logits_close  # noqa: E402
from test_gpu_zz_detector import _match, _model  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene72():
    from tests.support import synthetic as syn
    from tests.support.scene import make_scene

    tmp = tempfile.mkdtemp(prefix="mp_t16b_")
    est, obs, det, gt = make_scene(n_objects=1, seed=0, SO3_grid_size=72, tmp_dir=tmp)
    ds = syn.make_object_dataset(tmp, n_objects=1, seed=0)
    return ds, est, obs, det, gt


def test_detector_vs_a_fresh_oracle_run_and_through_the_detector_wrapper():
    """small frame (128 x 160, 3 classes), full oracle on the host: detections, pasted masks, and the reference-shaped wrapper"""
    from megapose6d_amd.detector import Detector
    from megapose6d_amd.types import ObservationTensor
    from oracle import mask_rcnn as om

    C, H, W = 3, 128, 160
    torch.set_num_threads(16)
    images = om.synthetic_images(2, H, W)
    ref = om.mask_rcnn_forward(om.synthetic_state_dict(C), list(images), H, W)
    m = _model(C, H, W)
    out = m(list(images.cuda()))
    for o, r in zip(out, ref):
        k = len(r["boxes"])
        ob, os_, ol = o["boxes"].cpu().numpy(), o["scores"].cpu().numpy(), o["labels"].cpu().numpy()
        assert _match(ob, os_, ol, r["boxes"].numpy(), r["scores"].numpy(), r["labels"].numpy()) >= k - 3 - k // 20
        for j in range(min(k, 10)):   # pasted masks of detections that sit at the same place in both lists
            if j < len(ob) and ol[j] == int(r["labels"][j]) and np.abs(ob[j] - r["boxes"][j].numpy()).max() < 1e-3:
                assert (o["masks"][j, 0].cpu() - r["masks"][j, 0]).abs().max().item() < 5e-2
    m.config = SimpleNamespace(label_to_category_id={"ds-obj_000001": 1, "ds-obj_000002": 2})
    det = Detector(m)
    d = det.get_detections(ObservationTensor(images=images.cuda()), output_masks=True, detection_th=0.3)
    assert set(d.infos.columns) >= {"batch_im_id", "label", "score", "instance_id"} and (d.infos["score"] > 0.3).all()
    assert d.bboxes.shape == (len(d), 4) and d.masks.shape == (len(d), H, W) and d.masks.dtype == torch.bool
    assert set(d.infos["label"]) <= {"ds-obj_000001", "ds-obj_000002"}
    one = det.get_detections(ObservationTensor(images=images.cuda()), one_instance_per_class=True)
    assert one.infos.groupby(["batch_im_id", "label"]).size().max() == 1


def test_pipeline_in_fp16_renders_mode_vs_oracle(scene72):
    """72-rotation grid, top-2, 3 refiner iterations: HIP pipeline with render_dtype=float16 vs the oracle with the same rounding
    of its CNN input.  Tolerances of test_gpu_pipeline.py (poses 1e-4, logits 1e-4 / 5e-4 of their scale)."""
    from oracle import harness

    ds, est, obs, det, gt = scene72
    oest, db = harness.make_oracle_estimator(ds, 72)
    oest.coarse.input_f16 = oest.refiner.input_f16 = True
    infos = pd.DataFrame(dict(label=[o.label for o in ds.list_objects], batch_im_id=0, instance_id=[0]))
    est.render_dtype = torch.float16
    try:
        assert est.coarse_model.render_dtype == torch.float16 and est.refiner_model.render_dtype == torch.float16
        final, extra = est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=3, n_pose_hypotheses=2)
        f32_final = None
    finally:
        est.render_dtype = torch.float32
    f32_final, _ = est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=3, n_pose_hypotheses=2)
    res = oest.run(obs.images.cpu(), obs.K.cpu(), infos, det.bboxes.cpu(), n_refiner_iterations=3, n_pose_hypotheses=2)
    lg = extra["coarse"]["data"]["logits"].flatten().cpu().numpy()
    scale = max(1.0, float(res["coarse_logits"].abs().max()))
    assert_logits_close(lg, res["coarse_logits"].numpy(), scale)
    hyp = extra["coarse_filter"]["preds"].infos["hypothesis_id"].tolist()
    ohyp = res["filtered_infos"]["hypothesis_id"].tolist()
    assert sorted(hyp) == sorted(ohyp)
    order = [hyp.index(h) for h in ohyp]
    for n in range(3):
        p = extra["refiner_all_hypotheses"]["preds"][f"iteration={n + 1}"].poses.cpu()[order]
        assert (p - res["refiner_poses"][n]).abs().max().item() < 1e-4, n
    assert_logits_close(extra["scoring"]["data"]["logits"].flatten().cpu().numpy()[order], res["scoring_logits"].numpy(), scale)
    assert (final.poses.cpu() - res["final_TCO"]).abs().max().item() < 1e-4
    # the mode is a (small) deviation from the fp32 reference arithmetic, never a silent no-op and never a different answer
    d = (final.poses - f32_final.poses).abs().max().item()
    assert d < 5e-3


def test_load_named_model_from_run_directories_equals_the_direct_api(tmp_path, monkeypatch):
    """`load_named_model` (reference utils/load_model.py:50-97) on the two run directories it reads (config.yaml + checkpoint.pth.tar,
    written in the reference layout): the estimator it assembles gives the poses of one assembled directly from the same weights."""
    from megapose6d_amd import load_model as lm
    from tests.support import synthetic as syn
    from tests.support.scene import make_scene

    est, obs, det, gt = make_scene(n_objects=1, seed=0, tmp_dir=str(tmp_path / "scene"))
    ds = syn.make_object_dataset(tmp_path / "scene", n_objects=1, seed=0)
    info = lm.NAMED_MODELS["megapose-1.0-RGB-multi-hypothesis"]
    for role, seed, run_id in (("coarse", 11, info["coarse_run_id"]), ("refiner", 12, info["refiner_run_id"])):
        cfg = syn.make_cfg(role)
        head, n_out = ("pose", 9) if role == "refiner" else ("logits", 1)
        lm.save_run(tmp_path / "megapose-models" / run_id, cfg, syn.make_state_dict("vanilla_resnet34", syn.n_inputs_for(cfg), head, n_out, seed=seed))
    monkeypatch.setattr(lm, "LOCAL_DATA_DIR", tmp_path)
    named = lm.load_named_model("megapose-1.0-RGB-multi-hypothesis", ds)
    f_named, _ = named.run_inference_pipeline(obs, detections=det, **info["inference_parameters"])
    f_direct, _ = est.run_inference_pipeline(obs, detections=det, **info["inference_parameters"])
    assert np.abs(f_named.poses.cpu().numpy() - f_direct.poses.cpu().numpy()).max() < 1e-6
