"""CPU: the row-sampling parity harness (oracle/harness.py) indexes the pipeline tables correctly: fed with an `extra_data`
structure built from the ORACLE's own full run it must report zero error on every sampled row."""
import tempfile
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pandas as pd
import torch

GOLD = Path(__file__).resolve().parent / "golden"


def test_oracle_fp16_renders_mode_rounds_the_cnn_input_through_binary16():
    """oracle side of the engine's "fp16 renders" mode (OraclePosePredictor.input_f16): the CNN input is the fp32 one rounded to
    nearest-even binary16 (depth channels: rounded as rendered, normalised, rounded again), everything else is unchanged"""
    from tests.support import synthetic as syn
    from oracle import harness

    g = {k: v for k, v in np.load(GOLD / "pipeline.npz").items()}
    images = (torch.from_numpy(g["img_u8"]).float() / 255).permute(2, 0, 1)[None]
    K = torch.from_numpy(g["K"]).float().reshape(-1, 3, 3)
    T = torch.from_numpy(g["gt_TCO"][:1]).float()
    ds = syn.make_object_dataset(tempfile.mkdtemp(prefix="mp_h16_"), n_objects=1, seed=0)
    coarse, refiner, db = harness.make_oracle_models(ds)
    im = torch.zeros(1, dtype=torch.long)
    a = coarse.step(images, im, K, [ds[0].label], T)
    coarse.input_f16 = True
    b = coarse.step(images, im, K, [ds[0].label], T)
    assert torch.equal(a["x"].half().float(), b["x"]) and not torch.equal(a["x"], b["x"])
    assert torch.equal(a["K_crop"], b["K_crop"]) and torch.equal(a["TCV_O"], b["TCV_O"])
    la, lb = a["net"]["renderings_logits"], b["net"]["renderings_logits"]
    assert la.shape == lb.shape and 0 < (la - lb).abs().max().item() < 1e-2 * max(1.0, la.abs().max().item())
    # RGBD refiner: normalised depth channels are rounded twice
    ds2 = syn.make_object_dataset(tempfile.mkdtemp(prefix="mp_h16d_"), n_objects=1, seed=0)
    _, ref_d, _ = harness.make_oracle_models(ds2, rgbd=True)
    depth = torch.full((1, 1, *images.shape[-2:]), 0.6)
    rgbd = torch.cat([images, depth], 1)
    ra = ref_d.step(rgbd, im, K, [ds2[0].label], T)
    ref_d.input_f16 = True
    rb = ref_d.step(rgbd, im, K, [ds2[0].label], T)
    assert torch.equal(rb["x"], rb["x"].half().float())
    assert (ra["x"] - rb["x"]).abs().max().item() < 2e-3   # |x| <= 2 (clamped depth): half an ulp of binary16 at 2.0 is 4.9e-4, twice


def test_sampled_rows_parity_is_exact_on_the_oracles_own_run():
    from tests.support import synthetic as syn
    from oracle import harness

    g = {k: v for k, v in np.load(GOLD / "pipeline.npz").items()}
    images = (torch.from_numpy(g["img_u8"]).float() / 255).permute(2, 0, 1)[None]
    K = torch.from_numpy(g["K"]).float().reshape(-1, 3, 3)
    bboxes = torch.from_numpy(g["bboxes"]).float()
    ds = syn.make_object_dataset(tempfile.mkdtemp(prefix="mp_h_"), n_objects=1, seed=0)
    oest, db = harness.make_oracle_estimator(ds, 72, bsz=8)
    infos = pd.DataFrame(dict(label=[ds[0].label], batch_im_id=[0], instance_id=[0]))
    n_it = 1
    res = oest.run(images, K, infos, bboxes, n_refiner_iterations=n_it, n_pose_hypotheses=2, max_coarse_rows=16)
    # the HIP pipeline's extra_data layout, filled with the oracle's values
    dfc = res["coarse_infos"]
    extra = {
        "coarse": {"preds": SimpleNamespace(infos=dfc, poses=res["coarse_TCO"]), "data": {"logits": res["coarse_logits"].reshape(1, -1)}},
        "coarse_filter": {"preds": SimpleNamespace(infos=res["filtered_infos"])},
        "refiner_all_hypotheses": {
            "preds": {f"iteration={n + 1}": SimpleNamespace(poses=res["refiner_poses"][n]) for n in range(n_it)},
            "data": {"pose_outputs": {f"iteration={n + 1}": res["refiner_pose_out"][n] for n in range(n_it)}}},
        "scoring": {"data": {"logits": res["scoring_logits"].reshape(-1, 1)}},
    }
    out = harness.sampled_rows_parity(oest, db, images, K, bboxes, extra, coarse_rows=[0, 7, 15], refine_rows=[0, 1], n_iterations=n_it)
    assert out["coarse_TCO_max_err"] == 0.0 and out["coarse_logit_max_err"] < 2e-6 * out["logit_scale"], out
    assert max(out["pose_max_err_per_iter"]) < 1e-6 and max(out["pose_out_max_err_per_iter"]) < 1e-6, out
    assert out["score_logit_max_err"] < 2e-6 * out["logit_scale"], out
    assert harness.parity_ok(out)
    # and it must notice a wrong row: shift the refined poses by one row
    extra["refiner_all_hypotheses"]["preds"]["iteration=1"] = SimpleNamespace(poses=res["refiner_poses"][0].flip(0))
    bad = harness.sampled_rows_parity(oest, db, images, K, bboxes, extra, coarse_rows=[], refine_rows=[0, 1], n_iterations=n_it)
    assert not harness.parity_ok(bad)


def test_chained_logit_rules_strict_by_default_and_capped_when_opted_in():
    """the strict flip rule gates the chained score logits by default; the pose-aware rule is an explicit opt-in and capped."""
    from oracle import harness

    base = {"logit_scale": 1.0, "coarse_TCO_max_err": 0.0, "pose_max_err_per_iter": [5e-5], "score_logit_errs_teacher_forced": [1e-6] * 8}
    drift = dict(base, score_logit_errs=[2.0e-4] + [1e-5] * 7, final_pose_errs=[5.6e-5] + [1e-6] * 7)   # the round-5 RGBD figures
    assert not harness.parity_ok(drift)                          # 2.0e-4 is not < 2 x tol: the default gate refuses it
    assert harness.parity_ok(drift, chained="pose_aware")        # ... the opt-in rule carries the row's pose difference (2e-4 + 5.6e-4, capped at 3e-4)
    r = harness.chained_score_rule(drift)
    assert r["ok"] and abs(r["cap"] - 3e-4) < 1e-12
    big = dict(base, score_logit_errs=[3.5e-4] + [1e-5] * 7, final_pose_errs=[9e-5] + [1e-6] * 7)
    assert not harness.parity_ok(big, chained="pose_aware")      # over the cap, whatever the pose difference
    fine = dict(base, score_logit_errs=[3e-5] * 8, final_pose_errs=[2e-6] * 8)                           # the headline config's figures
    assert harness.parity_ok(fine) and harness.parity_ok(fine, chained="pose_aware")
