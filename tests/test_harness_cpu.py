"""CPU: the row-sampling parity harness (oracle/harness.py) indexes the pipeline tables correctly: fed with an `extra_data`
structure built from the ORACLE's own full run it must report zero error on every sampled row."""
import tempfile
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pandas as pd
import torch

GOLD = Path(__file__).resolve().parent / "golden"


def test_sampled_rows_parity_is_exact_on_the_oracles_own_run():
    from megapose6d_amd import synthetic as syn
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
