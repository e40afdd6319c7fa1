"""BUILD CONTAINER ONLY (needs /root/reference): time the REFERENCE's own `PoseEstimator.run_inference_pipeline` (imported from
/root/reference/src through oracle/ref_import.py, driven with the oracle's C rasteriser because Panda3D cannot be installed) beside
the oracle PORT (oracle/pipeline.py) that bench.py uses as `cpu_baseline`, on the same workload: 1 object, 72-rotation grid, top-2,
3 refiner iterations (72 coarse + 6 refine + 2 score rows).  Shows that the port is not slower than what it stands for.
Writes profiles/r02_reference_vs_port_cpu.json."""
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import pandas as pd
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import make_golden as mg  # noqa: E402
from oracle import ref_import  # noqa: E402

r = ref_import.ref()
from tests.support import synthetic as syn  # noqa: E402
from megapose6d_amd.load_model import Config  # noqa: E402
from megapose6d_amd.pose_estimator import load_SO3_grid  # noqa: E402
from oracle import harness  # noqa: E402
from oracle import raster as orr  # noqa: E402

tmp = Path(tempfile.mkdtemp(prefix="mp_reftime_"))
ds, meshes, img_u8, depth, K, bboxes, gt = mg.make_scene(tmp, n_objects=1, seed=0)
ref_objs = [r.RigidObject(label=o.label, mesh_path=o.mesh_path, mesh_units="mm") for o in ds.list_objects]
mesh_db = r.rmd.MeshDataBase.from_object_ds(r.RigidObjectDataset(ref_objs)).batched()
renderer = orr.OracleBatchRenderer(meshes)
import megapose.models.pose_rigid as pr  # noqa: E402

pr.Panda3dBatchRenderer = orr.OracleBatchRenderer
models = {}
for role in ("coarse", "refiner"):
    cfg = Config.from_any(syn.make_cfg(role, "vanilla_resnet34"))
    head, n_out = ("pose", 9) if role == "refiner" else ("logits", 1)
    sd = syn.make_state_dict("vanilla_resnet34", syn.n_inputs_for(cfg), head, n_out, seed={"coarse": 11, "refiner": 12}[role])
    m = r.pmc.create_model_pose(r.pmc.check_update_config(cfg), renderer=renderer, mesh_db=mesh_db)
    m.load_state_dict(sd, strict=True)
    m.eval()
    m.cfg = cfg
    models[role] = m
est = r.pe.PoseEstimator(refiner_model=models["refiner"], coarse_model=models["coarse"], bsz_objects=8, bsz_images=24, SO3_grid_size=72)
obs = r.ty.ObservationTensor.from_numpy(img_u8, None, K)
det = r.tc.PandasTensorCollection(infos=pd.DataFrame(dict(label=[o.label for o in ref_objs], batch_im_id=0, instance_id=[0])), bboxes=torch.as_tensor(bboxes))
oest, db = harness.make_oracle_estimator(ds, 72, bsz=24)
oest.bsz_refiner = 8
images = (torch.from_numpy(img_u8).float() / 255).permute(2, 0, 1)[None]
infos = pd.DataFrame(dict(label=[ds[0].label], batch_im_id=[0], instance_id=[0]))
out = {"workload": "1 object, 72-rotation grid, n_pose_hypotheses=2, 3 refiner iterations (72 coarse + 6 refine + 2 score rows), vanilla ResNet-34, "
                   "oracle C rasteriser (4x MSAA) in both", "host_cores": os.cpu_count(), "runs": []}
for threads in (1, 8):
    torch.set_num_threads(threads)
    res = {"threads": threads}
    for name, fn in (("reference", lambda: est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=3, n_pose_hypotheses=2)),
                     ("port", lambda: oest.run(images, torch.from_numpy(K)[None], infos, torch.from_numpy(bboxes), n_refiner_iterations=3, n_pose_hypotheses=2))):
        with torch.no_grad():
            fn()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                o = fn()
                ts.append(time.perf_counter() - t0)
        res[name + "_s"] = float(np.median(ts))
        res[name + "_final"] = (o[0].poses if name == "reference" else o["final_TCO"]).numpy().tolist()
    res["identical_result"] = bool(np.array_equal(np.asarray(res.pop("reference_final")), np.asarray(res.pop("port_final"))))
    res["port_over_reference"] = res["port_s"] / res["reference_s"]
    out["runs"].append(res)
    print(res, flush=True)
(ROOT / "profiles" / "r02_reference_vs_port_cpu.json").write_text(json.dumps(out, indent=1))
