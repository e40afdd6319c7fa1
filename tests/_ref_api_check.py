"""Run in a SUBPROCESS by tests/test_reference_api_cpu.py (build container only: needs /root/reference).
(i) diffs every public signature of the drop-in classes against the imported reference, (ii) executes the monkey-patch
block of INTEGRATION.md verbatim and checks that each patched attribute now resolves to the engine's object.
Prints a JSON list of problems (empty = OK)."""
import inspect
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_import  # noqa: E402

ref_import.install()

import megapose.datasets.object_dataset as r_od  # noqa: E402
import megapose.inference.depth_refiner as r_dr  # noqa: E402
import megapose.inference.icp_refiner as r_icp  # noqa: E402
import megapose.inference.pose_estimator as r_pe  # noqa: E402
import megapose.inference.utils as r_iu  # noqa: E402
import megapose.models.pose_rigid as r_pr  # noqa: E402
import megapose.panda3d_renderer.panda3d_batch_renderer as r_pbr  # noqa: E402
import megapose.training.pose_models_cfg as r_pmc  # noqa: E402
import megapose.utils.load_model as r_lm  # noqa: E402

import megapose6d_amd as mp  # noqa: E402

problems = []


def params(fn):
    return [p for p in inspect.signature(fn).parameters.values() if p.name != "self"]


def diff(name, ref_fn, our_fn):
    rp, op = params(ref_fn), params(our_fn)
    for i, r in enumerate(rp):
        if r.kind in (inspect.Parameter.VAR_KEYWORD, inspect.Parameter.VAR_POSITIONAL):
            if not any(o.kind == r.kind for o in op):
                problems.append(f"{name}: missing *{r.name}")
            continue
        if i >= len(op) or op[i].name != r.name:
            problems.append(f"{name}: parameter {i} is {op[i].name if i < len(op) else None!r}, reference has {r.name!r}")
            continue
        if r.default is not inspect.Parameter.empty and op[i].default != r.default:
            problems.append(f"{name}: default of {r.name!r} is {op[i].default!r}, reference {r.default!r}")
        if r.default is inspect.Parameter.empty and op[i].default is not inspect.Parameter.empty:
            pass  # a default where the reference has none is a superset
    fixed = [r for r in rp if r.kind not in (inspect.Parameter.VAR_KEYWORD, inspect.Parameter.VAR_POSITIONAL)]
    for o in op[len(fixed):]:  # engine extensions: must be optional
        if o.kind in (inspect.Parameter.VAR_KEYWORD, inspect.Parameter.VAR_POSITIONAL):
            continue
        if o.default is inspect.Parameter.empty:
            problems.append(f"{name}: extra parameter {o.name!r} has no default")


PAIRS = [
    (r_pe.PoseEstimator, mp.pose_estimator.PoseEstimator,
     ["__init__", "load_SO3_grid", "forward_refiner", "forward_scoring_model", "forward_coarse_model", "forward_detection_model",
      "run_depth_refiner", "run_inference_pipeline", "filter_pose_estimates"]),
    (r_pr.PosePredictor, mp.pose_rigid.PosePredictor,
     ["__init__", "crop_inputs", "compute_crops_multiview", "update_pose", "net_forward", "render_images_multiview", "normalize_images",
      "normalize_depth", "forward", "forward_coarse_tensor", "forward_coarse"]),
    (r_pbr.Panda3dBatchRenderer, mp.renderer.Panda3dBatchRenderer, ["__init__", "render", "stop"]),
    (r_icp.ICPRefiner, mp.icp_refiner.ICPRefiner, ["__init__", "refine_poses"]),
    (r_dr.DepthRefiner, mp.icp_refiner.DepthRefiner, ["refine_poses"]),
    (r_od.RigidObject, mp.object_dataset.RigidObject, ["__init__", "make_symmetry_poses"]),
    (r_od.RigidObjectDataset, mp.object_dataset.RigidObjectDataset, ["__init__", "__getitem__", "get_object_by_label", "__len__", "filter_objects"]),
]
for rc, oc, methods in PAIRS:
    for m in methods:
        if not hasattr(oc, m):
            problems.append(f"{oc.__name__}.{m}: missing")
            continue
        diff(f"{oc.__name__}.{m}", getattr(rc, m), getattr(oc, m))
for name, rf, of in [("create_model_pose", r_pmc.create_model_pose, mp.load_model.create_model_pose),
                     ("load_named_model", r_lm.load_named_model, mp.load_model.load_named_model),
                     ("load_pose_models", r_iu.load_pose_models, mp.load_model.load_pose_models)]:
    diff(name, rf, of)

# (i-b) Detector (row f-4): signatures + BEHAVIOUR of get_detections against the reference class driven by the same fake model
import megapose.inference.detector as r_det  # noqa: E402

from megapose6d_amd import detector as o_det  # noqa: E402

for m in ["__init__", "image_tensor_from_numpy", "get_detections"]:
    diff(f"Detector.{m}", getattr(r_det.Detector, m), getattr(o_det.Detector, m))


def _detector_behaviour():
    import numpy as np
    import torch
    from types import SimpleNamespace

    class FakeMaskRCNN(torch.nn.Module):
        """deterministic stand-in with torchvision's Mask R-CNN output format"""

        def __init__(self, per_image):
            super().__init__()
            self.config = SimpleNamespace(label_to_category_id={"ycbv-obj_000001": 1, "ycbv-obj_000002": 2, "ycbv-obj_000005": 3})
            self.per_image = per_image

        def forward(self, images):
            outs = []
            for n, im in enumerate(images):
                k = self.per_image[n]
                g = torch.Generator().manual_seed(100 + n)
                xy = torch.rand(k, 2, generator=g) * 200
                wh = torch.rand(k, 2, generator=g) * 100 + 5
                outs.append(dict(boxes=torch.cat([xy, xy + wh], 1), labels=torch.randint(1, 4, (k,), generator=g),
                                 scores=torch.rand(k, generator=g), masks=torch.rand(k, 1, *im.shape[-2:], generator=g)))
            return outs

    rng = np.random.RandomState(0)
    images = torch.from_numpy(rng.rand(3, 3, 24, 32).astype(np.float32))
    from megapose.inference.types import ObservationTensor as RObs

    from megapose6d_amd.types import ObservationTensor as OObs

    for per_image in ([4, 0, 6], [0, 0, 0], [1, 1, 1]):
        for kw in (dict(), dict(detection_th=0.4), dict(output_masks=True, mask_th=0.5), dict(one_instance_per_class=True),
                   dict(detection_th=0.3, output_masks=True, one_instance_per_class=True)):
            if sum(per_image) == 0 and (kw.get("one_instance_per_class") or kw.get("detection_th") is not None):
                continue  # the reference itself raises on an empty frame there (groupby/score on empty columns)
            tag = f"Detector.get_detections per_image={per_image} {kw}"
            try:
                a = r_det.Detector(FakeMaskRCNN(per_image)).get_detections(RObs(images=images), **kw)
                b = o_det.Detector(FakeMaskRCNN(per_image)).get_detections(OObs(images=images), **kw)
            except Exception as e:  # noqa: BLE001
                problems.append(f"{tag}: raised {type(e).__name__}: {e}")
                continue
            da, db = a.infos.reset_index(drop=True), b.infos.reset_index(drop=True)
            if len(da) != len(db):
                problems.append(f"{tag}: {len(db)} detections, reference {len(da)}")
                continue
            if len(da) == 0:   # empty result: pandas gives the reference no instance_id column at all; ours keeps the (empty) column
                db = db.drop(columns=["instance_id"], errors="ignore")
            if sorted(da.columns) != sorted(db.columns):
                problems.append(f"{tag}: columns {sorted(db.columns)}, reference {sorted(da.columns)}")
                continue
            for c in da.columns:
                va, vb = da[c].tolist(), db[c].tolist()
                if c == "score":
                    if not np.allclose(va, vb, atol=1e-7):
                        problems.append(f"{tag}: scores differ")
                elif va != vb:
                    problems.append(f"{tag}: column {c}: {vb} vs reference {va}")
            if not torch.equal(a.bboxes.float().cpu(), b.bboxes.cpu()):
                problems.append(f"{tag}: bboxes differ")
            if kw.get("output_masks") and sum(per_image):
                if not torch.equal(a.masks.cpu(), b.masks.cpu()):
                    problems.append(f"{tag}: masks differ")
    d = o_det.Detector(FakeMaskRCNN([1]))
    u8 = (rng.rand(5, 7, 3) * 255).astype(np.uint8)
    if not torch.equal(d.image_tensor_from_numpy(u8), r_det.Detector(FakeMaskRCNN([1])).image_tensor_from_numpy(u8)):
        problems.append("Detector.image_tensor_from_numpy differs")


try:
    _detector_behaviour()
except Exception as e:  # noqa: BLE001
    import traceback

    problems.append("detector behaviour check crashed: " + traceback.format_exc()[-600:])

# (ii) INTEGRATION.md monkey-patch block, executed verbatim
md = (ROOT / "INTEGRATION.md").read_text()
blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
patch = next((b for b in blocks if "sitecustomize" in b), None)
if patch is None:
    problems.append("INTEGRATION.md: monkey-patch block not found")
else:
    code = "\n".join(l[3:] if l.startswith("   ") else l for l in patch.splitlines())
    ns = {}
    try:
        exec(compile(code, "INTEGRATION.md", "exec"), ns)
    except Exception as e:  # noqa: BLE001
        problems.append(f"INTEGRATION.md patch block raised {type(e).__name__}: {e}")
    else:
        import megapose.inference.icp_refiner as icp
        import megapose.inference.pose_estimator as pe
        import megapose.inference.utils as iu
        import megapose.panda3d_renderer.panda3d_batch_renderer as pbr
        import megapose.utils.load_model as lm

        checks = [(pbr, "Panda3dBatchRenderer", mp.renderer.Panda3dBatchRenderer), (iu, "Panda3dBatchRenderer", mp.renderer.Panda3dBatchRenderer),
                  (iu, "create_model_pose", mp.load_model.create_model_pose), (iu, "MeshDataBase", mp.mesh_db.MeshDataBase),
                  (pe, "PoseEstimator", mp.pose_estimator.PoseEstimator), (lm, "PoseEstimator", mp.pose_estimator.PoseEstimator),
                  (icp, "ICPRefiner", mp.icp_refiner.ICPRefiner), (lm, "ICPRefiner", mp.icp_refiner.ICPRefiner)]
        for mod, attr, want in checks:
            if getattr(mod, attr, None) is not want:
                problems.append(f"after the patch {mod.__name__}.{attr} is not the engine's object")
        # every name the patch assigns to must have existed in the reference module beforehand (no typo'd attribute)
        for m_alias, attr in re.findall(r"^\s*(\w+)\.(\w+)\s*=", code, flags=re.M):
            pass
print("REF_API_JSON " + json.dumps(problems))
