"""Pinning recipe for the oracles that restate THIRD-PARTY code absent from this image (SURVEY.md 8c "parity unpinned"):

    python -m oracle.make_thirdparty_fixtures [--out tests/golden]

Run it in an environment where the reference's dependencies exist (the reference's conda env / Dockerfile.megapose: panda3d,
opencv-contrib-python, torchvision 0.12).  For every package that imports it writes one fixture; every package that does not is
skipped with a message.  The CPU tests in tests/test_thirdparty_pinning_cpu.py compare the restatements with whatever fixtures are
present and `pytest.skip("fixture absent")` otherwise.  Nothing here is imported by the product.

  thirdparty_roi_align.npz   torchvision.ops.roi_align(sampling_ratio=4, aligned=False) on seeded inputs (reference call sites
                             lib3d/cropping.py:113-144)                                   -> oracle.thirdparty.roi_align
  thirdparty_maskrcnn.npz    torchvision.models.detection.maskrcnn_resnet50_fpn (the class behind models/mask_rcnn.py:23-46) with the
                             hash-defined synthetic weights of oracle.mask_rcnn, eval mode, on oracle.mask_rcnn.synthetic_images
                             -> boxes / labels / scores / masks of oracle.mask_rcnn.mask_rcnn_forward
  thirdparty_icp.npz         cv2.ppf_match_3d_ICP(100, 0.05, 2.5, 4).registerModelToScene on seeded point clouds with normals
  thirdparty_inpaint.npz     cv2.inpaint(depth, holes, 5, cv2.INPAINT_NS) on a seeded depth map with holes (the un-restated call of get_normal)
                             (reference inference/icp_refiner.py:166-169)                 -> oracle.icp_opencv.opencv_icp
  thirdparty_panda3d.npz     the reference's Panda3dSceneRenderer (panda3d_scene_renderer.py:58-101, 298-358: 4x MSAA, mip-mapping,
                             16x anisotropy) on the lathe test mesh: rgb / normals / depth of a few views
                             -> oracle.raster (per-pixel mismatch STATISTICS are reported, not bit-exactness)
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def roi_align_inputs():
    import torch

    g = torch.Generator().manual_seed(41)
    img = torch.rand(2, 4, 60, 80, generator=g)
    rois = torch.tensor([[0, 5.5, 3.2, 50.1, 40.7], [1, -10.0, -5.0, 70.0, 50.0], [1, 30.0, 20.0, 95.0, 75.0], [0, 10.0, 10.0, 10.4, 10.3]])
    return img, rois


def make_roi_align(out: Path) -> bool:
    try:
        import torchvision
        from torchvision.ops import roi_align
    except Exception as e:  # noqa: BLE001
        print(f"[skip] torchvision not importable ({e})")
        return False
    img, rois = roi_align_inputs()
    res = roi_align(img, rois, output_size=(12, 16), spatial_scale=1.0, sampling_ratio=4)
    np.savez_compressed(out / "thirdparty_roi_align.npz", out=res.numpy(), torchvision_version=str(torchvision.__version__))
    print("thirdparty_roi_align.npz written, torchvision", torchvision.__version__)
    return True


MASKRCNN_CASE = dict(n_classes=5, n=1, h=96, w=128)


def make_maskrcnn(out: Path) -> bool:
    try:
        import torch
        import torchvision
        from torchvision.models.detection import maskrcnn_resnet50_fpn
    except Exception as e:  # noqa: BLE001
        print(f"[skip] torchvision not importable ({e})")
        return False
    from oracle import mask_rcnn as om

    c = MASKRCNN_CASE
    try:
        model = maskrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=c["n_classes"], min_size=c["h"], max_size=c["w"])
    except TypeError:  # newer torchvision: weights= API
        model = maskrcnn_resnet50_fpn(weights=None, weights_backbone=None, num_classes=c["n_classes"], min_size=c["h"], max_size=c["w"])
    sd = om.synthetic_state_dict(c["n_classes"])
    missing = model.load_state_dict(sd, strict=False)
    print("load_state_dict:", missing)
    model.eval()
    with torch.no_grad():
        res = model(list(om.synthetic_images(c["n"], c["h"], c["w"])))
    np.savez_compressed(out / "thirdparty_maskrcnn.npz", boxes=res[0]["boxes"].numpy(), labels=res[0]["labels"].numpy(),
                        scores=res[0]["scores"].numpy(), masks=res[0]["masks"].numpy(), torchvision_version=str(torchvision.__version__),
                        **{k: np.asarray(v) for k, v in c.items()})
    print("thirdparty_maskrcnn.npz written:", len(res[0]["boxes"]), "detections, torchvision", torchvision.__version__)
    return True


def icp_inputs():
    """two noisy samplings of a bumpy surface patch, the scene moved by a small rigid transform; [n,6] float32 (xyz, unit normals)"""
    rng = np.random.RandomState(17)

    def cloud(n, noise):
        u, v = rng.uniform(-0.06, 0.06, n), rng.uniform(-0.05, 0.05, n)
        z = 0.5 + 0.02 * np.sin(40 * u) * np.cos(30 * v) + 0.3 * u * u
        dzu = 0.8 * np.cos(40 * u) * np.cos(30 * v) + 0.6 * u
        dzv = -0.6 * np.sin(40 * u) * np.sin(30 * v)
        nrm = np.stack([-dzu, -dzv, np.ones(n)], 1)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        pts = np.stack([u, v, z], 1) + rng.normal(0, noise, (n, 3))
        return np.concatenate([pts, nrm], 1).astype(np.float32)

    src, dst = cloud(3000, 2e-4), cloud(3500, 2e-4)
    a = 0.03
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) @ np.array([[1, 0, 0], [0, np.cos(0.02), -np.sin(0.02)], [0, np.sin(0.02), np.cos(0.02)]])
    t = np.array([0.004, -0.003, 0.006])
    dst[:, :3] = (dst[:, :3] @ R.T + t).astype(np.float32)
    dst[:, 3:] = (dst[:, 3:] @ R.T).astype(np.float32)
    return src, dst


def make_icp(out: Path) -> bool:
    try:
        import cv2

        icp = cv2.ppf_match_3d_ICP(100, 0.05, 2.5, 4)
    except Exception as e:  # noqa: BLE001
        print(f"[skip] cv2.ppf_match_3d_ICP not available ({e})")
        return False
    src, dst = icp_inputs()
    retval, residual, pose = icp.registerModelToScene(src, dst)
    np.savez_compressed(out / "thirdparty_icp.npz", retval=retval, residual=residual, pose=np.asarray(pose), cv2_version=str(cv2.__version__))
    print("thirdparty_icp.npz written: residual", residual, "cv2", cv2.__version__)
    return True


def inpaint_inputs():
    """a 120 x 160 metric depth map of a tilted, rippled surface with three kinds of holes (specks, a blob on the surface, a border strip)"""
    rng = np.random.RandomState(9)
    v, u = np.mgrid[0:120, 0:160].astype(np.float32)
    d = (0.6 + 0.0009 * u - 0.0006 * v + 0.004 * np.sin(u / 9.0) * np.cos(v / 7.0)).astype(np.float32)
    d[rng.rand(120, 160) < 0.03] = 0.0
    d[50:62, 70:90] = 0.0
    d[:, :5] = 0.0
    return d


def make_inpaint(out: Path) -> bool:
    """cv2.inpaint(depth, mask, 5, cv2.INPAINT_NS) exactly as inference/icp_refiner.py:54 calls it: the one cv2 call of get_normal that is
    NOT restated (engine and oracle use an onion-peel mean fill instead, INTEGRATION.md "Known deviation"); the fixture quantifies the gap."""
    try:
        import cv2
    except Exception as e:  # noqa: BLE001
        print(f"[skip] cv2 not available ({e})")
        return False
    d = inpaint_inputs()
    mask = (d == 0).astype(np.uint8)
    filled = cv2.inpaint(d, mask, 5, cv2.INPAINT_NS)
    np.savez_compressed(out / "thirdparty_inpaint.npz", depth=d, filled=np.asarray(filled, np.float32), cv2_version=str(cv2.__version__))
    print("thirdparty_inpaint.npz written, cv2", cv2.__version__)
    return True


def panda3d_views():
    """the lathe test mesh (tests/support/synthetic.make_lathe_mesh, seed 0) under three poses, K of the example scaled to 320x240"""
    from tests.support import synthetic as syn

    rng = np.random.RandomState(23)
    T = np.stack([syn.random_pose(rng, z_range=(0.35, 0.6)) for _ in range(3)])
    K = syn.K_EXAMPLE.astype(np.float32).copy()
    K[:2] *= 0.5
    return T, K


def make_panda3d(out: Path) -> bool:
    try:
        import panda3d  # noqa: F401

        # the REAL reference package (its own environment: `pip install -e` of the megapose6d checkout), not oracle/ref_import's stubs
        from megapose.datasets.object_dataset import RigidObject, RigidObjectDataset
        from megapose.panda3d_renderer.panda3d_scene_renderer import Panda3dSceneRenderer
        from megapose.panda3d_renderer.types import Panda3dCameraData, Panda3dLightData, Panda3dObjectData
        from megapose.lib3d.transform import Transform
    except Exception as e:  # noqa: BLE001
        print(f"[skip] panda3d / the reference renderer not importable ({e})")
        return False
    import tempfile

    from tests.support import synthetic as syn

    tmp = Path(tempfile.mkdtemp(prefix="mp_p3d_"))
    ds_ = syn.make_object_dataset(tmp, n_objects=1, seed=0)
    ds = RigidObjectDataset([RigidObject(label=o.label, mesh_path=o.mesh_path, mesh_units=o.mesh_units) for o in ds_.list_objects])
    renderer = Panda3dSceneRenderer(ds)
    T, K = panda3d_views()
    rgbs, normals, depths = [], [], []
    for Tv in T:
        cam = Panda3dCameraData(K=K, resolution=(240, 320), TWC=Transform(np.linalg.inv(Tv)))
        obj = Panda3dObjectData(label=ds_[0].label, TWO=Transform(np.eye(4)))
        lights = [Panda3dLightData(light_type="ambient", color=(1.0, 1.0, 1.0, 1.0))]
        rend = renderer.render_scene([obj], [cam], lights, render_depth=True, render_normals=True)[0]
        rgbs.append(rend.rgb); normals.append(rend.normals); depths.append(rend.depth)
    np.savez_compressed(out / "thirdparty_panda3d.npz", rgb=np.stack(rgbs), normals=np.stack(normals), depth=np.stack(depths), T=T, K=K,
                        panda3d_version=str(getattr(panda3d, "__version__", "?")))
    print("thirdparty_panda3d.npz written")
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden"))
    a = ap.parse_args()
    out = Path(a.out)
    out.mkdir(parents=True, exist_ok=True)
    done = [f.__name__ for f in (make_roi_align, make_maskrcnn, make_icp, make_inpaint, make_panda3d) if f(out)]
    print("fixtures written by:", done or "none (no third-party package importable here)")


if __name__ == "__main__":
    main()
