"""CPU: the engine's depth-refiner ALGORITHM (oracle/icp.py = numpy restatement of csrc/icp.hip: projective point-to-plane ICP)
against the restatement of the REFERENCE's refiner (oracle/icp_opencv.py: get_normal + OpenCV ppf_match_3d ICP with kd-tree
association, icp_refiner.py:37-175) on synthetic scenes.  Stated bounds: identical accept/reject decisions; accepted poses within
1 mm / 2 degrees of each other; both within 1 mm of the ground truth translation."""
import tempfile
from pathlib import Path

import numpy as np
import pytest


def _rot_err_deg(A, B):
    R = A[:3, :3] @ B[:3, :3].T
    return float(np.rad2deg(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))))


def make_icp_scenes(n_scenes: int, seed: int = 3):
    """-> list of (depth_measured [480,640], K, init pose, gt pose, mesh) rendered with the oracle rasteriser (centre samples)"""
    from megapose6d_amd import mesh_io
    from tests.support import synthetic as syn
    from oracle import icp as oicp
    from oracle import raster as orr

    ds = syn.make_object_dataset(tempfile.mkdtemp(prefix="mp_icp_o_"), n_objects=3, seed=31)
    meshes = [mesh_io.load_rigid_object(o) for o in ds.list_objects]
    K = syn.K_EXAMPLE.astype(np.float32)
    rng = np.random.RandomState(seed)

    def perturb(T, rot_deg, trans):
        a = np.deg2rad(rot_deg) * rng.uniform(-1, 1, 3)
        P = T.copy().astype(np.float64)
        P[:3, :3] = oicp._rodrigues(a) @ P[:3, :3]
        P[:3, 3] += rng.uniform(-trans, trans, 3)
        return P.astype(np.float32)

    scenes = []
    for s in range(n_scenes):
        m = meshes[s % 3]
        gt = syn.random_pose(rng, (0.4, 0.8), 0.2)
        dm = orr.render(m, gt[None], K[None], 480, 640, 2)[2][0]
        dm = np.where(dm > 0, dm + np.random.RandomState(s).randn(480, 640).astype(np.float32) * 0.001, dm)
        if s % 4 == 3:  # an occluder in front of the left part of the frame
            dm[:, :300] = np.where(dm[:, :300] > 0, 0.3, dm[:, :300])
        init = perturb(gt, 1.0, 0.3) if s % 8 == 7 else perturb(gt, 3.0, 0.008)   # every 8th: 30 cm off -> must be rejected
        scenes.append((dm.astype(np.float32), K, init, gt, m, ds.list_objects[s % 3].label))
    return ds, scenes


def test_engine_icp_algorithm_agrees_with_the_opencv_restatement():
    from oracle import icp as oicp
    from oracle import icp_opencv as ocv
    from oracle import raster as orr

    _, scenes = make_icp_scenes(4)
    scenes.append(make_icp_scenes(8)[1][7])   # the far-off initial pose
    for dm, K, init, gt, mesh, _ in scenes:
        dr = orr.render(mesh, init[None], K[None], 480, 640, 2)[2][0]
        Ta, ra, _ = oicp.icp_refine(dm, dr, K, init)
        Tb, rb, _ = ocv.icp_refinement(dm, dr, ocv.compute_masks_threshold(dr, dm), K, init)
        assert ra == rb
        if ra == 0:
            assert np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]) < 1e-3 and _rot_err_deg(Ta, Tb) < 2.0
            assert np.linalg.norm(Ta[:3, 3] - gt[:3, 3]) < 1e-3 and np.linalg.norm(Tb[:3, 3] - gt[:3, 3]) < 1e-3
            assert np.linalg.norm(Ta[:3, 3] - gt[:3, 3]) < 0.2 * np.linalg.norm(init[:3, 3] - gt[:3, 3])
        else:
            assert np.array_equal(Ta, init) and np.array_equal(Tb, init)


def test_user_masks_replace_the_threshold_mask():
    """reference icp_refiner.py:249-250: with caller masks the 0.1 m threshold test is not applied -- a pose 15 cm off in depth is
    then refined instead of rejected for lack of points"""
    from oracle import icp as oicp
    from oracle import raster as orr

    from tests.support import synthetic as syn

    _, scenes = make_icp_scenes(1)
    _, K, _, _, mesh, _ = scenes[0]
    gt = syn.random_pose(np.random.RandomState(5), (0.42, 0.45), 0.05)      # close to the camera: a large silhouette
    dm = orr.render(mesh, gt[None], K[None], 480, 640, 2)[2][0]
    off = gt.copy()
    off[2, 3] += 0.15
    dr = orr.render(mesh, off[None], K[None], 480, 640, 2)[2][0]
    T0, r0, _ = oicp.icp_refine(dm, dr, K, off)
    assert r0 == -1
    seg = dm > 0   # a perfect segmentation of the object
    T1, r1, _ = oicp.icp_refine(np.where(seg, dm, 0), dr, K, off, user_masks=True)
    assert r1 == 0 and np.linalg.norm(T1[:3, 3] - gt[:3, 3]) < 5e-3


@pytest.mark.skipif(not Path("/root/reference/src/megapose/__init__.py").is_file(), reason="reference sources only exist in the build container")
def test_restatement_is_pinned_to_the_references_own_refiner_code_around_opencv():
    """getXYZ, get_normal, compute_masks and the orchestration of icp_refinement (mask / range selection, 1000-point rule, centroid
    pre-shift, float32 pose update, accept rule) are the REFERENCE's functions, imported and run; only cv2.inpaint and
    cv2.ppf_match_3d_ICP are stand-ins (the oracle's own): everything of row a20 except those two third-party calls is pinned
    bit for bit (tests/_ref_icp_check.py)."""
    import json
    import subprocess
    import sys

    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, str(root / "tests" / "_ref_icp_check.py")], capture_output=True, text=True, timeout=1200)
    line = next((l for l in p.stdout.splitlines() if l.startswith("REF_ICP_JSON ")), None)
    assert line is not None, p.stdout[-2000:] + p.stderr[-4000:]
    problems = json.loads(line[len("REF_ICP_JSON "):])
    assert problems == [], "\n".join(problems)
