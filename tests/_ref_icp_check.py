"""Run in a SUBPROCESS by tests/test_icp_oracles_cpu.py (build container only: needs /root/reference).
Pins oracle/icp_opencv.py to the REFERENCE's own depth-refiner code (src/megapose/inference/icp_refiner.py, refiner_utils.py) wherever
that code does not depend on OpenCV: getXYZ, get_normal, compute_masks and the orchestration of icp_refinement are the reference's
functions, imported and executed here; only the two cv2 calls are stand-ins (cv2.inpaint -> the oracle's hole fill,
cv2.ppf_match_3d_ICP -> the oracle's restatement of OpenCV's ICP).  Prints a JSON list of problems (empty = OK)."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from oracle import ref_import  # noqa: E402

ref_import.install()

import megapose.inference.icp_refiner as r_icp  # noqa: E402
import megapose.inference.refiner_utils as r_ru  # noqa: E402

from oracle import icp_opencv as ocv  # noqa: E402
from oracle import raster as orr  # noqa: E402
from test_icp_oracles_cpu import make_icp_scenes  # noqa: E402

problems = []


class _ICP:   # cv2.ppf_match_3d_ICP(iterations, tolerence=, numLevels=) -> .registerModelToScene(src, dst)
    def __init__(self, iterations, tolerence=0.05, rejectionScale=2.5, numLevels=6):
        self.a = (iterations, tolerence, rejectionScale, numLevels)

    def registerModelToScene(self, src, dst):
        return ocv.opencv_icp(src, dst, *self.a)


r_icp.cv2.inpaint = lambda depth, mask, radius, flags: ocv._fill_holes(depth)
r_icp.cv2.ppf_match_3d_ICP = _ICP

_, scenes = make_icp_scenes(8)
for n, (dm, K, init, gt, mesh, _) in enumerate(scenes[:4] + [scenes[7]]):
    dr = orr.render(mesh, init[None], K[None], 480, 640, 2)[2][0]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]   # numpy float32 scalars, as icp_refinement passes them
    if not np.array_equal(r_icp.getXYZ(dm, fx, fy, cx, cy), ocv.get_xyz(dm, fx, fy, cx, cy)):
        problems.append(f"scene {n}: getXYZ differs")
    for refine in (False, True):
        a, b = r_icp.get_normal(dm, fx=fx, fy=fy, cx=cx, cy=cy, refine=refine), ocv.get_normal(dm, fx, fy, cx, cy, refine=refine)
        if a.dtype != b.dtype or not np.array_equal(a, b):
            problems.append(f"scene {n}: get_normal(refine={refine}) differs (max {np.abs(a - b).max():.3e})")
    mr, mm = r_ru.compute_masks("threshold", dr, dm, 0.1)
    mine = ocv.compute_masks_threshold(dr, dm)
    if not (np.array_equal(mm, mine) and np.array_equal(mr, mine)):
        problems.append(f"scene {n}: compute_masks differs")
    T_ref, rv_ref = r_icp.icp_refinement(dm, dr, mm, K, init, n_min_points=1000)
    T_o, rv_o, _ = ocv.icp_refinement(dm, dr, mine, K, init)
    T_ref = np.asarray(T_ref, dtype=np.float32)
    if rv_ref != rv_o:
        problems.append(f"scene {n}: retval {rv_ref} vs {rv_o}")
    elif rv_o == 0 and not np.array_equal(T_ref, T_o):
        problems.append(f"scene {n}: refined pose differs (max {np.abs(T_ref - T_o).max():.3e})")
print("REF_ICP_JSON " + json.dumps(problems))
