"""CPU restatement of the REFERENCE's depth refiner: /root/reference/src/megapose/inference/icp_refiner.py
(`get_normal` :37-95, `getXYZ` :98-125, `icp_refinement` :128-175, mask rule refiner_utils.py:30-56) with its two third-party
calls restated from their published sources, because neither package is installed here ("parity unpinned" for both):

* `cv2.ppf_match_3d_ICP(100, tolerence=0.05, numLevels=4).registerModelToScene(src, dst)` -- opencv_contrib 4.x,
  modules/surface_matching/src/icp.cpp (`ICP::registerModelToScene`, `minimizePointToPlaneMetric`, `getRejectionThreshold`,
  `eulerToDCM`) and src/c_utils.hpp / ppf_helpers.cpp (`transformPCPose`, `samplePCUniform`, `computeDistToOrigin`, `medianF`):
  mean/scale normalisation, a 4-level pyramid (every 8th / 4th / 2nd / every point, 25 / 33 / 50 / 100 iterations at most,
  relative-change stop at tolerance * (level + 1)^2), nearest-neighbour association with a kd-tree on the (sub-sampled) scene,
  robust rejection at median + 2.5 * 1.4826 * MAD of the squared distances, "picky" one-to-one filtering, a linearised
  point-to-plane least-squares step (SVD) for the FULL level transform from the level's start points, residual = Frobenius norm of
  the matched 6-d rows / number of model points.  FLANN's kd-tree is replaced by scipy.spatial.cKDTree (both exact).
* `cv2.inpaint(depth, mask, 2, cv2.INPAINT_NS)` in get_normal -- the Navier-Stokes inpainting of holes in the depth map is NOT
  reproduced; holes are filled by an onion-peel mean of valid 8-neighbours (same role: no zero-depth cliffs under the Gaussian).
  `scipy.ndimage.gaussian_filter(., 2)` and `np.gradient(., 2, edge_order=2)` are the reference's own calls.

TEST INFRASTRUCTURE ONLY: tests compare the engine's on-device refiners with this restatement on synthetic scenes -- csrc/icp_nn.hip
(the default: this algorithm step for step, in the same arithmetic; measured bit-identical poses) and csrc/icp.hip (the optional
projective-association point-to-plane ICP; the tests state the bounds within which it agrees)."""
from __future__ import annotations

import numpy as np
from scipy import ndimage
from scipy.spatial import cKDTree


# ---------------------------------------------------------------------------------------------------------------------------
# get_normal / getXYZ (icp_refiner.py:37-125, copied semantics incl. the int16 truncation of the pixel-offset table)
# ---------------------------------------------------------------------------------------------------------------------------
def _fill_holes(depth: np.ndarray) -> np.ndarray:
    """substitute for cv2.inpaint(..., 2, INPAINT_NS): zeros are filled ring by ring with the mean of their valid 8-neighbours"""
    d = depth.astype(np.float32).copy()
    valid = d != 0
    if not valid.any():
        return d
    k = np.ones((3, 3), np.float32)
    while not valid.all():
        s = ndimage.convolve(np.where(valid, d, 0).astype(np.float32), k, mode="constant")
        c = ndimage.convolve(valid.astype(np.float32), k, mode="constant")
        new = (~valid) & (c > 0)
        if not new.any():
            break
        d[new] = s[new] / c[new]
        valid = valid | new
    return d


def _uv_table(res_y: int, res_x: int, cx: float, cy: float) -> np.ndarray:
    uv = np.zeros((res_y, res_x, 2), dtype=np.int16)
    column = np.arange(0, res_y)
    uv[:, :, 1] = np.arange(0, res_x) - cx          # float -> int16: truncation toward zero, as in the reference
    uv[:, :, 0] = column[:, np.newaxis] - cy
    return uv


def get_normal(depth: np.ndarray, fx: float, fy: float, cx: float, cy: float, refine: bool = True) -> np.ndarray:
    res_y, res_x = depth.shape
    constant_x, constant_y = 1 / fx, 1 / fy
    d = depth
    if refine:
        d = np.nan_to_num(d).astype(np.float32)
        d = _fill_holes(d)
        d = ndimage.gaussian_filter(d.astype(np.float32), 2)
    uv = _uv_table(res_y, res_x, cx, cy)
    dig = np.gradient(d, 2, edge_order=2)
    v_y = np.zeros((res_y, res_x, 3))
    v_x = np.zeros((res_y, res_x, 3))
    v_y[:, :, 0] = uv[:, :, 1] * constant_x * dig[0]
    v_y[:, :, 1] = d * constant_y + (uv[:, :, 0] * constant_y) * dig[0]
    v_y[:, :, 2] = dig[0]
    v_x[:, :, 0] = d * constant_x + uv[:, :, 1] * constant_x * dig[1]
    v_x[:, :, 1] = uv[:, :, 0] * constant_y * dig[1]
    v_x[:, :, 2] = dig[1]
    cross = np.cross(v_x.reshape(-1, 3), v_y.reshape(-1, 3))
    norm = np.expand_dims(np.linalg.norm(cross, axis=1), axis=1)
    norm[norm == 0] = 1
    cross = cross / norm
    return np.nan_to_num(cross.reshape(res_y, res_x, 3))


def get_xyz(depth: np.ndarray, fx: float, fy: float, cx: float, cy: float) -> np.ndarray:
    uv = _uv_table(depth.shape[0], depth.shape[1], cx, cy)
    xyz = np.zeros((depth.shape[0], depth.shape[1], 3))
    xyz[:, :, 0] = uv[:, :, 1] * depth * 1 / fx
    xyz[:, :, 1] = uv[:, :, 0] * depth * 1 / fy
    xyz[:, :, 2] = depth
    return xyz


# ---------------------------------------------------------------------------------------------------------------------------
# cv::ppf_match_3d::ICP
# ---------------------------------------------------------------------------------------------------------------------------
def _cv_round(x: float) -> int:
    return int(np.rint(x))  # cvRound: round half to even


def _median_f(a: np.ndarray) -> float:
    """medianF (quick-select): the LOWER median, element (n - 1) // 2 of the sorted array"""
    n = a.shape[0]
    return float(np.partition(a, (n - 1) // 2)[(n - 1) // 2])


def _rejection_threshold(r: np.ndarray, scale: float) -> float:
    med = _median_f(r)
    s = 1.48257968 * _median_f(np.abs(r.astype(np.float64) - med).astype(np.float32))
    return float(np.float32(scale * s + med))


def _euler_to_dcm(e: np.ndarray) -> np.ndarray:
    cx, sx, cy, sy, cz, sz = np.cos(e[0]), np.sin(e[0]), np.cos(e[1]), np.sin(e[1]), np.cos(e[2]), np.sin(e[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ (Ry @ Rz)


def _transform_pc(pc: np.ndarray, pose: np.ndarray) -> np.ndarray:
    """transformPCPose: points by the pose, normals by its rotation (re-normalised); float32 like the CV_32F cloud"""
    out = np.empty_like(pc)
    R, t = pose[:3, :3], pose[:3, 3]
    out[:, :3] = (pc[:, :3].astype(np.float64) @ R.T + t).astype(np.float32)
    n = pc[:, 3:6].astype(np.float64) @ R.T
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    out[:, 3:6] = np.where(ln > 1e-12, n / np.where(ln > 1e-12, ln, 1), n).astype(np.float32)
    return out


def _point_to_plane(src: np.ndarray, dst: np.ndarray):
    axis = np.cross(src[:, :3], dst[:, 3:6])
    A = np.concatenate([axis, dst[:, 3:6]], axis=1)
    b = ((dst[:, :3] - src[:, :3]) * dst[:, 3:6]).sum(1)
    x = np.linalg.lstsq(A, b, rcond=None)[0]        # cv::solve(A, b, x, DECOMP_SVD)
    return x[:3], x[3:]


def opencv_icp(src_pc: np.ndarray, dst_pc: np.ndarray, iterations: int = 100, tolerance: float = 0.05, rejection_scale: float = 2.5,
               num_levels: int = 4, info: dict | None = None):
    """registerModelToScene(srcPC [n,6], dstPC [m,6]) -> (retval 0, residual, pose 4x4 mapping src onto dst).
    `info` (optional dict) receives telemetry: 'iters' = iterations run per level (index = level)."""
    n = src_pc.shape[0]
    src = src_pc.astype(np.float32).copy()
    dst = dst_pc.astype(np.float32).copy()
    mean_avg = 0.5 * (src[:, :3].astype(np.float64).mean(0) + dst[:, :3].astype(np.float64).mean(0))
    src[:, :3] = (src[:, :3] - mean_avg).astype(np.float32)
    dst[:, :3] = (dst[:, :3] - mean_avg).astype(np.float32)
    dist_src = np.linalg.norm(src[:, :3].astype(np.float64), axis=1).sum()   # computeDistToOrigin: SUM of the norms
    dist_dst = np.linalg.norm(dst[:, :3].astype(np.float64), axis=1).sum()
    scale = n / ((dist_src + dist_dst) * 0.5)
    src[:, :3] *= np.float32(scale)
    dst[:, :3] *= np.float32(scale)
    pose = np.eye(4)
    residual = 0.0
    for level in range(num_levels - 1, -1, -1):
        div = 2.0 ** level
        num_samples = _cv_round(n / div)
        tol_p = tolerance * (level + 1) * (level + 1)
        max_it = _cv_round(iterations / (level + 1))
        step = max(_cv_round(n / num_samples), 1)
        src_t = _transform_pc(src, pose)[::step]
        dst_s = dst[::step]
        tree = cKDTree(dst_s[:, :3])
        fval_old, fval_perc, fval_min = 9999999999.0, 0.0, 9999999999.0
        moved = src_t.copy()
        pose_x = np.eye(4)
        i = 0
        while (not (1 - tol_p < fval_perc < 1 + tol_p)) and i < max_it:
            d, ind = tree.query(moved[:, :3])
            d2 = (d * d).astype(np.float32)                               # FLANN's L2 functor returns SQUARED distances
            new_i, new_j = np.arange(len(moved)), ind
            if rejection_scale > 0:
                acc = d2 < _rejection_threshold(d2, rejection_scale)
                new_i, new_j = new_i[acc], new_j[acc]
            # picky ICP: a scene point keeps only its closest model point
            order = np.lexsort((d2[new_i], new_j))
            nj, ni = new_j[order], new_i[order]
            first = np.ones(len(nj), bool)
            first[1:] = nj[1:] != nj[:-1]
            idx_model, idx_scene = ni[first], nj[first]
            if len(idx_model) < 6:
                break
            s_m, d_m = src_t[idx_model].astype(np.float64), dst_s[idx_scene].astype(np.float64)
            rpy, t = _point_to_plane(s_m, d_m)
            if np.isnan(rpy).any() or np.isnan(t).any():
                break
            pose_x = np.eye(4)
            pose_x[:3, :3] = _euler_to_dcm(rpy)
            pose_x[:3, 3] = t
            moved = _transform_pc(src_t, pose_x)
            fval = float(np.linalg.norm(s_m - d_m)) / len(moved)
            fval_perc = fval / fval_old
            fval_old = fval
            fval_min = min(fval_min, fval)
            i += 1
        pose = pose_x @ pose
        residual = fval_min
        if info is not None:
            info.setdefault("iters", [0] * num_levels)[level] = i
    R, c = pose[:3, :3], pose[:3, 3]
    c = c / scale + mean_avg - R @ mean_avg
    out = np.eye(4)
    out[:3, :3], out[:3, 3] = R, c
    return 0, residual, out


# ---------------------------------------------------------------------------------------------------------------------------
def compute_masks_threshold(depth_rendered: np.ndarray, depth_measured: np.ndarray, depth_delta_thresh: float = 0.1) -> np.ndarray:
    """refiner_utils.py:30-56, mask_type='threshold': the measured-side mask"""
    mask_measured = np.logical_and(depth_measured > 0, depth_rendered > 0)
    mask_measured[np.abs(depth_measured - depth_rendered) > depth_delta_thresh] = 0
    return mask_measured


def icp_refinement(depth_measured, depth_rendered, object_mask_measured, cam_K, TCO_pred, n_min_points=1000, info=None):
    """icp_refiner.py:128-175 -> (TCO_refined, retval, residual); retval -1 = rejected (the caller keeps the input pose)"""
    # the reference hands numpy SCALARS of the float32 intrinsics to getXYZ / get_normal (:135-150), not python floats: `1 / fx` and every
    # product of the int16 offset table with it are then float32 operations (under numpy 1.x value-based casting and NEP 50 alike)
    cam_K = np.asarray(cam_K, dtype=np.float32)
    fx, fy, cx, cy = cam_K[0, 0], cam_K[1, 1], cam_K[0, 2], cam_K[1, 2]
    H, W = depth_measured.shape
    pts_tgt = np.zeros((H, W, 6), np.float32)
    pts_tgt[:, :, :3] = get_xyz(depth_measured, fx, fy, cx, cy)
    pts_tgt[:, :, 3:] = get_normal(depth_measured, fx, fy, cx, cy, refine=True)
    depth_valid = np.logical_and(depth_measured > 0.2, depth_measured < 5)
    depth_valid = np.logical_and(depth_valid, object_mask_measured)
    pts_tgt = pts_tgt[depth_valid]
    pts_src = np.zeros((H, W, 6), np.float32)
    pts_src[:, :, :3] = get_xyz(depth_rendered, fx, fy, cx, cy)
    pts_src[:, :, 3:] = get_normal(depth_rendered, fx, fy, cx, cy, refine=True)
    pts_src = pts_src[np.logical_and(depth_valid, depth_rendered > 0)]
    if len(pts_tgt) < n_min_points or len(pts_src) < n_min_points:
        return TCO_pred.copy(), -1, -1.0
    T = np.asarray(TCO_pred, dtype=np.float32).copy()                  # (the reference's pose is a float32 array; += stays float32)
    # np.mean over axis 0 of a float32 [n, 3] view: a SEQUENTIAL float32 accumulation (pairwise summation only runs along the fast axis)
    shift = pts_tgt[:, :3].mean(0) - pts_src[:, :3].mean(0)
    T[:3, 3] += shift.reshape(-1)
    pts_src[:, :3] += shift[None]
    tolerance = 0.05
    retval, residual, pose = opencv_icp(pts_src.reshape(-1, 6), pts_tgt.reshape(-1, 6), 100, tolerance, 2.5, 4, info=info)
    T = pose @ T
    if residual > tolerance or residual < 0:
        retval = -1
    return T.astype(np.float32), retval, residual
