"""Offline count (numpy, no GPU): how many block visits would the rasteriser save if one visit served several records, one per lane group
(pixel rows / halves / 2x2 quads of a 4x4 block)?  Bench mesh, pipeline-like camera, bbox-based touching.  Result:
rows 0.75, halves 0.83, quads 0.71 of the present visit count -- the densest group of a block dominates."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from tests.support import synthetic as syn
from megapose6d_amd import mesh_io
ds = syn.make_object_dataset('/tmp/mp_vs', 1, 0)
m = mesh_io.load_rigid_object(ds[0])
V = np.asarray(m['vertices'], np.float64); F = np.asarray(m['faces'])
print(V.shape, F.shape, np.abs(V).max())
rng = np.random.RandomState(0)
f = 1000.0; W, H = 320, 240
tot_n = tot_rows = tot_half = tot_quad = tot_unionrows = 0
tot_rec_blocks = 0
for it in range(12):
    T = syn.random_pose(rng, z_range=(0.45, 0.7), xy_frac=0.02)
    P = V @ T[:3, :3].T + T[:3, 3]
    x = f * P[:, 0] / P[:, 2] + 160; y = f * P[:, 1] / P[:, 2] + 120
    tx = x[F]; ty = y[F]
    # both orientations are rendered (two-sided): all triangles
    x0 = tx.min(1); x1 = tx.max(1); y0 = ty.min(1); y1 = ty.max(1)
    # sample positions span [px+0.125, px+0.875]; pixel index range whose sample range intersects the bbox
    px0 = np.clip(np.ceil(x0 - 0.875).astype(int), 0, W - 1); px1 = np.clip(np.floor(x1 - 0.125).astype(int), -1, W - 1)
    py0 = np.clip(np.ceil(y0 - 0.875).astype(int), 0, H - 1); py1 = np.clip(np.floor(y1 - 0.125).astype(int), -1, H - 1)
    ok = (px1 >= px0) & (py1 >= py0)
    from collections import defaultdict
    blocks = defaultdict(lambda: np.zeros(4, int)); halves = defaultdict(lambda: np.zeros(2, int)); quads = defaultdict(lambda: np.zeros(4, int)); cnt = defaultdict(int)
    for i in np.nonzero(ok)[0]:
        for bx in range(px0[i] // 4, px1[i] // 4 + 1):
            for by in range(py0[i] // 4, py1[i] // 4 + 1):
                r0 = max(py0[i], by * 4) - by * 4; r1 = min(py1[i], by * 4 + 3) - by * 4
                c0 = max(px0[i], bx * 4) - bx * 4; c1 = min(px1[i], bx * 4 + 3) - bx * 4
                cnt[(bx, by)] += 1
                blocks[(bx, by)][r0:r1 + 1] += 1
                halves[(bx, by)][r0 // 2:r1 // 2 + 1] += 1
                for qy in range(r0 // 2, r1 // 2 + 1):
                    for qx in range(c0 // 2, c1 // 2 + 1):
                        quads[(bx, by)][qy * 2 + qx] += 1
    n = sum(cnt.values()); rows = sum(v.max() for v in blocks.values()); hv = sum(v.max() for v in halves.values()); qd = sum(v.max() for v in quads.values())
    tot_n += n; tot_rows += rows; tot_half += hv; tot_quad += qd
    print(it, 'visits(block)', n, 'rows', rows, 'halves', hv, 'quads', qd, 'tri/px extent', np.median((x1 - x0)[ok]), np.median((y1 - y0)[ok]))
print('ratio rows', tot_rows / tot_n, 'halves', tot_half / tot_n, 'quads', tot_quad / tot_n)
