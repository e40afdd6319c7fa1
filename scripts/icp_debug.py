"""Debug aid: run mp_icp_refine_nn on two of the synthetic ICP scenes with a caller-owned workspace and compare the intermediate device
buffers (points + normals of every pixel, compacted clouds, centroid shift, normalisation) with oracle/icp_opencv.py."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from test_icp_oracles_cpu import make_icp_scenes  # noqa: E402

from megapose6d_amd import _lib  # noqa: E402
from megapose6d_amd.renderer import Panda3dBatchRenderer  # noqa: E402
from oracle import icp_opencv as ocv  # noqa: E402

ids = [int(a) for a in sys.argv[1:]] or [5, 2]
ds, scenes = make_icp_scenes(12)
r = Panda3dBatchRenderer(ds, n_workers=1)
N = len(ids)
H, W = 480, 640
K = torch.from_numpy(scenes[0][1]).cuda()[None].repeat(N, 1, 1).contiguous()
depth = torch.from_numpy(np.stack([scenes[i][0] for i in ids])).cuda().contiguous()
init = torch.from_numpy(np.stack([scenes[i][2] for i in ids])).cuda().contiguous()
rend = r.render_depth([scenes[i][5] for i in ids], init, K, (H, W)).contiguous()
lib = _lib.load()
ws = torch.zeros(lib.mp_icp_nn_workspace_bytes(N, N, H, W), dtype=torch.uint8, device="cuda")
out = torch.empty_like(init)
retval = torch.empty(N, dtype=torch.int32, device="cuda")
residual = torch.empty(N, dtype=torch.float32, device="cuda")
im_ids = torch.arange(N, dtype=torch.int32, device="cuda")
rc = lib.mp_icp_refine_nn(depth.data_ptr(), N, im_ids.data_ptr(), rend.data_ptr(), K.data_ptr(), K.data_ptr(), init.data_ptr(), N, H, W, 100, 4, 0.05, 1000,
                          None, out.data_ptr(), retval.data_ptr(), residual.data_ptr(), None, ws.data_ptr(), ws.numel(), None)
assert rc == 0
torch.cuda.synchronize()
w = ws.cpu().numpy()
a256 = lambda v: (v + 255) & ~255
px, cap, imgs = H * W, min(H * W, lib.mp_icp_nn_max_points()), 2 * N
off = 0
def take(nbytes):
    global off
    o = off
    off += a256(nbytes)
    return o
o_fa, o_fb, o_raw = take(imgs * px * 4), take(imgs * px * 4), take(imgs * px * 4)
o_va, o_vb = take(imgs * px), take(imgs * px)
o_pts = take(imgs * px * 24)
o_src, o_dst = take(N * cap * 24), take(N * cap * 24)
for _ in range(3):
    take(N * cap * 24)
for _ in range(3):
    take(N * cap * 4)
take(N * cap * 8)
take(N * cap * 8)
o_rows = take(0)
row_dt = np.dtype([("n", "i4"), ("m", "i4"), ("status", "i4"), ("iters", "i4", 8), ("active", "i4"), ("level", "i4"), ("it", "i4"), ("nl", "i4"),
                   ("ml", "i4"), ("max_it", "i4"), ("tol_p", "f8"), ("fval", "f8", 3), ("pose_x", "f8", 16), ("scale", "f8"), ("mean_avg", "f8", 3),
                   ("shift", "f8", 3), ("pose", "f8", 16), ("residual", "f8")], align=True)
rows = np.frombuffer(w[o_rows:o_rows + N * row_dt.itemsize].tobytes(), dtype=row_dt)
pts = np.frombuffer(w[o_pts:o_pts + imgs * px * 24].tobytes(), dtype=np.float32).reshape(imgs, H, W, 6)
rend_np, depth_np = rend.cpu().numpy(), depth.cpu().numpy()
for k, sid in enumerate(ids):
    dm, Kn, T0 = scenes[sid][0], scenes[sid][1], scenes[sid][2]
    fx, fy, cx, cy = Kn[0, 0], Kn[1, 1], Kn[0, 2], Kn[1, 2]   # numpy float32 scalars, as the reference passes them
    mask = ocv.compute_masks_threshold(rend_np[k], dm)
    valid = (dm > 0.2) & (dm < 5) & mask
    for name, d, img in (("measured", dm, k), ("rendered", rend_np[k], N + k)):
        sel = valid if name == "measured" else valid & (rend_np[k] > 0)
        xyz = ocv.get_xyz(d, fx, fy, cx, cy).astype(np.float32)
        nrm = ocv.get_normal(d, fx, fy, cx, cy).astype(np.float32)
        dx = np.abs(pts[img][..., :3] - xyz)[sel]
        dn = np.abs(pts[img][..., 3:] - nrm)[sel]
        print(f"scene {sid} {name}: points {sel.sum()}  max|xyz diff| {dx.max():.3e}  max|normal diff| {dn.max():.3e}  (normals differing > 1e-6: {(dn.max(1) > 1e-6).sum()})")
        if dn.max() > 1e-6:
            yy, xx = np.nonzero(sel)
            bad = dn.max(1) > 1e-6
            print("   worst pixels (y, x):", list(zip(yy[bad][:8].tolist(), xx[bad][:8].tolist())), " bbox of the selection:", yy.min(), yy.max(), xx.min(), xx.max())
    info = {}
    T_cv, rv, res = ocv.icp_refinement(dm, rend_np[k], mask, Kn, T0, info=info)
    pt = ocv.get_xyz(dm, fx, fy, cx, cy).astype(np.float32)[valid]
    ps = ocv.get_xyz(rend_np[k], fx, fy, cx, cy).astype(np.float32)[valid & (rend_np[k] > 0)]
    shift = pt.mean(0) - ps.mean(0)
    print(f"   n/m device {rows[k]['n']}/{rows[k]['m']} oracle {len(ps)}/{len(pt)}; shift device {rows[k]['shift']} oracle {shift}")
    print(f"   residual device {residual[k].item():.10g} oracle {res:.10g}; iters device {rows[k]['iters'][:4]} oracle {info.get('iters')}; max|T diff| {np.abs(out[k].cpu().numpy() - T_cv).max():.3e}")
