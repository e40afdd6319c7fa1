"""Micro-benchmark of the rasteriser on crop-like (zoomed) views: ms per launch of 2304 views = 576 items x 4 views written
into a 32-channel CNN input tensor, and the algorithmic-bytes bandwidth (output channels + mesh once per view)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from megapose6d_amd import engine as eng, mesh_io
from tests.support import synthetic as syn

ds = syn.make_object_dataset("/tmp/mp_rb", 1, 0)
mesh = mesh_io.load_rigid_object(ds[0])
db = eng.MeshDB([mesh])
n = 2304
rng = np.random.RandomState(0)
T = torch.from_numpy(np.stack([syn.random_pose(rng, z_range=(0.4, 0.6), xy_frac=0.02) for _ in range(n)])).cuda()
K = torch.tensor([[1500.0, 0, 160], [0, 1500.0, 120], [0, 0, 1]]).repeat(n, 1, 1).cuda()  # crop-like zoom: object fills the view
ids = torch.zeros(n, dtype=torch.int32, device="cuda")
# MP_RB_LAYOUT="CP,C0": channels per pixel record and first written channel (default 32,3: 24 written channels of a 128-byte record;
# "24,0" = the records are written completely: rows are contiguous, no holes)
import os
CP, C0 = (int(v) for v in os.environ.get("MP_RB_LAYOUT", "32,3").split(","))
out = torch.zeros(n // 4, 246, 326, CP, device="cuda")
L = eng.make_lights()
flags_list = [int(a) for a in sys.argv[1:]] or [1]
for flags in flags_list:
    for rep in range(3):
        eng.profile_begin()
        eng.raster_render(db, ids, T, K, 240, 320, flags, L, out, 246 * 326 * CP, 326 * CP, CP, C0, C0 + 3, -1, (3 * 326 + 3) * CP,
                          views_per_item=4, stride_view=6)
        prof = eng.profile_end()
    tot = sum(v["ms"] for v in prof.values())
    by = sum(v["bytes"] for k, v in prof.items() if k.startswith("raster_bands") or k.startswith("raster_tiles"))
    print(f"flags={flags}: {tot:8.3f} ms for {n} views  ({by / tot / 1e6:.0f} GB/s algorithmic)  " +
          ", ".join(f"{k} {v['ms']:.3f}" for k, v in prof.items()))
cov = (out[..., C0:C0 + 3].sum(-1) > 0).float().mean().item()
print("coverage of view 0 channel block:", cov)
