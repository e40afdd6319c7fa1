"""Micro-benchmark of the rasteriser with phase-skip debug bits (profiling aid)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from megapose6d_amd import engine as eng, mesh_io, synthetic as syn

ds = syn.make_object_dataset("/tmp/mp_rb", 1, 0)
mesh = mesh_io.load_rigid_object(ds[0])
db = eng.MeshDB([mesh])
n = 2304
rng = np.random.RandomState(0)
T = torch.from_numpy(np.stack([syn.random_pose(rng, z_range=(0.4, 0.6), xy_frac=0.02) for _ in range(n)])).cuda()
K = torch.tensor([[1500.0, 0, 160], [0, 1500.0, 120], [0, 0, 1]]).repeat(n, 1, 1).cuda()  # crop-like zoom: object fills the view
ids = torch.zeros(n, dtype=torch.int32, device="cuda")
out = torch.zeros(n // 4, 246, 326, 32, device="cuda")
L = eng.make_lights()
names = {0: "full", 1 << 16: "skip pass1", 1 << 17: "skip wave-queue", (1 << 16) | (1 << 17): "skip pass1+1b", 1 << 18: "skip shading",
         1 << 19: "skip stores", (1 << 18) | (1 << 19): "skip shading+stores", (1 << 16) | (1 << 17) | (1 << 18) | (1 << 19): "skip all"}
for dbg, name in names.items():
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.raster_render(db, ids, T, K, 240, 320, 1 | dbg, L, out, 246 * 326 * 32, 326 * 32, 32, 3, 6, -1, (3 * 326 + 3) * 32,
                          views_per_item=4, stride_view=6)
        e1.record()
        torch.cuda.synchronize()
    print(f"{name:22s} {e0.elapsed_time(e1):8.3f} ms for {n} views")
cov = (out[..., 3:6].sum(-1) > 0).float().mean().item()
print("coverage of view 0 channel block:", cov)
