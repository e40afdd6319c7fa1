"""Per-phase shader-cycle breakdown of raster_tiles (profiling build: make -C megapose6d_amd/csrc prof; run with
MP_ENGINE_LIB=scripts/microbench/_build/libmp_engine_prof.so).  Phases: 0 list fetch, 1 piece set-up, 2 scatter, 3 sweep,
4 z read-back + task build, 5 shading, 6 resolve + staging, 7 crop, 8 store."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from megapose6d_amd import _lib, engine as eng, mesh_io
from tests.support import synthetic as syn

lib = _lib.load()
lib.mp_raster_prof_read.restype = C.c_int
ds = syn.make_object_dataset("/tmp/mp_rp", 1, 0)
db = eng.MeshDB([mesh_io.load_rigid_object(ds[0])])
n = 2304
names = ["init/empty", "list fetch", "block visits", "sweep (large)", "z read + tasks", "shading", "resolve+stage", "crop", "store"]
for label, f, zr in (("zoomed (object fills the view)", 1500.0, (0.4, 0.6)), ("pipeline-like (crop lambda 1.4)", 1000.0, (0.45, 0.7))):
    rng = np.random.RandomState(0)
    T = torch.from_numpy(np.stack([syn.random_pose(rng, z_range=zr, xy_frac=0.02) for _ in range(n)])).cuda()
    K = torch.tensor([[f, 0, 160], [0, f, 120], [0, 0, 1]]).repeat(n, 1, 1).cuda()
    ids = torch.zeros(n, dtype=torch.int32, device="cuda")
    out = torch.zeros(n // 4, 246, 326, 32, device="cuda")
    for flags in (1, 17):
        buf = (C.c_ulonglong * 16)()
        for rep in range(2):
            lib.mp_raster_prof_read(buf, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.raster_render(db, ids, T, K, 240, 320, flags, eng.make_lights(), out, 246 * 326 * 32, 326 * 32, 32, 3, 6, -1, (3 * 326 + 3) * 32,
                              views_per_item=4, stride_view=6)
            e1.record()
            torch.cuda.synchronize()
        lib.mp_raster_prof_read(buf, 0)
        tot = sum(buf[i] for i in range(9))
        if buf[11]:
            print(f"    per tile-view with geometry: {buf[10] / buf[11]:.1f} records, {buf[9] / buf[11]:.1f} visits, {buf[12] / buf[11]:.2f} batches "
                  f"(sampled waves: {buf[11]} tile-views)")
            if buf[12]:
                print(f"    inside the block-visit phase, wave cycles per batch: set-up (BlkRec, block tests) {buf[13] / buf[12]:.0f}, visits {buf[14] / buf[12]:.0f} "
                      f"= {100.0 * buf[13] / max(tot, 1):.1f} % / {100.0 * buf[14] / max(tot, 1):.1f} % of all wave cycles (the wait for the records is counted as list fetch)")
        print(f"{label}, flags={flags}: {e0.elapsed_time(e1):.2f} ms; wave-cycles by phase: " +
              ", ".join(f"{names[i]} {100.0 * buf[i] / tot:.1f}%" for i in range(9)) + f"  (total {tot / 1e9:.2f} G wave-cycles)", flush=True)
    cov = (out[..., 3:6].sum(-1) > 0).float().mean().item()
    print("  coverage", cov)
