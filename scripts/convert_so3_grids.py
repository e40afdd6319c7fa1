"""Convert the SO(3) grid quaternion tables (Yershova/LaValle incremental Hopf-fibration grids, x y z w per line --
INPUT DATA of the algorithm, reference src/megapose/data/data_{72,512,576,4608}.qua, loaded by
src/megapose/utils/transform_utils.py:27-50) into compact float64 .npy files shipped with the engine.
Run once in the build container:  python scripts/convert_so3_grids.py"""
import sys
from pathlib import Path

import numpy as np

src = Path("/root/reference/src/megapose/data")
dst = Path(__file__).resolve().parent.parent / "megapose6d_amd" / "data"
dst.mkdir(exist_ok=True)
for n in (72, 512, 576, 4608):
    q = np.loadtxt(src / f"data_{n}.qua", dtype=np.float64)
    assert q.ndim == 2 and q.shape[1] == 4, q.shape  # NB: the reference "512" table actually holds 576 rows
    np.save(dst / f"so3_grid_{n}_xyzw.npy", q)
    print(n, q[:2])
