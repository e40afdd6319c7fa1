"""Time the two on-device depth refiners on the 12 synthetic ICP scenes (one call refines all 12 objects)."""
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from test_icp_oracles_cpu import make_icp_scenes  # noqa: E402

from megapose6d_amd.icp_refiner import ICPRefiner  # noqa: E402
from megapose6d_amd.renderer import Panda3dBatchRenderer  # noqa: E402
from megapose6d_amd.tcoll import PandasTensorCollection  # noqa: E402

ds, scenes = make_icp_scenes(12)
r = Panda3dBatchRenderer(ds, n_workers=1)
K = torch.from_numpy(scenes[0][1]).cuda()[None].repeat(12, 1, 1)
depth = torch.from_numpy(np.stack([s[0] for s in scenes])).cuda()
init = np.stack([s[2] for s in scenes])
preds = PandasTensorCollection(pd.DataFrame(dict(label=[s[5] for s in scenes], batch_im_id=np.arange(12), instance_id=0)), poses=torch.from_numpy(init).cuda())
for assoc in ("nn", "projective"):
    ref = ICPRefiner(None, r, association=assoc)
    ref.refine_poses(preds, depth=depth, K=K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out, extra = ref.refine_poses(preds, depth=depth, K=K)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    line = f"{assoc}: {ms:.1f} ms per call of 12 objects"
    if "iterations_per_level" in extra:
        line += f"; iterations per level (0..3) mean {extra['iterations_per_level'].float().mean(0).tolist()}"
    print(line)
