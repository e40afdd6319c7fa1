"""Per-layer-shape conv timing through the in-library event profiler (run with MP_PROF_DETAIL=1)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from megapose6d_amd import engine as eng
from tests.support import synthetic as syn

cin, b = int(sys.argv[1]) if len(sys.argv) > 1 else 27, int(sys.argv[2]) if len(sys.argv) > 2 else 576
head, n_out = ("pose", 9) if cin != 9 else ("logits", 1)
bb = eng.Backbone("vanilla_resnet34", cin, head, n_out, syn.make_state_dict("vanilla_resnet34", cin, head, n_out))
x = eng.padded_nhwc(b, 240, 320, bb.c_in_p, bb.in_border, "cuda")
out = torch.empty(b, n_out, device="cuda")
bb.forward(x, b, 240, 320, out)
torch.cuda.synchronize()
eng.profile_begin()
for _ in range(3):
    bb.forward(x, b, 240, 320, out)
prof = eng.profile_end()
tot = sum(v["ms"] for v in prof.values())
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0.0
    print(f"{k:60s} {v['launches']:4d} launches {v['ms'] / 3:8.3f} ms/fwd {100 * v['ms'] / tot:5.1f}%  {tf:6.1f} TFLOP/s")
