"""Micro-benchmark of the backbone executor (fp32 MFMA conv stack): ms/forward and achieved TFLOP/s."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from megapose6d_amd import engine as eng
from tests.support import synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="vanilla_resnet34")
ap.add_argument("--cin", type=int, default=27)
ap.add_argument("--batch", type=int, nargs="+", default=[32, 128, 576])
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
head, n_out = ("pose", 9) if a.cin != 9 else ("logits", 1)
bb = eng.Backbone(a.kind, a.cin, head, n_out, syn.make_state_dict(a.kind, a.cin, head, n_out))
for b in a.batch:
    x = eng.padded_nhwc(b, 240, 320, bb.c_in_p, bb.in_border, "cuda")
    eng.padded_view(x, b, 240, 320, bb.c_in_p, bb.in_border)[..., : a.cin] = torch.rand(b, 240, 320, a.cin, device="cuda")
    out = torch.empty(b, n_out, device="cuda")
    bb.forward(x, b, 240, 320, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        bb.forward(x, b, 240, 320, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    fl = bb.flops(b, 240, 320)
    print(f"{a.kind} cin={a.cin} batch={b}: {ms:.2f} ms/forward, {fl / ms / 1e9:.1f} TFLOP/s ({fl / ms / 1e9 / 157.3 * 100:.1f}% of fp32 MFMA peak), {b / ms * 1e3:.0f} rows/s")
    del x
