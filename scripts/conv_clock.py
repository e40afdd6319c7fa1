"""Effective shader clock INSIDE the conv kernels (profiling build, MP_ENGINE_LIB=scripts/microbench/_build/libmp_engine_prof.so):
sum over workgroups of s_memtime cycles / s_memrealtime ticks (100 MHz) during the K loop, for random and for all-zero data."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from megapose6d_amd import _lib, engine as eng
from tests.support import synthetic as syn

lib = _lib.load()
lib.mp_conv_prof_read.restype = C.c_int
bb = eng.Backbone("vanilla_resnet34", 27, "pose", 9, syn.make_state_dict("vanilla_resnet34", 27, "pose", 9))
b = 576
for name in ("random", "zeros", "random"):
    x = eng.padded_nhwc(b, 240, 320, bb.c_in_p, bb.in_border, "cuda")
    if name == "random":
        eng.padded_view(x, b, 240, 320, bb.c_in_p, bb.in_border)[..., :27] = torch.rand(b, 240, 320, 27, device="cuda")
    out = torch.empty(b, 9, device="cuda")
    bb.forward(x, b, 240, 320, out)
    buf = (C.c_ulonglong * 10)()
    lib.mp_conv_prof_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        bb.forward(x, b, 240, 320, out)
    e1.record()
    torch.cuda.synchronize()
    lib.mp_conv_prof_read(buf, 0)
    ms = e0.elapsed_time(e1) / 3
    mhz = buf[0] / max(buf[1], 1) * 100.0
    tf = bb.flops(b, 240, 320) / ms / 1e9
    print(f"{name:6s} input: {ms:.2f} ms/forward = {tf:.1f} TFLOP/s; effective shader clock inside the conv K loops {mhz:.0f} MHz "
          f"-> fp32 MFMA peak at that clock {157.3 * mhz / 2400:.1f} TFLOP/s, conv stack at {100 * tf / (157.3 * mhz / 2400):.1f}% of it", flush=True)
