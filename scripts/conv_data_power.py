"""Does the conv kernel's throughput depend on the DATA (power-limited clock)?  Same launch (layer-3 shape, 4 full rounds of
resident workgroups) with all-zero, constant and random inputs/weights."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ctypes as C

from megapose6d_amd import _lib, engine as eng

lib = _lib.load()
PROF = hasattr(lib, "mp_conv_prof_read")   # profiling build (MP_ENGINE_LIB=scripts/microbench/_build/libmp_engine_prof.so): in-kernel clock

n_cu = eng.device_info()[0]
Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Cout = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = W = 16
rounds = 4
n_nblocks = (Cout + 127) // 128 if Cout > 64 else 1
N = rounds * 2 * n_cu // n_nblocks * 128 // (H * W)
print(f"Cin={Cin} Cout={Cout} N={N} ({rounds} rounds of resident workgroups)")
for name in ("zeros", "ones", "randn", "randn", "zeros"):
    x = eng.padded_nhwc(N, H, W, Cin, 1, "cuda")
    if name == "ones":
        eng.padded_view(x, N, H, W, Cin, 1)[:] = 1.0
        w = np.ones((Cout, Cin, 3, 3), np.float32) * 0.01
    elif name == "randn":
        eng.padded_view(x, N, H, W, Cin, 1)[:] = torch.randn(N, H, W, Cin, device="cuda")
        w = np.random.RandomState(0).randn(Cout, Cin, 3, 3).astype(np.float32) * 0.05
    else:
        w = np.zeros((Cout, Cin, 3, 3), np.float32)
    wp = torch.from_numpy(eng.conv_pack_weights(w, Cin, None)).cuda()
    y = eng.padded_nhwc(N, H, W, Cout, 1, "cuda")
    bias = torch.zeros(Cout, device="cuda")
    for _ in range(3):
        eng.conv2d_nhwc(x, N, H, W, Cin, 1, wp, bias, Cout, 3, 1, 1, y, 1, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    buf = (C.c_ulonglong * 10)()
    if PROF:
        lib.mp_conv_prof_read(buf, 1)
    e0.record()
    iters = 20
    for _ in range(iters):
        eng.conv2d_nhwc(x, N, H, W, Cin, 1, wp, bias, Cout, 3, 1, 1, y, 1, relu=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    clk = ""
    if PROF:
        lib.mp_conv_prof_read(buf, 0)
        clk = f", shader clock inside the kernel {buf[0] / max(buf[1], 1) * 100.0:.0f} MHz"
    print(f"{name:6s}: {ms * 1e3:8.1f} us per launch, {2.0 * N * H * W * Cout * 9 * Cin / ms / 1e9:6.1f} TFLOP/s{clk}", flush=True)
