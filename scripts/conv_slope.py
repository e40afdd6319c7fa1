"""Where does the conv kernel lose time?  Launch grids of exactly R full rounds of resident workgroups and vary the K length:
time(K) = a + b*K per round -> `b` is the main-loop cost per 32-float chunk (ideal: 64 MFMAs x 64 cycles per wave at 2 waves/SIMD),
`a` the per-tile prologue + epilogue + launch cost.  Prints cycles (at 2.4 GHz) per chunk per workgroup-pair and the overhead."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from megapose6d_amd import engine as eng

n_cu = eng.device_info()[0]
resident = 2 * n_cu
GHZ = 2.4


def run(Cin, Cout, rounds, iters=8, residual=True):
    n_nb = max(1, Cout // 128)
    tiles_m = rounds * resident // n_nb
    H = W = 16
    N = tiles_m * 128 // (H * W)
    x = eng.padded_nhwc(N, H, W, Cin, 1, "cuda")
    eng.padded_view(x, N, H, W, Cin, 1)[:] = torch.randn(N, H, W, Cin, device="cuda")
    w = np.random.RandomState(0).randn(Cout, Cin, 3, 3).astype(np.float32) * 0.05
    wp = torch.from_numpy(eng.conv_pack_weights(w, Cin, None)).cuda()
    y = eng.padded_nhwc(N, H, W, Cout, 1, "cuda")
    res = eng.padded_nhwc(N, H, W, Cout, 1, "cuda") if residual else None
    bias = torch.zeros(Cout, device="cuda")
    for _ in range(2):
        eng.conv2d_nhwc(x, N, H, W, Cin, 1, wp, bias, Cout, 3, 1, 1, y, 1, residual=res, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        eng.conv2d_nhwc(x, N, H, W, Cin, 1, wp, bias, Cout, 3, 1, 1, y, 1, residual=res, relu=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * N * H * W * Cout * 9 * Cin
    return ms, fl / ms / 1e9


for Cout in (64, 128, 256):
    for rounds in (1, 4):
        pts = []
        for Cin in (64, 128, 256, 512):
            ms, tf = run(Cin, Cout, rounds)
            chunks = 9 * Cin // 32
            pts.append((chunks, ms))
            print(f"Cout={Cout:3d} rounds={rounds} Cin={Cin:3d} chunks={chunks:3d}: {ms * 1e3:8.1f} us  {tf:6.1f} TFLOP/s", flush=True)
        c = np.array([p[0] for p in pts], float)
        t = np.array([p[1] for p in pts], float) * 1e3  # us
        b, a = np.polyfit(c, t, 1)
        mfma_per_chunk = 64 if Cout >= 128 else 32
        ideal = mfma_per_chunk * 64 * 2 / (GHZ * 1e3) * rounds   # us per chunk: 2 co-resident waves share a SIMD
        print(f"  -> slope {b:.3f} us/chunk (ideal {ideal:.3f}: main loop at {100 * ideal / b:.1f}%), intercept {a:.1f} us per launch "
              f"= {a / rounds:.1f} us per round", flush=True)
