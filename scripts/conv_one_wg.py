"""One workgroup per CU vs two: the same conv launches (Cout = 256, Cin = 256 / 512, exactly R rounds of resident workgroups) with
MP_CONV_LDS_PAD_KB=0 (two co-resident workgroups per CU) and =20 (one).  Run twice, once per setting (the knob is read once)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ctypes as C

from megapose6d_amd import _lib
from megapose6d_amd import engine as eng

lib = _lib.load()
try:
    lib.mp_conv_prof_read.restype = C.c_int
    lib.mp_conv_prof_read((C.c_ulonglong * 10)(), 1)
except AttributeError:
    pass
n_cu = eng.device_info()[0]
per_cu = 1 if int(os.environ.get("MP_CONV_LDS_PAD_KB", "0")) >= 12 else 2
for Cin in (256, 512):
    for rounds in (1, 4):
        Cout, H, W = 256, 16, 16
        tiles_m = rounds * per_cu * n_cu // 2
        N = tiles_m * 128 // (H * W)
        x = eng.padded_nhwc(N, H, W, Cin, 1, "cuda")
        eng.padded_view(x, N, H, W, Cin, 1)[:] = torch.randn(N, H, W, Cin, device="cuda")
        w = np.random.RandomState(0).randn(Cout, Cin, 3, 3).astype(np.float32) * 0.05
        wp = torch.from_numpy(eng.conv_pack_weights(w, Cin, None)).cuda()
        y = eng.padded_nhwc(N, H, W, Cout, 1, "cuda")
        bias = torch.zeros(Cout, device="cuda")
        for _ in range(3):
            eng.conv2d_nhwc(x, N, H, W, Cin, 1, wp, bias, Cout, 3, 1, 1, y, 1, relu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.conv2d_nhwc(x, N, H, W, Cin, 1, wp, bias, Cout, 3, 1, 1, y, 1, relu=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        if hasattr(lib, "mp_conv_prof_read"):
            buf = (C.c_ulonglong * 10)()
            lib.mp_conv_prof_read(buf, 1)
            tot = sum(buf[2 + k] for k in range(5)) or 1
            seg = " | per-wave cycles: " + ", ".join(f"{n} {100.0 * buf[2 + k] / tot:.1f}%" for k, n in enumerate(
                ["load issue", "MFMA groups 0-1", "global-load wait + LDS writes", "MFMA groups 2-3", "barrier"]))
        else:
            seg = ""
        chunks = 9 * Cin // 32
        cyc = ms * 1e-3 * 2.367e9 / (rounds * chunks)   # cycles per chunk per round at the measured 2367 MHz
        print(f"{per_cu} WG/CU, Cin={Cin}, {rounds} round(s): {ms * 1e3:8.1f} us, {2.0 * N * H * W * Cout * 9 * Cin / ms / 1e9:6.1f} TFLOP/s, "
              f"{cyc:7.0f} cycles per chunk-round (ideal {4096 * per_cu})" + seg, flush=True)
