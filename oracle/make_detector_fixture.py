"""Writes the detector parity fixture: the CPU oracle's (oracle/mask_rcnn.py) intermediates and outputs for synthetic weights and
images that BOTH sides regenerate from a hash (no 44 M-parameter file travels).  TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_detector_fixture [out_dir]        (default tests/_build/)

Consumers: scripts/microbench/native_detector_check.cpp (torch-free, runs in seconds on the GPU box) and
tests/test_gpu_zz_detector.py.  Record format (little endian): u32 name_len, name, u32 dtype (0 = f32, 1 = i32), u32 ndim,
i64 dims[ndim], data.
"""
from __future__ import annotations

import struct
import sys
import time
from pathlib import Path

import numpy as np
import torch

from . import mask_rcnn as om

CASES = {
    # name: (n_images, H, W, min_size, max_size, n_classes)
    "native": (1, 192, 256, 192, 256, 5),     # no resize
    "resized": (1, 150, 200, 192, 256, 5),    # x1.28 (float32: 1.27999997) bilinear resize to 191 x 255, padded to 192 x 256
    "batch2": (2, 160, 224, 160, 224, 7),     # two images per call, 7 classes
}


def write_records(path: Path, recs) -> None:
    with open(path, "wb") as f:
        for name, arr in recs:
            a = np.ascontiguousarray(arr)
            if a.dtype in (np.int64, np.int32, np.bool_):
                a, code = a.astype(np.int32), 1
            else:
                a, code = a.astype(np.float32), 0
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<II", code, a.ndim) + struct.pack(f"<{a.ndim}q", *a.shape))
            f.write(a.tobytes())


def read_records(path: Path) -> dict:
    out, buf, o = {}, Path(path).read_bytes(), 0
    while o < len(buf):
        (nl,) = struct.unpack_from("<I", buf, o); o += 4
        name = buf[o : o + nl].decode(); o += nl
        code, nd = struct.unpack_from("<II", buf, o); o += 8
        dims = struct.unpack_from(f"<{nd}q", buf, o); o += 8 * nd
        n = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(buf, dtype=np.int32 if code else np.float32, count=n, offset=o).reshape(dims)
        o += 4 * n
    return out


def make_case(name: str, out_dir: Path, threads: int = 16) -> Path:
    n, H, W, mn, mx, C = CASES[name]
    torch.set_num_threads(threads)
    sd = om.synthetic_state_dict(C)
    images = om.synthetic_images(n, H, W)
    t0 = time.time()
    out, dbg = om.mask_rcnn_forward(sd, list(images), mn, mx, return_intermediates=True)
    print(f"[fixture {name}] oracle forward {time.time() - t0:.1f} s, {[len(o['boxes']) for o in out]} detections", file=sys.stderr)
    recs = [("config", np.array([n, H, W, mn, mx, C], np.int32)), ("batch", dbg["batch"].numpy())]
    for l, f in enumerate(dbg["feats"]):
        recs.append((f"P{l + 2}", f.permute(0, 2, 3, 1).numpy()))   # NHWC like the engine
    R = om.RPN_POST_NMS_TOP_N
    props = np.zeros((n, R, 4), np.float32)
    pcnt = np.zeros((n,), np.int32)
    for i, p in enumerate(dbg["proposals"]):
        props[i, : len(p)] = p.numpy()
        pcnt[i] = len(p)
    recs += [("proposals", props), ("proposal_counts", pcnt), ("proposal_scores", np.stack([np.pad(s.numpy(), (0, R - len(s))) for s in dbg["rpn"]["scores"]]))]
    recs += [("class_logits", dbg["class_logits"].numpy()), ("box_regression", dbg["box_regression"].numpy())]
    D = om.BOX_DETECTIONS_PER_IMG
    boxes, scores, labels, cnt = np.zeros((n, D, 4), np.float32), np.zeros((n, D), np.float32), np.zeros((n, D), np.int32), np.zeros((n,), np.int32)
    m28 = np.zeros((n, D, 28, 28), np.float32)
    pasted = np.zeros((n, 8, H, W), np.float32)
    for i, o in enumerate(out):
        k = len(o["boxes"])
        cnt[i] = k
        boxes[i, :k], scores[i, :k], labels[i, :k] = o["boxes"].numpy(), o["scores"].numpy(), o["labels"].numpy()
        m28[i, :k] = o["masks28"][:, 0].numpy()
        pasted[i, : min(k, 8)] = o["masks"][:8, 0].numpy()
    recs += [("boxes", boxes), ("scores", scores), ("labels", labels), ("counts", cnt), ("masks28", m28), ("masks_pasted_first8", pasted)]
    out_dir.mkdir(parents=True, exist_ok=True)
    path = out_dir / f"detector_fixture_{name}.bin"
    write_records(path, recs)
    print(f"[fixture {name}] {path} ({path.stat().st_size / 1e6:.1f} MB)", file=sys.stderr)
    return path


if __name__ == "__main__":
    d = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parent.parent / "tests" / "_build"
    for c in CASES:
        make_case(c, d)


def make_golden(fixture: Path, out: Path) -> None:
    """compact, committed digest of a fixture (tests/golden/detector_<case>.npz, ~100 KB): outputs in full, intermediates
    sub-sampled -- what the GPU test compares the engine with when the full fixture has not been generated"""
    r = read_records(fixture)
    g = {"config": r["config"], "boxes": r["boxes"], "scores": r["scores"], "labels": r["labels"], "counts": r["counts"],
         "proposals_first64": r["proposals"][:, :64], "proposal_scores_first64": r["proposal_scores"][:, :64],
         "proposal_counts": r["proposal_counts"], "class_logits_first64": r["class_logits"][:64],
         "box_regression_first64": r["box_regression"][:64], "masks28_first4": r["masks28"][:, :4]}
    for l in range(2, 7):
        f = r[f"P{l}"]
        g[f"P{l}_sub"] = f[:, ::4, ::4, ::16].copy()          # every 4th pixel, every 16th channel
        g[f"P{l}_absmean"] = np.array([np.abs(f).mean()], np.float32)
    out.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(out, **g)
    print(f"[golden] {out} ({out.stat().st_size / 1e3:.0f} KB)", file=sys.stderr)
