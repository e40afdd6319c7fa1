"""CPU: the detector row without a GPU -- the oracle's building blocks against known answers, the engine's
host-side checkpoint layout against the oracle's, the parameter tree of DetectorMaskRCNN, and the oracle pinned to its committed
golden outputs (tests/golden/detector_*.npz, written by oracle/make_detector_fixture.py)."""
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).resolve().parent / "golden"


def test_engine_checkpoint_layout_equals_the_oracles_and_hosts_a_reference_state_dict():
    from megapose6d_amd import engine as eng
    from megapose6d_amd.mask_rcnn import DetectorMaskRCNN
    from oracle import mask_rcnn as om

    for n_classes in (2, 22):
        assert eng.DetectorNet.state_spec(n_classes) == [(n, tuple(s)) for n, s in om.state_spec(n_classes)]
    m = DetectorMaskRCNN(input_resize=(480, 640), n_classes=22)
    sd = {n: torch.zeros(s) for n, s in om.state_spec(22)}
    sd["backbone.body.bn1.num_batches_tracked"] = torch.tensor(0)   # BatchNorm2d-style checkpoints carry these counters
    assert not m.load_state_dict(sd, strict=True).missing_keys
    assert sorted(m.state_dict().keys()) == sorted(n for n, _ in om.state_spec(22))
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == dict(om.state_spec(22))
    # torchvision's layout: conv / linear tensors are parameters, FrozenBatchNorm2d terms are buffers
    params = {n for n, _ in m.named_parameters()}
    assert "backbone.body.conv1.weight" in params and "roi_heads.box_head.fc6.bias" in params
    assert "backbone.body.layer1.0.bn3.running_var" not in params and "backbone.body.layer2.0.downsample.1.weight" not in params
    assert m.min_size == 480 and m.max_size == 640
    with pytest.raises(NotImplementedError):
        m.train()(images=[torch.zeros(3, 8, 8)])
    with pytest.raises(AssertionError):
        DetectorMaskRCNN(backbone_str="resnet101-fpn")


def test_oracle_building_blocks_known_answers():
    from oracle import mask_rcnn as om

    # anchor_utils.py generate_anchors: the published base anchors of scale 32, ratios (0.5, 1, 2)
    assert om.base_anchors(32).tolist() == [[-23.0, -11.0, 23.0, 11.0], [-16.0, -16.0, 16.0, 16.0], [-11.0, -23.0, 11.0, 23.0]]
    a = om.grid_anchors([(2, 3)], (64, 96))[0]
    assert a.shape == (18, 4) and a[0].tolist() == [-23.0, -11.0, 23.0, 11.0] and a[3].tolist() == [9.0, -11.0, 55.0, 11.0]   # stride 32 in x
    assert a[9].tolist() == [-23.0, 21.0, 23.0, 43.0]                                                                     # second row: +32 in y
    # BoxCoder: zero deltas give the box back, unit dw doubles-by-e the width about the centre, dw is clamped at log(1000/16)
    b = torch.tensor([[10.0, 20.0, 30.0, 60.0]])
    assert torch.allclose(om.decode_boxes(torch.zeros(1, 4), b, (1, 1, 1, 1)), b)
    d = om.decode_boxes(torch.tensor([[0.0, 0.0, 1.0, 0.0]]), b, (1, 1, 1, 1))[0]
    assert abs((d[2] - d[0]).item() - 20 * np.e) < 1e-4 and abs(((d[0] + d[2]) / 2).item() - 20) < 1e-5
    big = om.decode_boxes(torch.tensor([[0.0, 0.0, 50.0, 0.0]]), b, (1, 1, 1, 1))[0]
    assert abs((big[2] - big[0]).item() - 20 * 1000 / 16) < 1e-2
    assert torch.allclose(om.decode_boxes(torch.tensor([[10.0, 0.0, 0.0, 0.0]]), b, (10, 10, 5, 5))[0, 0], torch.tensor(30.0))   # dx / 10 * width
    # NMS: greedy, IoU > thr suppresses, result ordered by decreasing score; brute-force check on random boxes
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(60, 2, generator=g) * 50
    boxes = torch.cat([xy, xy + torch.rand(60, 2, generator=g) * 30 + 1], 1)
    scores = torch.rand(60, generator=g)
    keep = om.nms(boxes, scores, 0.5).tolist()
    order = scores.argsort(descending=True).tolist()
    ref = []
    for i in order:
        def iou(p, q):
            iw = max(0.0, min(p[2], q[2]) - max(p[0], q[0])); ih = max(0.0, min(p[3], q[3]) - max(p[1], q[1]))
            inter = iw * ih
            return inter / ((p[2] - p[0]) * (p[3] - p[1]) + (q[2] - q[0]) * (q[3] - q[1]) - inter)
        if all(iou(boxes[i].tolist(), boxes[j].tolist()) <= 0.5 for j in ref):
            ref.append(i)
    assert keep == ref
    lab = torch.randint(0, 3, (60,), generator=g)
    kb = om.batched_nms(boxes, scores, lab, 0.5)
    assert sorted(kb.tolist()) == sorted(sum([[int(torch.where(lab == c)[0][k]) for k in om.nms(boxes[lab == c], scores[lab == c], 0.5)] for c in range(3)], []))
    assert (scores[kb][:-1] >= scores[kb][1:]).all()
    # paste_masks_in_image: a constant mask fills its (1-pixel-padded, truncated) box and nothing else
    m = om.paste_masks(torch.ones(1, 1, 28, 28), torch.tensor([[4.0, 6.0, 12.0, 16.0]]), (24, 32))
    assert m.shape == (1, 1, 24, 32) and m[0, 0, 11, 8] == 1.0 and m[0, 0, 0, 0] == 0 and m[0, 0, 23, 31] == 0
    ys, xs = torch.nonzero(m[0, 0] > 0, as_tuple=True)
    assert xs.min() >= 3 and xs.max() <= 13 and ys.min() >= 5 and ys.max() <= 17
    # transform: the resize ratio is a float32 quotient (192 / 150 -> 191 rows), batches are padded to multiples of 32
    batch, sizes, orig = om.transform_images([torch.rand(3, 150, 200)], 192, 256)
    assert sizes == [(191, 255)] and orig == [(150, 200)] and batch.shape == (1, 3, 192, 256) and batch[0, :, 191].abs().max() == 0
    batch, sizes, _ = om.transform_images([torch.rand(3, 96, 128)], 96, 128)
    assert sizes == [(96, 128)] and batch.shape == (1, 3, 96, 128)


def test_oracle_is_pinned_to_its_golden_outputs():
    """the committed goldens are what the GPU test compares the engine with; the oracle must still produce them"""
    from oracle import mask_rcnn as om

    g = np.load(GOLD / "detector_native.npz")
    n, H, W, mn, mx, C = (int(v) for v in g["config"])
    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 1 else 16))
    out, dbg = om.mask_rcnn_forward(om.synthetic_state_dict(C), list(om.synthetic_images(n, H, W)), mn, mx, return_intermediates=True)
    k = int(g["counts"][0])
    assert len(out[0]["boxes"]) == k
    assert np.abs(out[0]["boxes"].numpy() - g["boxes"][0, :k]).max() < 1e-3 and np.abs(out[0]["scores"].numpy() - g["scores"][0, :k]).max() < 1e-5
    assert (out[0]["labels"].numpy() == g["labels"][0, :k]).all()
    for l in range(2, 7):
        f = dbg["feats"][l - 2].permute(0, 2, 3, 1).numpy()
        assert np.abs(f[:, ::4, ::4, ::16] - g[f"P{l}_sub"]).max() < 1e-4
    assert np.abs(out[0]["masks28"][:4, 0].numpy() - g["masks28_first4"][0]).max() < 1e-5


def test_native_checker_generates_the_oracles_synthetic_network(tmp_path):
    """scripts/microbench/native_detector_check.cpp (the torch-free parity runner used on the GPU box) regenerates weights and images
    from the same hash as oracle.mask_rcnn.synthetic_*: its --checksums mode (host only) must agree tensor by tensor, bit for bit"""
    import shutil
    import subprocess

    root = Path(__file__).resolve().parent.parent
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).is_file() or not (root / "megapose6d_amd" / "libmp_engine.so").is_file():
        pytest.skip("needs hipcc and the built engine library")
    from oracle import mask_rcnn as om

    exe = tmp_path / "native_detector_check"
    subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", f"-I{root / 'include'}", str(root / "scripts/microbench/native_detector_check.cpp"),
                    "-o", str(exe), f"-L{root / 'megapose6d_amd'}", "-lmp_engine", f"-Wl,-rpath,{root / 'megapose6d_amd'}"], check=True, timeout=300)
    lines = subprocess.run([str(exe), "--checksums", "4"], check=True, capture_output=True, text=True, timeout=300).stdout.split("\n")
    rows = [l.split() for l in lines if l.strip()]

    def ck(a):
        b = np.ascontiguousarray(a, np.float32).reshape(-1).view(np.uint32).astype(np.uint64)
        k = (np.arange(b.size, dtype=np.uint64) % np.uint64(251)) + np.uint64(1)
        with np.errstate(over="ignore"):
            return int((b * k).sum(dtype=np.uint64))

    spec = om.state_spec(4)
    assert len(rows) == len(spec) + 1
    for (name, shape), row in zip(spec, rows):
        v = om.synthetic_tensor(name, shape)
        assert row[0] == name and int(row[1]) == v.size and int(row[2]) == ck(v), name
    assert int(rows[-1][2]) == ck(om.synthetic_images(1, 24, 32).numpy())
