"""CPU restatement (plain torch fp32) of the detector network behind the reference's `Detector`.
TEST INFRASTRUCTURE ONLY.

The reference's `DetectorMaskRCNN` (/root/reference/src/megapose/models/mask_rcnn.py:23-46) is torchvision's
`MaskRCNN(resnet_fpn_backbone("resnet50"), num_classes, rpn_anchor_generator=AnchorGenerator(((32,),(64,),(128,),(256,),(512,)),
((0.5, 1.0, 2.0),) * 5), min_size=min(input_resize), max_size=max(input_resize))` with every other hyper-parameter at torchvision's
default; it is driven by `Detector.get_detections` (/root/reference/src/megapose/inference/detector.py:63-136).
torchvision (pinned 0.12.0, conda/environment_full.yaml:14) is NOT under /root/reference and is absent from this image, so this
file restates the published inference algorithm of torchvision 0.12 -- **parity unpinned**:
  models/detection/transform.py   GeneralizedRCNNTransform (normalize, resize, batch to a multiple of 32, postprocess)
  models/resnet.py                ResNet-50 (Bottleneck, stride on the 3x3), ops/misc.py FrozenBatchNorm2d (eps 1e-5)
  ops/feature_pyramid_network.py  FPN + LastLevelMaxPool
  models/detection/anchor_utils.py, rpn.py, _utils.py (BoxCoder), ops/boxes.py (clip, remove_small, nms, batched_nms)
  ops/poolers.py                  MultiScaleRoIAlign + LevelMapper, ops/roi_align (oracle/thirdparty.py)
  models/detection/roi_heads.py   box head / predictor / postprocess_detections, mask head / predictor / maskrcnn_inference,
                                  paste_masks_in_image
State-dict layout = torchvision's (`backbone.body.*`, `backbone.fpn.{inner,layer}_blocks.N.{weight,bias}`, `rpn.head.*`,
`roi_heads.box_head.fc6/fc7`, `roi_heads.box_predictor.*`, `roi_heads.mask_head.mask_fcnN`, `roi_heads.mask_predictor.*`),
which is what a checkpoint of the reference's detector holds.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import thirdparty as tp

IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)
ANCHOR_SIZES = ((32,), (64,), (128,), (256,), (512,))   # models/mask_rcnn.py:29
ASPECT_RATIOS = (0.5, 1.0, 2.0)                          # models/mask_rcnn.py:35
BBOX_XFORM_CLIP = math.log(1000.0 / 16)
RPN_PRE_NMS_TOP_N, RPN_POST_NMS_TOP_N, RPN_NMS_THRESH, RPN_SCORE_THRESH, RPN_MIN_SIZE = 1000, 1000, 0.7, 0.0, 1e-3
BOX_SCORE_THRESH, BOX_NMS_THRESH, BOX_DETECTIONS_PER_IMG, BOX_MIN_SIZE = 0.05, 0.5, 100, 1e-2
RESNET50_BLOCKS = (3, 4, 6, 3)


# ---------------------------------------------------------------------------------------------------------------------------------
# architecture spec: every tensor of the checkpoint, in a fixed order (name, shape).  csrc/detector.hip exposes the same list through
# mp_detector_state_spec; tests compare the two.
# ---------------------------------------------------------------------------------------------------------------------------------
def state_spec(n_classes: int) -> List[Tuple[str, Tuple[int, ...]]]:
    spec: List[Tuple[str, Tuple[int, ...]]] = []

    def bn(prefix, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            spec.append((f"{prefix}.{s}", (c,)))

    B = "backbone.body."
    spec.append((B + "conv1.weight", (64, 3, 7, 7)))
    bn(B + "bn1", 64)
    inplanes = 64
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), RESNET50_BLOCKS)):
        for bi in range(n):
            P = f"{B}layer{li + 1}.{bi}."
            spec.append((P + "conv1.weight", (planes, inplanes, 1, 1)))
            bn(P + "bn1", planes)
            spec.append((P + "conv2.weight", (planes, planes, 3, 3)))
            bn(P + "bn2", planes)
            spec.append((P + "conv3.weight", (planes * 4, planes, 1, 1)))
            bn(P + "bn3", planes * 4)
            if bi == 0:
                spec.append((P + "downsample.0.weight", (planes * 4, inplanes, 1, 1)))
                bn(P + "downsample.1", planes * 4)
            inplanes = planes * 4
    for i, c in enumerate((256, 512, 1024, 2048)):
        spec.append((f"backbone.fpn.inner_blocks.{i}.weight", (256, c, 1, 1)))
        spec.append((f"backbone.fpn.inner_blocks.{i}.bias", (256,)))
        spec.append((f"backbone.fpn.layer_blocks.{i}.weight", (256, 256, 3, 3)))
        spec.append((f"backbone.fpn.layer_blocks.{i}.bias", (256,)))
    A = len(ASPECT_RATIOS)
    spec += [("rpn.head.conv.weight", (256, 256, 3, 3)), ("rpn.head.conv.bias", (256,)),
             ("rpn.head.cls_logits.weight", (A, 256, 1, 1)), ("rpn.head.cls_logits.bias", (A,)),
             ("rpn.head.bbox_pred.weight", (4 * A, 256, 1, 1)), ("rpn.head.bbox_pred.bias", (4 * A,)),
             ("roi_heads.box_head.fc6.weight", (1024, 256 * 7 * 7)), ("roi_heads.box_head.fc6.bias", (1024,)),
             ("roi_heads.box_head.fc7.weight", (1024, 1024)), ("roi_heads.box_head.fc7.bias", (1024,)),
             ("roi_heads.box_predictor.cls_score.weight", (n_classes, 1024)), ("roi_heads.box_predictor.cls_score.bias", (n_classes,)),
             ("roi_heads.box_predictor.bbox_pred.weight", (4 * n_classes, 1024)), ("roi_heads.box_predictor.bbox_pred.bias", (4 * n_classes,))]
    for i in range(1, 5):
        spec += [(f"roi_heads.mask_head.mask_fcn{i}.weight", (256, 256, 3, 3)), (f"roi_heads.mask_head.mask_fcn{i}.bias", (256,))]
    spec += [("roi_heads.mask_predictor.conv5_mask.weight", (256, 256, 2, 2)), ("roi_heads.mask_predictor.conv5_mask.bias", (256,)),
             ("roi_heads.mask_predictor.mask_fcn_logits.weight", (n_classes, 256, 1, 1)),
             ("roi_heads.mask_predictor.mask_fcn_logits.bias", (n_classes,))]
    return spec


# ---------------------------------------------------------------------------------------------------------------------------------
# deterministic synthetic weights shared with the native C++ checker (scripts/microbench/native_detector_check.cpp): value i of a
# tensor = f(FNV-1a(name), i) through splitmix64 -- no file with 44 M parameters has to travel to the GPU box.
# ---------------------------------------------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_unit(seed: int, n: int) -> np.ndarray:
    """n float32 values in [-1, 1): 24 top bits of splitmix64(seed + i * golden) as u * 2^-23 - 1 (exact in float32)"""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64)
        x = (np.uint64(seed) + (i + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    u24 = (z >> np.uint64(40)).astype(np.float32)
    return u24 * np.float32(2.0 ** -23) - np.float32(1.0)


def synthetic_tensor(name: str, shape: Sequence[int]) -> np.ndarray:
    """value rule by tensor role (same in the C++ checker): BN scale in [0.5, 1.5) (x 0.3 for the last BN of a bottleneck so that the
    residual trunk stays O(1) over 16 blocks), running_var in [0.5, 1.5), running_mean / biases in [-0.1, 0.1), weights uniform with
    variance 1 / fan_in (x 2 for layers followed by a ReLU would be He; the plain 1 / fan_in keeps the logits O(1))."""
    n = int(np.prod(shape))
    u = hash_unit(fnv1a(name), n)
    is_bn = ".bn" in name or "downsample.1." in name
    if name.endswith(".running_var"):
        v = u * np.float32(0.5) + np.float32(1.0)
    elif is_bn and name.endswith(".weight"):
        v = u * np.float32(0.5) + np.float32(1.0)
        if ".bn3." in name:
            v = v * np.float32(0.3)
    elif name.endswith(".running_mean") or name.endswith(".bias"):
        v = u * np.float32(0.1)
    else:
        fan_in = n // int(shape[0])
        v = u * np.float32(math.sqrt(3.0 / fan_in))
    return v.reshape(shape).astype(np.float32)


def synthetic_state_dict(n_classes: int) -> Dict[str, torch.Tensor]:
    return {name: torch.from_numpy(synthetic_tensor(name, shape)) for name, shape in state_spec(n_classes)}


def synthetic_images(n: int, h: int, w: int, seed: int = 7) -> torch.Tensor:
    """[n,3,h,w] in [0,1): smooth blobs + hash noise (same formula in the C++ checker)"""
    u = hash_unit(fnv1a(f"images/{seed}"), n * 3 * h * w).reshape(n, 3, h, w)
    ys = np.arange(h, dtype=np.float32)[:, None] / np.float32(h)
    xs = np.arange(w, dtype=np.float32)[None, :] / np.float32(w)
    base = (np.float32(0.5) + np.float32(0.25) * ys - np.float32(0.2) * xs).astype(np.float32)
    img = base[None, None] + np.float32(0.25) * u
    return torch.from_numpy(np.clip(img, 0.0, 0.999).astype(np.float32))


# ---------------------------------------------------------------------------------------------------------------------------------
# network
# ---------------------------------------------------------------------------------------------------------------------------------
def _bn(sd, p, x):  # FrozenBatchNorm2d (ops/misc.py): scale = w * rsqrt(var + eps), bias = b - mean * scale, eps = 1e-5
    scale = sd[p + ".weight"] * (sd[p + ".running_var"] + 1e-5).rsqrt()
    shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


def resnet50_fpn(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> List[torch.Tensor]:
    """-> [P2, P3, P4, P5, pool] (FPN outputs "0".."3","pool"), 256 channels each"""
    B = "backbone.body."
    x = F.relu(_bn(sd, B + "bn1", F.conv2d(x, sd[B + "conv1.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, n in enumerate(RESNET50_BLOCKS):
        for bi in range(n):
            P = f"{B}layer{li + 1}.{bi}."
            stride = 2 if (bi == 0 and li > 0) else 1
            idn = x
            o = F.relu(_bn(sd, P + "bn1", F.conv2d(x, sd[P + "conv1.weight"])))
            o = F.relu(_bn(sd, P + "bn2", F.conv2d(o, sd[P + "conv2.weight"], stride=stride, padding=1)))
            o = _bn(sd, P + "bn3", F.conv2d(o, sd[P + "conv3.weight"]))
            if bi == 0:
                idn = _bn(sd, P + "downsample.1", F.conv2d(x, sd[P + "downsample.0.weight"], stride=stride))
            x = F.relu(o + idn)
        feats.append(x)
    F_ = "backbone.fpn."
    last = F.conv2d(feats[3], sd[F_ + "inner_blocks.3.weight"], sd[F_ + "inner_blocks.3.bias"])
    outs = [F.conv2d(last, sd[F_ + "layer_blocks.3.weight"], sd[F_ + "layer_blocks.3.bias"], padding=1)]
    for i in (2, 1, 0):
        lat = F.conv2d(feats[i], sd[F_ + f"inner_blocks.{i}.weight"], sd[F_ + f"inner_blocks.{i}.bias"])
        last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
        outs.insert(0, F.conv2d(last, sd[F_ + f"layer_blocks.{i}.weight"], sd[F_ + f"layer_blocks.{i}.bias"], padding=1))
    outs.append(F.max_pool2d(outs[-1], 1, 2, 0))
    return outs


def transform_images(images: Sequence[torch.Tensor], min_size: int, max_size: int):
    """GeneralizedRCNNTransform.forward (eval): -> (batched [n,3,Hp,Wp], image_sizes after resize, original sizes)"""
    mean, std = torch.tensor(IMAGE_MEAN).reshape(3, 1, 1), torch.tensor(IMAGE_STD).reshape(3, 1, 1)
    out, sizes, orig = [], [], []
    for im in images:
        h, w = im.shape[-2:]
        orig.append((h, w))
        x = (im - mean) / std
        # transform.py _resize_image_and_masks: the ratio is a float32 tensor division and reaches interpolate through .item()
        # (192 / 150 -> 1.2799999713897705 -> floor(150 * scale) = 191 rows, not 192)
        smin, smax = torch.tensor(float(min(h, w)), dtype=torch.float32), torch.tensor(float(max(h, w)), dtype=torch.float32)
        scale = torch.min(float(min_size) / smin, float(max_size) / smax).item()
        x = F.interpolate(x[None], scale_factor=scale, mode="bilinear", recompute_scale_factor=True, align_corners=False)[0]
        sizes.append(tuple(x.shape[-2:]))
        out.append(x)
    Hp = int(math.ceil(max(s[0] for s in sizes) / 32) * 32)
    Wp = int(math.ceil(max(s[1] for s in sizes) / 32) * 32)
    batch = torch.zeros(len(out), 3, Hp, Wp)
    for i, x in enumerate(out):
        batch[i, :, : x.shape[1], : x.shape[2]] = x
    return batch, sizes, orig


def base_anchors(size: int) -> torch.Tensor:
    ar = torch.tensor(ASPECT_RATIOS)
    h_r = torch.sqrt(ar)
    w_r = 1 / h_r
    ws = (w_r[:, None] * torch.tensor([float(size)])[None, :]).view(-1)
    hs = (h_r[:, None] * torch.tensor([float(size)])[None, :]).view(-1)
    return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()


def grid_anchors(feat_shapes: Sequence[Tuple[int, int]], image_hw: Tuple[int, int]) -> List[torch.Tensor]:
    out = []
    for (gh, gw), (size,) in zip(feat_shapes, ANCHOR_SIZES):
        sh, sw = image_hw[0] // gh, image_hw[1] // gw
        sx = torch.arange(0, gw, dtype=torch.float32) * sw
        sy = torch.arange(0, gh, dtype=torch.float32) * sh
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
        out.append((shifts.view(-1, 1, 4) + base_anchors(size).view(1, -1, 4)).reshape(-1, 4))
    return out


def decode_boxes(deltas: torch.Tensor, boxes: torch.Tensor, weights: Tuple[float, float, float, float]) -> torch.Tensor:
    """BoxCoder.decode_single (_utils.py): deltas [n, 4k] for boxes [n, 4] -> [n, 4k]"""
    wx, wy, ww, wh = weights
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw, dh = torch.clamp(deltas[:, 2::4] / ww, max=BBOX_XFORM_CLIP), torch.clamp(deltas[:, 3::4] / wh, max=BBOX_XFORM_CLIP)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = torch.exp(dw) * widths[:, None]
    ph = torch.exp(dh) * heights[:, None]
    x1, y1, x2, y2 = pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph
    return torch.stack((x1, y1, x2, y2), dim=2).flatten(1)


def clip_boxes(boxes: torch.Tensor, hw: Tuple[int, int]) -> torch.Tensor:
    b = boxes.clone()
    b[..., 0::2] = b[..., 0::2].clamp(min=0, max=hw[1])
    b[..., 1::2] = b[..., 1::2].clamp(min=0, max=hw[0])
    return b


def nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """torchvision/csrc/ops/cpu/nms_kernel.cpp: greedy, descending score (stable), suppress IoU > thr"""
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.long)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order]
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    n = b.shape[0]
    dead = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        xx1 = torch.maximum(b[i, 0], b[i + 1 :, 0])
        yy1 = torch.maximum(b[i, 1], b[i + 1 :, 1])
        xx2 = torch.minimum(b[i, 2], b[i + 1 :, 2])
        yy2 = torch.minimum(b[i, 3], b[i + 1 :, 3])
        inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
        ovr = inter / (areas[i] + areas[i + 1 :] - inter)
        dead[i + 1 :] |= ovr > thr
    return order[torch.tensor(keep, dtype=torch.long)]


def batched_nms(boxes, scores, idxs, thr) -> torch.Tensor:
    """ops/boxes.py _batched_nms_vanilla: per-category NMS, result sorted by decreasing score"""
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for c in torch.unique(idxs):
        ci = torch.where(idxs == c)[0]
        keep_mask[ci[nms(boxes[ci], scores[ci], thr)]] = True
    ki = torch.where(keep_mask)[0]
    return ki[torch.sort(scores[ki], descending=True, stable=True).indices]


def rpn_proposals(sd, feats: List[torch.Tensor], image_sizes, padded_hw) -> Tuple[List[torch.Tensor], Dict[str, object]]:
    obj, dlt = [], []
    for f in feats:
        t = F.relu(F.conv2d(f, sd["rpn.head.conv.weight"], sd["rpn.head.conv.bias"], padding=1))
        o = F.conv2d(t, sd["rpn.head.cls_logits.weight"], sd["rpn.head.cls_logits.bias"])     # [N, A, H, W]
        d = F.conv2d(t, sd["rpn.head.bbox_pred.weight"], sd["rpn.head.bbox_pred.bias"])        # [N, 4A, H, W]
        N, A, H, W = o.shape
        obj.append(o.permute(0, 2, 3, 1).reshape(N, -1))                                        # (h, w, a)
        dlt.append(d.view(N, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, 4))
    anchors = grid_anchors([tuple(f.shape[-2:]) for f in feats], padded_hw)
    n_per = [a.shape[0] for a in anchors]
    all_anchors = torch.cat(anchors)
    objectness, deltas = torch.cat(obj, 1), torch.cat(dlt, 1)
    N = objectness.shape[0]
    proposals = decode_boxes(deltas.reshape(-1, 4), all_anchors.repeat(N, 1), (1.0, 1.0, 1.0, 1.0)).view(N, -1, 4)
    levels = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(n_per)])
    out, out_scores = [], []
    for n in range(N):
        idx, off = [], 0
        for k in n_per:
            top = torch.sort(objectness[n, off : off + k], descending=True, stable=True).indices[: min(RPN_PRE_NMS_TOP_N, k)]
            idx.append(top + off)
            off += k
        idx = torch.cat(idx)
        b = clip_boxes(proposals[n, idx], image_sizes[n])
        s = torch.sigmoid(objectness[n, idx])
        lv = levels[idx]
        ws, hs = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
        keep = torch.where((ws >= RPN_MIN_SIZE) & (hs >= RPN_MIN_SIZE))[0]
        b, s, lv = b[keep], s[keep], lv[keep]
        keep = torch.where(s >= RPN_SCORE_THRESH)[0]
        b, s, lv = b[keep], s[keep], lv[keep]
        keep = batched_nms(b, s, lv, RPN_NMS_THRESH)[:RPN_POST_NMS_TOP_N]
        out.append(b[keep])
        out_scores.append(s[keep])
    return out, {"objectness": objectness, "deltas": deltas, "scores": out_scores}


def multiscale_roi_align(feats4: List[torch.Tensor], boxes: List[torch.Tensor], image_sizes, out_size: int) -> torch.Tensor:
    """MultiScaleRoIAlign(["0".."3"], out_size, sampling_ratio=2) (ops/poolers.py)"""
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b], 1) for i, b in enumerate(boxes)])
    oh, ow = max(s[0] for s in image_sizes), max(s[1] for s in image_sizes)
    scales = []
    for f in feats4:
        s1 = 2.0 ** float(torch.tensor(float(f.shape[-2]) / float(oh)).log2().round())
        s2 = 2.0 ** float(torch.tensor(float(f.shape[-1]) / float(ow)).log2().round())
        assert s1 == s2
        scales.append(s1)
    k_min, k_max = -math.log2(scales[0]), -math.log2(scales[-1])
    area = (rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])
    lvl = torch.floor(4 + torch.log2(torch.sqrt(area) / 224) + torch.tensor(1e-6))
    lvl = (torch.clamp(lvl, min=k_min, max=k_max).long() - int(k_min))
    out = torch.zeros(len(rois), feats4[0].shape[1], out_size, out_size)
    for li, (f, sc) in enumerate(zip(feats4, scales)):
        sel = torch.where(lvl == li)[0]
        if len(sel):
            out[sel] = tp.roi_align(f, rois[sel], (out_size, out_size), spatial_scale=sc, sampling_ratio=2)
    return out


def box_branch(sd, feats4, proposals, image_sizes):
    x = multiscale_roi_align(feats4, proposals, image_sizes, 7).flatten(1)
    x = F.relu(F.linear(x, sd["roi_heads.box_head.fc6.weight"], sd["roi_heads.box_head.fc6.bias"]))
    x = F.relu(F.linear(x, sd["roi_heads.box_head.fc7.weight"], sd["roi_heads.box_head.fc7.bias"]))
    logits = F.linear(x, sd["roi_heads.box_predictor.cls_score.weight"], sd["roi_heads.box_predictor.cls_score.bias"])
    reg = F.linear(x, sd["roi_heads.box_predictor.bbox_pred.weight"], sd["roi_heads.box_predictor.bbox_pred.bias"])
    return logits, reg


def postprocess_detections(logits, reg, proposals, image_sizes):
    n_cls = logits.shape[-1]
    pred = decode_boxes(reg, torch.cat(proposals), (10.0, 10.0, 5.0, 5.0)).view(len(reg), -1, 4)
    scores = F.softmax(logits, -1)
    res, off = [], 0
    for n, p in enumerate(proposals):
        b = clip_boxes(pred[off : off + len(p)], image_sizes[n])
        s = scores[off : off + len(p)]
        off += len(p)
        lab = torch.arange(n_cls).view(1, -1).expand_as(s)
        b, s, lab = b[:, 1:].reshape(-1, 4), s[:, 1:].reshape(-1), lab[:, 1:].reshape(-1)
        inds = torch.where(s > BOX_SCORE_THRESH)[0]
        b, s, lab = b[inds], s[inds], lab[inds]
        ws, hs = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
        keep = torch.where((ws >= BOX_MIN_SIZE) & (hs >= BOX_MIN_SIZE))[0]
        b, s, lab = b[keep], s[keep], lab[keep]
        keep = batched_nms(b, s, lab, BOX_NMS_THRESH)[:BOX_DETECTIONS_PER_IMG]
        res.append(dict(boxes=b[keep], scores=s[keep], labels=lab[keep]))
    return res


def mask_branch(sd, feats4, dets, image_sizes) -> List[torch.Tensor]:
    boxes = [d["boxes"] for d in dets]
    if sum(len(b) for b in boxes) == 0:
        return [torch.zeros(0, 1, 28, 28) for _ in dets]
    x = multiscale_roi_align(feats4, boxes, image_sizes, 14)
    for i in range(1, 5):
        x = F.relu(F.conv2d(x, sd[f"roi_heads.mask_head.mask_fcn{i}.weight"], sd[f"roi_heads.mask_head.mask_fcn{i}.bias"], padding=1))
    x = F.relu(F.conv_transpose2d(x, sd["roi_heads.mask_predictor.conv5_mask.weight"], sd["roi_heads.mask_predictor.conv5_mask.bias"], stride=2))
    x = F.conv2d(x, sd["roi_heads.mask_predictor.mask_fcn_logits.weight"], sd["roi_heads.mask_predictor.mask_fcn_logits.bias"])
    prob = x.sigmoid()
    labels = torch.cat([d["labels"] for d in dets])
    prob = prob[torch.arange(len(labels)), labels][:, None]
    return list(prob.split([len(b) for b in boxes], dim=0))


def paste_masks(masks: torch.Tensor, boxes: torch.Tensor, hw: Tuple[int, int], padding: int = 1) -> torch.Tensor:
    """roi_heads.py paste_masks_in_image: [n,1,28,28] soft masks -> [n,1,H,W]"""
    M = masks.shape[-1]
    scale = float(M + 2 * padding) / M
    padded = F.pad(masks, (padding,) * 4)
    w_half = (boxes[:, 2] - boxes[:, 0]) * 0.5 * scale
    h_half = (boxes[:, 3] - boxes[:, 1]) * 0.5 * scale
    x_c = (boxes[:, 2] + boxes[:, 0]) * 0.5
    y_c = (boxes[:, 3] + boxes[:, 1]) * 0.5
    be = torch.stack([x_c - w_half, y_c - h_half, x_c + w_half, y_c + h_half], 1).to(torch.int64)
    im_h, im_w = hw
    out = torch.zeros(len(masks), 1, im_h, im_w)
    for i in range(len(masks)):
        b = [int(v) for v in be[i]]
        w, h = max(b[2] - b[0] + 1, 1), max(b[3] - b[1] + 1, 1)
        m = F.interpolate(padded[i : i + 1], size=(h, w), mode="bilinear", align_corners=False)[0, 0]
        x0, x1 = max(b[0], 0), min(b[2] + 1, im_w)
        y0, y1 = max(b[1], 0), min(b[3] + 1, im_h)
        if x1 > x0 and y1 > y0:
            out[i, 0, y0:y1, x0:x1] = m[(y0 - b[1]) : (y1 - b[1]), (x0 - b[0]) : (x1 - b[0])]
    return out


@torch.no_grad()
def mask_rcnn_forward(sd: Dict[str, torch.Tensor], images: Sequence[torch.Tensor], min_size: int, max_size: int, with_masks: bool = True,
                      return_intermediates: bool = False):
    """MaskRCNN.forward in eval mode -> list of dict(boxes, labels, scores, masks [n,1,H,W]) in ORIGINAL image coordinates"""
    batch, sizes, orig = transform_images(images, min_size, max_size)
    feats = resnet50_fpn(sd, batch)
    proposals, rpn_dbg = rpn_proposals(sd, feats, sizes, tuple(batch.shape[-2:]))
    logits, reg = box_branch(sd, feats[:4], proposals, sizes)
    dets = postprocess_detections(logits, reg, proposals, sizes)
    masks28 = mask_branch(sd, feats[:4], dets, sizes) if with_masks else [None] * len(dets)
    out = []
    for d, m, s, o in zip(dets, masks28, sizes, orig):
        rh, rw = torch.tensor(float(o[0])) / torch.tensor(float(s[0])), torch.tensor(float(o[1])) / torch.tensor(float(s[1]))
        b = d["boxes"]
        b = torch.stack((b[:, 0] * rw, b[:, 1] * rh, b[:, 2] * rw, b[:, 3] * rh), dim=1)
        r = dict(boxes=b, labels=d["labels"], scores=d["scores"])
        if with_masks:
            r["masks"] = paste_masks(m, b, o)
            r["masks28"] = m
        out.append(r)
    if return_intermediates:
        return out, dict(batch=batch, image_sizes=sizes, feats=feats, proposals=proposals, rpn=rpn_dbg, class_logits=logits,
                         box_regression=reg, detections_resized=dets)
    return out
