"""CPU restatement (plain torch.nn.functional, fp32) of the reference CNN backbones + heads.
TEST INFRASTRUCTURE ONLY.  Works directly on a reference-layout state_dict.

Follows: /root/reference/src/megapose/models/torchvision_resnet.py:104-120 (BasicBlock.forward),
:297-314 (_forward_impl); /root/reference/src/megapose/models/wide_resnet.py:50-56 (BasicBlockV2.forward),
:102-111 (WideResNet.forward); /root/reference/src/megapose/models/pose_rigid.py:314-334 (net_forward).
Pinned against the reference's own module classes by tests/test_oracle_pinning.py (container only).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

_EPS = 1e-5


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, _EPS)


def _layers(sd, prefix="backbone."):
    out = []
    for s in range(1, 5):
        i = 0
        while f"{prefix}layer{s}.{i}.conv1.weight" in sd:
            out.append((s, i))
            i += 1
    return out


def vanilla_resnet34_features(sd: Dict[str, torch.Tensor], x: torch.Tensor, prefix="backbone.") -> torch.Tensor:
    x = F.conv2d(x, sd[prefix + "conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(sd, prefix + "bn1", x))
    x = F.max_pool2d(x, 3, 2, 1)
    for s, i in _layers(sd, prefix):
        P = f"{prefix}layer{s}.{i}."
        stride = 2 if (i == 0 and s > 1) else 1
        idn = x
        out = F.relu(_bn(sd, P + "bn1", F.conv2d(x, sd[P + "conv1.weight"], stride=stride, padding=1)))
        out = _bn(sd, P + "bn2", F.conv2d(out, sd[P + "conv2.weight"], stride=1, padding=1))
        if P + "downsample.0.weight" in sd:
            idn = _bn(sd, P + "downsample.1", F.conv2d(x, sd[P + "downsample.0.weight"], stride=stride))
        x = F.relu(out + idn)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.linear(x, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])


def wide_resnet_features(sd: Dict[str, torch.Tensor], x: torch.Tensor, prefix="backbone.") -> torch.Tensor:
    x = F.conv2d(x, sd[prefix + "conv1.weight"], stride=2, padding=2)
    x = F.relu(_bn(sd, prefix + "bn1", x))
    x = F.max_pool2d(x, 3, 2, 1)
    for s, i in _layers(sd, prefix):
        P = f"{prefix}layer{s}.{i}."
        stride = 2 if (i == 0 and s > 1) else 1
        a = F.relu(_bn(sd, P + "bn1", x))
        res = F.conv2d(a, sd[P + "downsample.weight"], stride=stride) if P + "downsample.weight" in sd else x
        out = F.conv2d(a, sd[P + "conv1.weight"], stride=stride, padding=1)
        out = F.relu(_bn(sd, P + "bn2", out))
        out = F.conv2d(out, sd[P + "conv2.weight"], stride=1, padding=1)
        x = out + res
    return x.flatten(2).mean(dim=-1)  # pose_rigid.py:326-328


def net_forward(sd: Dict[str, torch.Tensor], backbone_str: str, x: torch.Tensor) -> Dict[str, torch.Tensor]:
    feat = vanilla_resnet34_features(sd, x) if backbone_str == "vanilla_resnet34" else wide_resnet_features(sd, x)   # (any width: shapes come from sd)
    out = {"features": feat}
    if "pose_fc.weight" in sd:
        out["pose"] = F.linear(feat, sd["pose_fc.weight"], sd["pose_fc.bias"])
    if "views_logits_head.weight" in sd:
        out["renderings_logits"] = F.linear(feat, sd["views_logits_head.weight"], sd["views_logits_head.bias"])
    return out
