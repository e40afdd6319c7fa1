"""DetectorMaskRCNN: the detection network of the reference's `Detector`, on the HIP engine.

Same constructor as the reference's src/megapose/models/mask_rcnn.py:23-46 (a torchvision `MaskRCNN` over
`resnet_fpn_backbone("resnet50")` with the megapose anchor sizes and `min_size / max_size = min / max(input_resize)`), the same
state_dict keys (a reference detector checkpoint loads with `strict=True`), and torchvision's eval-mode call contract:
`model(list of [3,H,W] tensors in [0,1]) -> list of dict(boxes [k,4], labels [k], scores [k], masks [k,1,H,W])`.
The module only HOSTS the parameters; the whole graph (transform, ResNet-50 + FPN, RPN, RoIAlign, box / mask heads, NMS, mask
pasting) is ONE native call, `mp_detector_forward` (csrc/detector.hip).  Inference only: calling it in training mode raises.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import engine as eng


class DetectorMaskRCNN(nn.Module):
    def __init__(self, input_resize: Tuple[int, int] = (240, 320), n_classes: int = 2, backbone_str: str = "resnet50-fpn",
                 anchor_sizes: Sequence[Sequence[int]] = ((32,), (64,), (128,), (256,), (512,))):
        super().__init__()
        assert backbone_str == "resnet50-fpn"  # models/mask_rcnn.py:32
        if len(anchor_sizes) != 5 or any(len(s) != 1 for s in anchor_sizes):
            raise NotImplementedError("one anchor size per pyramid level (5 levels), as in every released detector configuration")
        self.n_classes = int(n_classes)
        self.min_size, self.max_size = int(min(input_resize)), int(max(input_resize))
        self.anchor_sizes = tuple(int(s[0]) for s in anchor_sizes)
        self.engine_overrides: Dict[str, object] = {}   # mp_detector_config fields, e.g. {"box_score_thresh": 0.0}
        # False: skip the mask head and the [n, D, H, W] pasted masks (mp_detector_forward with d_masks = NULL); the returned dicts then
        # carry no "masks" entry.  Default True = torchvision's contract (Detector.get_detections reads the masks of every detection).
        self.compute_masks: bool = True
        self._engine: Optional[eng.DetectorNet] = None
        # parameter tree with torchvision's state_dict keys: conv / linear weights and biases are Parameters, the FrozenBatchNorm2d
        # statistics and affine terms are buffers (ops/misc.py), exactly as in a reference checkpoint
        for name, shape in eng.DetectorNet.state_spec(self.n_classes):
            *path, leaf = name.split(".")
            mod: nn.Module = self
            for p in path:
                if p not in mod._modules:
                    mod.add_module(p, nn.Module())
                mod = mod._modules[p]
            is_bn = ".bn" in name or "downsample.1." in name
            t = torch.zeros(shape)
            if is_bn:
                if leaf == "running_var" or leaf == "weight":
                    t.fill_(1.0)
                mod.register_buffer(leaf, t)
            else:
                mod.register_parameter(leaf, nn.Parameter(t, requires_grad=False))

    def load_state_dict(self, state_dict, strict: bool = True):
        self._engine = None
        # (BatchNorm2d checkpoints carry num_batches_tracked counters; FrozenBatchNorm2d ones do not -- accept both)
        sd = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        return super().load_state_dict(sd, strict=strict)

    def invalidate(self) -> None:
        """Drop the packed device weights (they are rebuilt from state_dict() on the next forward)."""
        self._engine = None

    def _fingerprint(self):
        """cheap change detector for the packed device weights: the in-place version counters of every parameter / buffer and
        the engine settings (an in-place weight edit, model.apply(...), or a changed override after the first forward rebuilds them)"""
        return (tuple(t._version for t in list(self.parameters()) + list(self.buffers())), tuple(sorted(self.engine_overrides.items())),
                self.min_size, self.max_size, self.anchor_sizes)

    def _net(self) -> eng.DetectorNet:
        fp = self._fingerprint()
        if self._engine is None or fp != getattr(self, "_engine_fp", None):
            self._engine = eng.DetectorNet(self.state_dict(), self.n_classes, self.min_size, self.max_size,
                                           anchor_sizes=self.anchor_sizes, **self.engine_overrides)
            self._engine_fp = fp
        return self._engine

    @torch.no_grad()
    def forward(self, images: List[torch.Tensor], targets=None) -> List[Dict[str, torch.Tensor]]:
        if self.training or targets is not None:
            raise NotImplementedError("DetectorMaskRCNN on the HIP engine is inference-only (call .eval())")
        if torch.is_tensor(images):
            images = list(images)
        out: List[Optional[Dict[str, torch.Tensor]]] = [None] * len(images)
        groups: Dict[Tuple[int, int], List[int]] = {}
        for i, im in enumerate(images):   # one launch per distinct frame size (the reference's Detector always passes one size)
            assert im.dim() == 3 and im.shape[0] == 3, "images are [3,H,W] tensors in [0,1]"
            groups.setdefault((int(im.shape[1]), int(im.shape[2])), []).append(i)
        net = self._net()
        for (H, W), idx in groups.items():
            batch = torch.stack([images[i] for i in idx]).to(device="cuda", dtype=torch.float32)
            boxes, scores, labels, counts, masks = net.forward(batch, with_masks=self.compute_masks)
            counts_h = counts.cpu().tolist()   # the only host touch: the per-image detection counts size the returned views
            for j, i in enumerate(idx):
                k = counts_h[j]
                out[i] = dict(boxes=boxes[j, :k], labels=labels[j, :k].long(), scores=scores[j, :k])
                if masks is not None:
                    out[i]["masks"] = masks[j, :k, None]
        return out  # type: ignore[return-value]
