"""Detector: the 2D-detection front-end of the pose pipeline.

Same class, constructor and `get_detections` contract as the reference's src/megapose/inference/detector.py:33-139: wraps a
detection model that maps a list of [3,H,W] images in [0,1] to a list of dicts `boxes [n,4]` (x1,y1,x2,y2), `labels [n]`
(category ids), `scores [n]`, `masks [n,1,H,W]` (torchvision's Mask R-CNN output format, which the reference's
`DetectorMaskRCNN`, src/megapose/models/mask_rcnn.py:23-46, inherits) and turns the result into the `DetectionsType`
PandasTensorCollection the pose estimator consumes (`infos`: batch_im_id / label / score / instance_id, `bboxes`, optional
`masks`).  The model is either the engine's own `megapose6d_amd.mask_rcnn.DetectorMaskRCNN` (HIP) or any module with the same
output format and a `.config.label_to_category_id` mapping.

Engine-side differences, by design: the output tensors stay on the device the model produced them on (the reference calls
`.cuda()` on them, detector.py:115-120, which is the same thing on its only supported set-up), and the per-detection Python
loop (`detector.py:99-112`, one `.item()` D2H sync per detection) is replaced by one host transfer of the scores / labels per image.
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import pandas as pd
import torch

from .pose_estimator import add_instance_id, filter_detections
from .tcoll import PandasTensorCollection
from .types import DetectionsType, ObservationTensor


class Detector(torch.nn.Module):
    def __init__(self, model: torch.nn.Module) -> None:
        super().__init__()
        self.model = model
        self.model.eval()
        self.config = model.config
        self.category_id_to_label = {v: k for k, v in self.config.label_to_category_id.items()}

    def image_tensor_from_numpy(self, rgb: np.ndarray) -> torch.Tensor:
        """uint8 [H,W,3] -> float [3,H,W] in [0,1] (detector.py:41-61)"""
        assert rgb.dtype == np.uint8
        rgb_tensor = torch.as_tensor(rgb).float() / 255
        if rgb_tensor.shape[-1] == 3:
            rgb_tensor = rgb_tensor.permute(2, 0, 1)
        return rgb_tensor

    @torch.no_grad()
    def get_detections(self, observation: ObservationTensor, detection_th: Optional[float] = None, output_masks: bool = False,
                       mask_th: float = 0.8, one_instance_per_class: bool = False) -> DetectionsType:
        """detector.py:63-136.  detection_th: keep detections scoring above it; mask_th: threshold of the soft masks;
        one_instance_per_class: keep the best detection of every (image, label)."""
        images = observation.images[:, [0, 1, 2]]  # [B,3,H,W]
        engine_model = hasattr(self.model, "compute_masks")
        if engine_model:   # the engine's Mask R-CNN can skip its mask head + the pasted [n, D, H, W] masks when nobody reads them
            saved, self.model.compute_masks = self.model.compute_masks, bool(output_masks and self.model.compute_masks)
        try:
            outputs_ = self.model([image_n for image_n in images])
        finally:
            if engine_model:
                self.model.compute_masks = saved
        device = images.device
        infos, bboxes, masks = [], [], []
        for n, out_n in enumerate(outputs_):
            n_det = len(out_n["boxes"])
            if n_det == 0:
                continue
            labels_n = [self.category_id_to_label[int(c)] for c in torch.as_tensor(out_n["labels"]).cpu().tolist()]
            scores_n = torch.as_tensor(out_n["scores"]).cpu().tolist()
            infos += [dict(batch_im_id=n, label=l, score=float(s)) for l, s in zip(labels_n, scores_n)]
            bboxes.append(torch.as_tensor(out_n["boxes"]).to(device))
            if "masks" in out_n and output_masks:   # (the [n, D, H, W] soft masks are only thresholded when the caller wants them)
                masks.append(torch.as_tensor(out_n["masks"])[:, 0].to(device) > mask_th)
            elif output_masks:   # (engine model with compute_masks = False)
                raise ValueError("output_masks=True, but the detection model returned no masks (DetectorMaskRCNN.compute_masks is off)")
        if len(bboxes) > 0:
            bboxes_t = torch.cat(bboxes).float()
            masks_t = torch.cat(masks) if masks else None
            infos_df = pd.DataFrame(infos)
        else:  # (the reference builds an empty frame from a dict of empty lists, detector.py:117-120)
            infos_df = pd.DataFrame(dict(score=[], label=[], batch_im_id=[]))
            bboxes_t = torch.empty(0, 4, device=device).float()
            masks_t = torch.empty(0, images.shape[2], images.shape[3], dtype=torch.bool, device=device)
        outputs = PandasTensorCollection(infos=infos_df, bboxes=bboxes_t)
        if output_masks:
            outputs.register_tensor("masks", masks_t)
        if detection_th is not None:
            keep = np.where(outputs.infos["score"] > detection_th)[0]
            outputs = outputs[keep]
        if one_instance_per_class:
            outputs = filter_detections(outputs, one_instance_per_class=True)
        # instance_id tells apart several detections of one object class in one image (inference/utils.py:151-171)
        outputs = add_instance_id(outputs)
        return outputs

    def __call__(self, *args: Any, **kwargs: Any) -> DetectionsType:
        return self.get_detections(*args, **kwargs)
