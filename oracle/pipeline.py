"""CPU restatement (torch fp32) of the reference's render-and-compare pipeline.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/src/megapose/models/pose_rigid.py (crop_inputs :180-247, compute_crops_multiview :249-303,
render_images_multiview :336-408, normalize_images :410-464, forward :498-604, forward_coarse :634-708) and
/root/reference/src/megapose/inference/pose_estimator.py (forward_coarse_model :324-483, filter_pose_estimates :643-667,
forward_refiner :101-215, forward_scoring_model :217-322, run_inference_pipeline :510-641), with oracle/thirdparty.py
for the un-vendored third-party calls and oracle/raster.c standing in for Panda3D.
Pinned end-to-end against the reference's own orchestration by oracle/make_golden.py -> tests/golden/pipeline_*.npz.
It is also the "port" CPU baseline timed by bench.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import pandas as pd
import torch

from . import backbones as ob
from . import geometry as og
from . import raster as orr
from . import thirdparty as tp


class OraclePosePredictor:
    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], labels: Sequence[str], points_padded: torch.Tensor,
                 renderer: "orr.OracleBatchRenderer", render_size=(240, 320)):
        self.cfg = cfg
        self.sd = {k: v.float() if v.dtype.is_floating_point else v for k, v in state_dict.items()}
        self.label_to_id = {l: i for i, l in enumerate(labels)}
        self.points = points_padded.float()  # [n_obj, Nmax, 3]
        self.renderer = renderer
        self.render_size = render_size
        self.V = cfg.n_rendered_views
        self.ids2000 = og.sample_point_ids(self.points.shape[1], 2000)
        self.ids200 = og.sample_point_ids(self.points.shape[1], 200)
        # "fp16 renders" mode of the engine (BASELINE.json configs[4]; NOT a reference mode -- the reference's renderer output path,
        # panda3d_batch_renderer.py:261-274, is uint8 -> fp32): the CNN input passes through IEEE binary16 (round to nearest even)
        # where the engine stores it as halves -- crop and renders as rendered, the depth channels once more after normalisation.
        self.input_f16 = False

    def _pts(self, labels, ids):
        obj = torch.tensor([self.label_to_id[l] for l in labels], dtype=torch.long)
        return self.points[obj][:, ids]

    def crop_inputs(self, images, im_ids, K, TCO, tCR, labels):
        P = self._pts(labels, self.ids2000)
        uv = og.project_points_robust(P, K, TCO)
        boxes_rend = og.boxes_from_uv(uv)
        boxes_crop = og.crop_boxes_robust(boxes_rend, K, TCO, tCR, P, tuple(images.shape[-2:]))
        rois = torch.cat([im_ids.float()[:, None], boxes_crop], dim=1)
        crops = tp.roi_align(images, rois, self.render_size, sampling_ratio=4)
        if images.shape[1] == 4:  # cropping.py:131-142
            valid = (images[:, 3:4] > 0).float()
            vc = tp.roi_align(valid, rois, self.render_size, sampling_ratio=4)
            crops[:, 3:4] = crops[:, 3:4] * (vc >= 0.99).float()
        K_crop = og.get_K_crop_resize(K.clone(), boxes_crop, self.render_size)
        return crops, K_crop, boxes_rend, boxes_crop

    def crops_multiview(self, im_hw, K, TCV_O, tCV_R, labels):
        b, V = TCV_O.shape[:2]
        labels_mv = [l for l in labels for _ in range(V)]
        P = self._pts(labels_mv, self.ids200)
        Tf, tf = TCV_O.flatten(0, 1), tCV_R.flatten(0, 1)
        Kf = K.unsqueeze(1).repeat(1, V, 1, 1).flatten(0, 1)
        br = og.boxes_from_uv(og.project_points_robust(P, Kf, Tf))
        bc = og.crop_boxes_robust(br, Kf, Tf, tf, P, im_hw)
        return og.get_K_crop_resize(Kf.clone(), bc, self.render_size).view(b, V, 3, 3)

    def render_multiview(self, labels, TCV_O, KV):
        from types import SimpleNamespace

        b, V = TCV_O.shape[:2]
        labels_mv = [l for l in labels for _ in range(V)]
        if self.cfg.render_normals:
            lights = [[SimpleNamespace(light_type="ambient", color=(1.0, 1.0, 1.0, 1.0))] for _ in labels_mv]
        else:
            lights = [[SimpleNamespace(light_type="ambient", color=(0.1, 0.1, 0.1, 1.0))]
                      + [SimpleNamespace(light_type="point", color=(0.4, 0.4, 0.4, 1.0)) for _ in range(6)] for _ in labels_mv]
        d = self.renderer.render(labels=labels_mv, TCO=TCV_O.flatten(0, 1), K=KV.flatten(0, 1), light_datas=lights,
                                 resolution=self.render_size, render_normals=self.cfg.render_normals, render_depth=self.cfg.render_depth)
        cat = [d.rgbs] + ([d.normals] if self.cfg.render_normals else []) + ([d.depths] if self.cfg.render_depth else [])
        r = torch.cat(cat, dim=1)
        return r.view(b, V, r.shape[1], *r.shape[-2:]).flatten(1, 2)

    def normalize_images(self, images_crop, renders, tCR):
        images_crop, renders = images_crop.clone(), renders.clone()
        mode = self.cfg.depth_normalization_type
        if self.cfg.input_depth:
            images_crop[:, 3:4] = og.normalize_depth(images_crop[:, 3:4], tCR, mode)
        if self.cfg.render_depth:
            nper = 3 + (3 if self.cfg.render_normals else 0) + 1
            dd = torch.arange(self.V) * nper + (nper - 1)
            renders[:, dd] = og.normalize_depth(renders[:, dd], tCR, mode)
        return images_crop, renders

    def step(self, images, im_ids, K, labels, TCO_in):
        if not self.cfg.input_depth:
            images = images[:, :3]
        TCO_n = og.normalize_T(TCO_in)
        tCR = TCO_n[:, :3, 3].clone()
        remove = bool(getattr(self.cfg, "remove_TCO_rendering", False))
        TCV_O = og.make_TCO_multiview(TCO_n, tCR, self.cfg.multiview_type, self.V, remove_TCO_rendering=remove)
        tCV_R = TCV_O[..., :3, 3]
        crops, K_crop, boxes_rend, boxes_crop = self.crop_inputs(images, im_ids, K, TCO_n, tCR, labels)
        if self.V > 1:
            KV = self.crops_multiview(tuple(images.shape[-2:]), K, TCV_O, tCV_R, labels)
            if not remove:   # models/pose_rigid.py:551-552
                KV[:, 0] = K_crop
        else:
            KV = K_crop.unsqueeze(1)
        renders = self.render_multiview(labels, TCV_O, KV)
        if self.input_f16:
            crops, renders = crops.half().float(), renders.half().float()
        crops_n, renders_n = self.normalize_images(crops, renders, tCR)
        x = torch.cat((crops_n, renders_n), dim=1)
        if self.input_f16:
            x = x.half().float()
        net = ob.net_forward(self.sd, self.cfg.backbone_str, x)
        return dict(TCO_n=TCO_n, tCR=tCR, TCV_O=TCV_O, KV_crop=KV, K_crop=K_crop, boxes_rend=boxes_rend, boxes_crop=boxes_crop, x=x, net=net)

    @torch.no_grad()
    def forward(self, images, im_ids, K, labels, TCO, n_iterations):
        outs = []
        T = TCO
        for _ in range(n_iterations):
            st = self.step(images, im_ids, K, labels, T)
            st["TCO_output"] = og.update_pose(st["TCO_n"], st["K_crop"], st["net"]["pose"], st["tCR"])
            outs.append(st)
            T = st["TCO_output"]
        return outs

    @torch.no_grad()
    def forward_coarse(self, images, im_ids, K, labels, TCO):
        st = self.step(images, im_ids, K, labels, TCO)
        st["logits"] = st["net"]["renderings_logits"]
        st["scores"] = torch.sigmoid(st["logits"])
        return st


def filter_top_k(df: pd.DataFrame, field: str, top_k: int) -> List[int]:
    g = df.sort_values(field, ascending=False, kind="stable").groupby(["batch_im_id", "label", "instance_id"]).head(top_k)
    return g.index.tolist()


class OraclePoseEstimator:
    def __init__(self, coarse: OraclePosePredictor, refiner: OraclePosePredictor, SO3_grid: torch.Tensor, bsz: int = 32,
                 bsz_refiner: Optional[int] = None):
        self.coarse, self.refiner, self.grid, self.bsz = coarse, refiner, SO3_grid.float(), bsz
        self.bsz_refiner = bsz_refiner or bsz  # (the reference batches coarse by bsz_images and refiner by bsz_objects)

    @torch.no_grad()
    def run(self, images: torch.Tensor, K_im: torch.Tensor, det_infos: pd.DataFrame, bboxes: torch.Tensor, n_refiner_iterations=5,
            n_pose_hypotheses=1, coarse_estimates=None, max_coarse_rows: Optional[int] = None):
        """Returns dict with every intermediate the parity tests compare."""
        res: Dict[str, object] = {}
        if coarse_estimates is None:
            B, M = len(det_infos), self.grid.shape[0]
            df = det_infos.reset_index(drop=True)
            dfh = df.loc[df.index.repeat(M)].copy()
            dfh["hypothesis_id"] = np.tile(np.arange(M), B)
            dfh["bbox_id"] = np.repeat(df.index.values, M)
            dfh = dfh.reset_index(drop=True)
            labels = dfh["label"].tolist()
            im = torch.as_tensor(dfh["batch_im_id"].values.astype(np.int64))
            Kr = K_im[im]
            obj = torch.tensor([self.coarse.label_to_id[l] for l in labels])
            TCO = torch.cat([og.TCO_init_from_boxes_autodepth_with_R(
                bboxes[dfh["bbox_id"].values[s : s + self.bsz]].float(), self.coarse.points[obj[s : s + self.bsz]], Kr[s : s + self.bsz],
                self.grid[dfh["hypothesis_id"].values[s : s + self.bsz]]) for s in range(0, len(dfh), self.bsz)])
            n_rows = len(dfh) if max_coarse_rows is None else min(len(dfh), max_coarse_rows)
            logits = torch.cat([self.coarse.forward_coarse(images, im[s : s + self.bsz], Kr[s : s + self.bsz], labels[s : s + self.bsz],
                                                           TCO[s : s + self.bsz])["logits"] for s in range(0, n_rows, self.bsz)])
            if n_rows < len(dfh):
                logits = torch.cat([logits, torch.full((len(dfh) - n_rows, 1), -1e9)])
            dfh["coarse_logit"] = logits.flatten().numpy()
            dfh["coarse_score"] = torch.sigmoid(logits).flatten().numpy()
            res["coarse_TCO"], res["coarse_logits"], res["coarse_infos"] = TCO, logits.flatten(), dfh
            keep = filter_top_k(dfh, "coarse_logit", n_pose_hypotheses)
            dff, T0 = dfh.iloc[keep].reset_index(drop=True), TCO[keep]
        else:
            dff, T0 = coarse_estimates
            dff = dff.reset_index(drop=True)
        res["filtered_infos"], res["filtered_TCO"] = dff, T0
        labels = dff["label"].tolist()
        im = torch.as_tensor(dff["batch_im_id"].values.astype(np.int64))
        Kr = K_im[im]
        per_iter = [[] for _ in range(n_refiner_iterations)]
        rb = self.bsz_refiner
        for s in range(0, len(dff), rb):
            outs = self.refiner.forward(images, im[s : s + rb], Kr[s : s + rb], labels[s : s + rb], T0[s : s + rb], n_refiner_iterations)
            for n, o in enumerate(outs):
                per_iter[n].append(o)
        res["refiner_poses"] = [torch.cat([o["TCO_output"] for o in it]) for it in per_iter]
        res["refiner_K_crop"] = [torch.cat([o["K_crop"] for o in it]) for it in per_iter]
        res["refiner_pose_out"] = [torch.cat([o["net"]["pose"] for o in it]) for it in per_iter]
        T_ref = res["refiner_poses"][-1]
        sl = torch.cat([self.coarse.forward_coarse(images, im[s : s + self.bsz], Kr[s : s + self.bsz], labels[s : s + self.bsz],
                                                   T_ref[s : s + self.bsz])["logits"] for s in range(0, len(dff), self.bsz)])
        dfs = dff.copy()
        dfs["pose_logit"] = sl.flatten().numpy()
        dfs["pose_score"] = torch.sigmoid(sl).flatten().numpy()
        res["scoring_logits"], res["scored_infos"] = sl.flatten(), dfs
        best = filter_top_k(dfs, "pose_logit", 1)
        res["final_infos"], res["final_TCO"] = dfs.iloc[best].reset_index(drop=True), T_ref[best]
        return res
