"""PosePredictor: one render-and-compare step (crop -> multiview cameras -> render -> CNN -> pose update).

Public surface = the reference's src/megapose/models/pose_rigid.py:81-708 (`forward`, `forward_coarse`,
`forward_coarse_tensor`, `crop_inputs`, `compute_crops_multiview`, `update_pose`, `net_forward`,
`render_images_multiview`, `normalize_images`, attributes cfg-driven).  The module only HOSTS the parameter graph
(state_dict keys identical to the reference checkpoints); every computation is a call into
libmp_engine.so: the crop, the rasteriser and the depth normalisation write straight into one padded-NHWC CNN input
tensor, the backbone executor consumes it, and the pose math runs as fused device kernels -- nothing leaves the GPU and
nothing synchronises.
"""
from __future__ import annotations

import os

import time
from collections import defaultdict
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch import nn

from . import engine as eng
from .mesh_db import BatchedMeshes
from .renderer import Panda3dBatchRenderer
from .types import Panda3dLightData, PosePredictorOutput, Resolution, make_scene_lights


# ----------------------------------------------------------------------------------------------------------------
# Parameter containers: module trees whose state_dict keys equal the reference backbones'.  They are never called.
# ----------------------------------------------------------------------------------------------------------------
class _PlainBlock(nn.Module):  # torchvision BasicBlock parameter layout
    def __init__(self, inplanes: int, planes: int, downsample: bool):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, bias=False), nn.BatchNorm2d(planes))


class _PreActBlock(nn.Module):  # wide_resnet BasicBlockV2 parameter layout
    def __init__(self, inplanes: int, planes: int, downsample: bool):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.conv1 = nn.Conv2d(inplanes, planes, 3, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, bias=False)
        if downsample:
            self.downsample = nn.Conv2d(inplanes, planes, 1, bias=False)


class HipBackbone(nn.Module):
    """Hosts the weights of vanilla_resnet34 / WideResNet34 / WideResNet18 / WideResNet34 x width (pose_models_cfg.py:106-118)."""

    def __init__(self, backbone_str: str, n_inputs: int):
        super().__init__()
        width = 1
        base = backbone_str
        if backbone_str.startswith("resnet34_width="):   # WideResNet34(width=N): stage widths 64N .. 512N (models/wide_resnet.py:62, :118-121)
            width, base = int(backbone_str.split("resnet34_width=")[1]), "resnet34"
        if base not in eng.BACKBONE_KINDS:
            raise ValueError("Unknown backbone", backbone_str)
        self.backbone_str = backbone_str
        self.n_inputs = n_inputs
        self.n_features = 512 * width
        wide = base != "vanilla_resnet34"
        widths = [64 * width, 128 * width, 256 * width, 512 * width]
        self.conv1 = nn.Conv2d(n_inputs, widths[0], 5 if wide else 7, bias=False)
        self.bn1 = nn.BatchNorm2d(widths[0])
        counts = [2, 2, 2, 2] if base == "resnet18" else [3, 4, 6, 3]
        inplanes = widths[0]
        for s, (planes, n) in enumerate(zip(widths, counts)):
            blocks = []
            for i in range(n):
                down = i == 0 and (s > 0 or inplanes != planes)
                blocks.append((_PreActBlock if wide else _PlainBlock)(inplanes, planes, down))
                inplanes = planes
            setattr(self, f"layer{s + 1}", nn.Sequential(*blocks))
        if not wide:
            self.fc = nn.Linear(512, 512)

    def forward(self, x):  # pragma: no cover - the graph is executed by the HIP engine through PosePredictor
        raise RuntimeError("HipBackbone only hosts parameters; use PosePredictor.net_forward (HIP engine)")


# ----------------------------------------------------------------------------------------------------------------
class _RefinerGraph:
    """One captured refiner call of `rows` rows: n_iterations x (pose_prepare, render + crop, backbone, pose update) on static buffers."""

    def __init__(self, model: "PosePredictor", rows: int, n_iterations: int, slot: int, packed: "eng.PackedObservation", device):
        V = model.n_rendered_views
        self.TCO = torch.zeros(rows, 4, 4, dtype=torch.float32, device=device)
        self.K = torch.zeros(rows, 3, 3, dtype=torch.float32, device=device)
        self.im_ids = torch.zeros(rows, dtype=torch.int32, device=device)
        self.pts_ids = torch.zeros(rows, dtype=torch.int32, device=device)
        self.ren_ids = torch.zeros(rows, dtype=torch.int32, device=device)
        self.obs = eng.PackedObservation.__new__(eng.PackedObservation)
        self.obs.n_im, self.obs.C, self.obs.H, self.obs.W = packed.n_im, packed.C, packed.H, packed.W
        self.obs.data = torch.empty_like(packed.data)
        width = sum(wd for _, wd in model._graph_widths())
        self.pack = torch.zeros(n_iterations, rows, width, dtype=torch.float32, device=device)
        self.graph = torch.cuda.CUDAGraph()
        self._fill = None
        # A captured call holds raw device addresses of the CNN-input buffer, the backbone workspace and the rasteriser workspace.  The
        # eager path's per-slot buffers are grow-only (a later, larger eager call on the same slot reallocates them and the old storage
        # goes back to the allocator), so every graph works on PRIVATE buffers under its own slot key: sized once for `rows`, never
        # touched by any other call, released by PosePredictor._drop_graph.
        self.slot_key = ("graph", rows, n_iterations, slot, id(self))
        self._model, self._n_iterations, self._slot = model, n_iterations, self.slot_key

    def _body(self):
        m = self._model
        TCO_input = self.TCO
        dummy = [""] * self.TCO.shape[0]
        for n in range(self._n_iterations):
            st = m._step(None, self.im_ids, self.K, dummy, TCO_input, want_sigmoid=False, slot=self._slot, ids=(self.pts_ids, self.ren_ids),
                         packed=self.obs)
            TCO_output = eng.pose_update(st["TCO_n"], st["K_crop"], st["out"], st["tCR"], 9)
            torch.cat([st["TCO_n"].flatten(1), TCO_output.flatten(1), st["K_crop"].flatten(1), st["KV_crop"].flatten(1), st["boxes_rend"],
                       st["boxes_crop"], st["out"], st["tCR"], st["TCV_O"].flatten(1)], dim=1, out=self.pack[n])
            TCO_input = TCO_output

    def run(self, TCO, K, im_ids, ids, packed) -> torch.Tensor:
        self.TCO.copy_(TCO)
        self.K.copy_(K)
        self.im_ids.copy_(im_ids)
        self.pts_ids.copy_(ids[0])
        self.ren_ids.copy_(ids[1])
        self.obs.data.copy_(packed.data)
        if self._fill is None:
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self._body()
            self._fill = True
        self.graph.replay()
        return self.pack.clone()


# ----------------------------------------------------------------------------------------------------------------
class PosePredictor(nn.Module):
    def __init__(
        self,
        backbone: HipBackbone,
        renderer: Panda3dBatchRenderer,
        mesh_db: BatchedMeshes,
        render_size: Resolution = (240, 320),
        multiview_type: str = "front_3views",
        views_inplane_rotations: bool = False,
        remove_TCO_rendering: bool = False,
        predict_pose_update: bool = True,
        predict_rendered_views_logits: bool = False,
        render_normals: bool = True,
        n_rendered_views: int = 1,
        input_depth: bool = False,
        render_depth: bool = False,
        depth_normalization_type: Optional[str] = None,
    ):
        super().__init__()
        # legacy names (training/pose_models_cfg.py:49-54)
        multiview_type = {"front_3views": "TCO+front_3views", "front_1view": "TCO+front_1view"}.get(multiview_type, multiview_type)
        # The view list is what make_TCO_multiview returns (lib3d/multiview.py:165-246): n_views == 1 -> [TCO] whatever the type;
        # else [TCO unless remove_TCO_rendering] + the type's camera offsets.  `views_inplane_rotations` is stored but -- exactly as
        # in the reference -- never reaches make_TCO_multiview from forward() (models/pose_rigid.py:531-537 passes only
        # remove_TCO_rendering; the 4x in-plane copies are used by the training loss only, megapose_forward_loss.py:115-116).
        if n_rendered_views == 1:
            # (forward() of a one-view model with remove_TCO_rendering -- the released COARSE recipe, run_megapose_training.py:131-142 --
            #  renders with the 200-point multiview crop camera, pose_rigid.py:550-552; forward_coarse, the hot path, never does)
            self._mv_mode = eng.MV_REMOVE_TCO if remove_TCO_rendering else 0
        else:
            if multiview_type not in eng.MV_MODES or eng.MV_MODES[multiview_type] == 0:
                raise ValueError(multiview_type)   # lib3d/multiview.py:233-234 ("TCO+front_5views" is not implemented there either)
            self._mv_mode = eng.MV_MODES[multiview_type] | (eng.MV_REMOVE_TCO if remove_TCO_rendering else 0)
            n_list = eng.multiview_n_views(self._mv_mode)
            if n_list != n_rendered_views:
                raise ValueError(f"multiview_type={multiview_type} (remove_TCO_rendering={remove_TCO_rendering}) renders {n_list} views, "
                                 f"the model is configured for {n_rendered_views}")
        self.backbone = backbone
        self.renderer = renderer
        self.mesh_db = mesh_db
        self.render_size = render_size
        self.n_rendered_views = n_rendered_views
        self.input_depth = input_depth
        self.multiview_type = multiview_type
        self.views_inplane_rotations = views_inplane_rotations
        self.render_normals = render_normals
        self.render_depth = render_depth
        self.depth_normalization_type = depth_normalization_type
        self.predict_rendered_views_logits = predict_rendered_views_logits
        self.remove_TCO_rendering = remove_TCO_rendering
        self.predict_pose_update = predict_pose_update

        n_features = backbone.n_features
        self.heads: Dict[str, nn.Linear] = dict()
        if self.predict_pose_update:
            self._pose_dim = 9
            self.pose_fc = nn.Linear(n_features, self._pose_dim, bias=True)
            self.heads["pose"] = self.pose_fc
        if self.predict_rendered_views_logits:
            self.views_logits_head = nn.Linear(n_features, self.n_rendered_views, bias=True)
            self.heads["renderings_logits"] = self.views_logits_head
        if len(self.heads) != 1:
            raise NotImplementedError("exactly one head (pose update or view logits) is supported, as in the released models")

        self._n_input_channels = 3 + (1 if input_depth else 0)
        self._n_single_render_channels = 3 + (3 if render_normals else 0) + (1 if render_depth else 0)
        n_in = self._n_input_channels + self._n_single_render_channels * n_rendered_views
        if n_in != backbone.n_inputs:
            raise ValueError(f"backbone has {backbone.n_inputs} input channels, configuration needs {n_in}")
        self.debug = False
        self.timing_dict: Dict[str, float] = defaultdict(float)
        # channel index helpers of the reference (models/pose_rigid.py:132-158) + its debug container (:69-78, filled when debug=True)
        self._input_rgb_dims = [0, 1, 2]
        self._input_depth_dims = [3] if input_depth else []
        self._render_rgb_dims = [0, 1, 2]
        self._render_normal_dims = [3, 4, 5] if render_normals else []
        self._render_depth_dims = [3 + len(self._render_normal_dims)] if render_depth else []
        self.debug_data = SimpleNamespace(output=None, images=None, origin_uv=None, ref_point_uv=None, origin_uv_crop=None,
                                          pose_predictor_outputs=None)
        self._engine_bb: Optional[eng.Backbone] = None
        # torch.float16 = the "fp16 renders" mode (BASELINE.json configs[4]): the rasteriser launch stores the whole CNN input
        # (renders + observation crop) as binary16 and the stem convolution widens it on its way into LDS; half the bytes of the
        # largest tensor of a step.  Narrower than the reference (fp32 inputs, models/pose_rigid.py:567): off by default.
        self.render_dtype: torch.dtype = torch.float32
        # Stem records (default ON where they apply: RGB models, <= 32 input channels): the rasteriser launch stores every pixel of the
        # CNN input as a bf16 RECORD -- the 8-bit integer k of each render channel (the reference's renders ARE k / 255,
        # panda3d_batch_renderer.py:261-274) and three exact bf16 pieces of each fp32 crop channel -- and the stem convolution runs on
        # the bf16 MFMA with the weights split into three exact pieces (csrc/conv_stem.hip): every product is exact, the sum is fp32;
        # against the fp32-MFMA stem the order of the fp32 additions differs, and the 1/255 of the integer channels is folded into their
        # weights (one rounding per weight instead of one per pixel value).  MP_STEM_RECORDS=0 / stem_records=False: fp32 tensor.
        self.stem_records: bool = os.environ.get("MP_STEM_RECORDS", "1") != "0"
        # Background tiles (round 5): the rasteriser's compacted launch knows which 8x8-pixel tiles of a row's crop no view reaches; the
        # stem's workgroups over such tiles walk only the record chunks of the observation crop (csrc/conv_stem.hip, SP instances).  RGB
        # models on records, one raster launch per step.  MP_STEM_SPARSE=0 / stem_sparse=False: every workgroup takes the dense walk.
        self.stem_sparse: bool = os.environ.get("MP_STEM_SPARSE", "1") != "0"
        self._x: Dict[int, torch.Tensor] = {}      # CNN input buffer per slot (= concurrent HIP stream)
        self._x_rows: Dict[int, int] = {}
        self._x_cp: Dict[int, int] = {}
        self._label_cache: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}
        # hipGraph capture of small refiner calls (n_iterations x ~45 launches for a handful of rows): calls of at most `graph_rows`
        # rows are captured once per (rows, frame shape, n_iterations, slot) and replayed.  OFF by default (0): measured on the MI355X
        # (tests/test_gpu_refiner_graph.py, 1 row x 5 iterations) replay 5.98 ms vs eager 5.96 ms -- the launches are asynchronous and
        # the small-batch call is bound by the device time of ~40 dependent small-grid kernels per iteration, not by launch latency
        self.graph_rows: int = int(os.environ.get("MP_REFINER_GRAPH_ROWS", "0"))
        self._graphs: Dict[tuple, "_RefinerGraph"] = {}
        self._graph_seen: Dict[tuple, int] = {}

    # -- engine plumbing -----------------------------------------------------------------------------------------
    def load_state_dict(self, *args, **kwargs):
        self.invalidate_graphs()   # captured refiner calls hold the old engine's device pointers
        self._engine_bb = None
        return super().load_state_dict(*args, **kwargs)

    def invalidate_graphs(self) -> None:
        """Drop the captured refiner calls (`graph_rows` > 0): needed whenever the engine backbone, the renderer's mesh database or a
        mode such as `stem_records` / `render_dtype` changes after a capture; `load_state_dict` does it itself."""
        for key in list(self._graphs):
            self._drop_graph(key)
        self._graph_seen.clear()

    def _drop_graph(self, key) -> None:
        """forget one captured call and release the private buffers it ran on (CNN input, backbone / rasteriser workspaces)"""
        g = self._graphs.pop(key, None)
        if g is None:
            return
        sk = g.slot_key
        self._x.pop(sk, None); self._x_rows.pop(sk, None); self._x_cp.pop(sk, None)
        if self._engine_bb is not None:
            self._engine_bb._ws.pop(sk, None)
        db = getattr(self.renderer, "_mesh_db", None)
        if db is not None:
            db._ws.pop(sk, None)

    def _backbone_engine(self) -> eng.Backbone:
        if self._engine_bb is None:
            head, n_out = ("pose", 9) if self.predict_pose_update else ("logits", self.n_rendered_views)
            self._engine_bb = eng.Backbone(self.backbone.backbone_str, self.backbone.n_inputs, head, n_out, self.state_dict())
        return self._engine_bb

    def _f32_mask(self) -> int:
        """bit c set = logical input channel c is fp32-kind in a stem record (three exact bf16 pieces): the observation crop's channels
        and, for depth models, the rendered depth of every view (pose_rigid.py:395-408 channel layout); rgb / normals are 8-bit integers"""
        nin, nper = self._n_input_channels, self._n_single_render_channels
        mask = (1 << nin) - 1
        if self.render_depth:
            d0 = nin + (6 if self.render_normals else 3)
            for v in range(self.n_rendered_views):
                mask |= 1 << (d0 + nper * v)
        return mask

    def _record_len(self) -> int:
        """bf16 elements per pixel if this model's CNN input is staged as stem records (see `stem_records`), else 0.  Depth models too
        since round 5 (the RGBD refiner's 32 channels = a 48-element record; the launch normalises the depth channels before the split)."""
        if (not self.stem_records or self.render_dtype != torch.float32
                or self.backbone.n_inputs > 32):   # (one rasteriser launch must write the whole record: <= 32 channels)
            return 0
        return self._backbone_engine().xrec_elements(f32_mask=self._f32_mask())

    def _x_layout(self) -> Tuple[torch.dtype, int]:
        """(element type, elements per pixel) of the CNN input tensor"""
        R = self._record_len()
        return (torch.bfloat16, R) if R else (self.render_dtype, self._backbone_engine().c_in_p)

    def _x_buffer(self, rows: int, device, slot: int = 0) -> torch.Tensor:
        bb = self._backbone_engine()
        h, w = self.render_size
        if self.render_dtype not in (torch.float32, torch.float16):
            raise ValueError(f"render_dtype must be torch.float32 or torch.float16, got {self.render_dtype}")
        dtype, Cp = self._x_layout()
        x = self._x.get(slot)
        if x is None or self._x_rows[slot] < rows or x.device != device or x.dtype != dtype or self._x_cp.get(slot) != Cp:
            self._x.pop(slot, None)
            self._x[slot] = x = eng.padded_nhwc(rows, h, w, Cp, bb.in_border, device, dtype=dtype)
            self._x_rows[slot] = rows
            self._x_cp[slot] = Cp
        return x

    def _x_geometry(self):
        bb = self._backbone_engine()
        h, w = self.render_size
        Cp = self._x_layout()[1]
        Wp, Hp, B = w + 2 * bb.in_border, h + 2 * bb.in_border, bb.in_border
        return Hp * Wp * Cp, Wp * Cp, Cp, (B * Wp + B) * Cp  # stride_row, stride_y, stride_x, interior offset (in elements)

    def _ids(self, labels: Sequence[str], device) -> Tuple[torch.Tensor, torch.Tensor]:
        key = (tuple(labels), str(device))
        hit = self._label_cache.get(key)
        if hit is None:
            pts_ids = torch.tensor(self.mesh_db.ids(labels), dtype=torch.int32, device=device)
            ren_ids = self.renderer.label_ids(labels, device)
            if len(self._label_cache) > 64:
                self._label_cache.clear()
            self._label_cache[key] = hit = (pts_ids, ren_ids)
        return hit

    def _lights(self) -> List[Panda3dLightData]:
        if self.render_normals:
            return [Panda3dLightData(light_type="ambient", color=(1.0, 1.0, 1.0, 1.0))]  # pose_rigid.py:374-376
        return make_scene_lights()

    def _nchw_view(self, rows: int, c0: int, c1: int, slot: int = 0) -> torch.Tensor:
        bb = self._backbone_engine()
        h, w = self.render_size
        x = self._x[slot]
        Cp = self._x_cp[slot]
        v = eng.padded_view(x, self._x_rows[slot], h, w, Cp, bb.in_border)[:rows]
        if x.dtype == torch.bfloat16:   # stem records -> the fp32 channels they stand for: x1 + x2 + x3 (exact), k / 255
            mask, n_in = self._f32_mask(), self.backbone.n_inputs
            f_ch = [c for c in range(n_in) if (mask >> c) & 1]
            u_ch = [c for c in range(n_in) if not (mask >> c) & 1]
            nf = len(f_ch)
            pieces = v[..., : 3 * nf].float().reshape(rows, h, w, nf, 3)
            f32 = (pieces[..., 0] + pieces[..., 1]) + pieces[..., 2]
            # (a tensor divisor: torch's GPU division by a Python scalar multiplies by the reciprocal, which is not k / 255 rounded once)
            u8 = v[..., 3 * nf : 3 * nf + len(u_ch)].float() / torch.full((), 255.0, device=x.device)
            full = torch.empty(rows, h, w, n_in, dtype=torch.float32, device=x.device)
            full[..., f_ch] = f32
            full[..., u_ch] = u8
            return full[..., c0:c1].permute(0, 3, 1, 2)
        v = v[:, :, :, c0:c1].permute(0, 3, 1, 2)
        return v if v.dtype == torch.float32 else v.float()   # (fp16 renders mode: callers always see fp32 crops / renders)

    def _packed(self, images: torch.Tensor) -> "eng.PackedObservation":
        """[n_im,C,H,W] frames -> NHWC4 copy for the fused crop, cached while the same (unmodified) tensor keeps coming in"""
        key = (images.untyped_storage().data_ptr(), images.storage_offset(), tuple(images.shape), tuple(images.stride()), images._version,
               images.device)  # a fresh `images[:, :3]` view of the same frame maps to the same key
        if getattr(self, "_packed_key", None) != key:
            self._packed_obs = eng.PackedObservation(images)
            self._packed_key = key
            self._packed_src = images  # keeps the storage alive: a freed frame's address could otherwise be reused by a new frame
            self._packed_event = torch.cuda.Event()
            self._packed_event.record()
        else:
            torch.cuda.current_stream().wait_event(self._packed_event)  # another slot's stream may have done the packing
        return self._packed_obs

    # -- the fused step ------------------------------------------------------------------------------------------
    def _step(self, images: torch.Tensor, im_ids: torch.Tensor, K: torch.Tensor, labels: Sequence[str], TCO_in: torch.Tensor,
              want_sigmoid: bool, slot: int = 0, events: bool = False, mv_mode: Optional[int] = None, ids=None, packed=None):
        """images [n_im,C,H,W] (C already trimmed to the model's input channels), im_ids [b] row -> image.
        Returns dict of device tensors; the CNN input stays in self._x.  `ids` = (pts_ids, ren_ids) / `packed` = the NHWC4
        observation override what `labels` / `images` would give (the graph-captured refiner call feeds static buffers).
        `events=True` records three HIP events on the launch stream (before the render+crop launch, after it, after the
        backbone) so that the caller can report DEVICE render / model times (`step_times`); without them `render_time`
        is the host time spent enqueueing the render (the launch is asynchronous, unlike the reference's Panda3D call)."""
        device = TCO_in.device
        b = TCO_in.shape[0]
        V = self.n_rendered_views
        h, w = self.render_size
        H, W = images.shape[-2:] if packed is None else (packed.H, packed.W)
        pts_ids, ren_ids = self._ids(labels, device) if ids is None else ids
        points = self.mesh_db.sampled_points(2000)
        TCO_n, tCR, TCV_O, KV_crop, boxes_rend, boxes_crop, K_main = eng.pose_prepare(
            TCO_in, K, pts_ids, points, 2000, 200, V, self._mv_mode if mv_mode is None else mv_mode, (H, W), (h, w), 1.4, with_K_main=True)
        x = self._x_buffer(b, device, slot)
        s_row, s_y, s_x, off = self._x_geometry()
        # the observation crop (channels 0..nin-1) is written by the rasteriser launch below (one launch fills the whole CNN input)
        nin, nper = self._n_input_channels, self._n_single_render_channels
        t0 = time.time()
        ev = None
        if events:
            ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
            ev[0].record()
        # one launch writes at most 32 channels per pixel: the released recipes (<= 4 views) need one; longer view lists
        # (sphere_26views) go in groups of views, the observation crop rides with the first
        vg = max(1, (32 - nin) // nper)
        records = x.dtype == torch.bfloat16
        mode = eng.DEPTH_NORM_MODES[self.depth_normalization_type]
        for v0 in range(0, V, vg):
            v1 = min(V, v0 + vg)
            nv = v1 - v0
            whole = nv == V
            Tg = TCV_O.view(b * V, 4, 4) if whole else TCV_O[:, v0:v1].reshape(b * nv, 4, 4)
            Kg = KV_crop.view(b * V, 3, 3) if whole else KV_crop[:, v0:v1].reshape(b * nv, 3, 3)
            view_ids = ren_ids.repeat_interleave(nv) if nv > 1 else ren_ids
            c0 = nin + nper * v0
            # stem records of a model with depth channels: the launch normalises them before the exact split (mp_raster_render_xrec)
            xrec = (self._f32_mask(), tCR, mode) if (records and (self.input_depth or self.render_depth)) else None
            self.renderer.render_into(view_ids, Tg, Kg, self._lights(), (h, w), x, s_row, s_y, s_x, c0,
                                      c0 + 3 if self.render_normals else -1,
                                      c0 + (6 if self.render_normals else 3) if self.render_depth else -1, off,
                                      views_per_item=nv, stride_view=nper, slot=slot,
                                      crop=((self._packed(images) if packed is None else packed), im_ids, boxes_crop, 0) if v0 == 0 else None,
                                      xrec=xrec)
        render_time = time.time() - t0
        if ev is not None:
            ev[1].record()
        bb = self._backbone_engine()
        depth_ch = []
        if self.input_depth:
            depth_ch.append(3)
        if self.render_depth:
            d0 = nin + (6 if self.render_normals else 3)
            depth_ch += [d0 + nper * v for v in range(V)]
        if depth_ch and mode and not records:   # (records: normalised inside the rasteriser launch)
            eng.normalize_depth(x, b, h, w, bb.in_border, bb.c_in_p, depth_ch, tCR, mode)
        n_out = bb.n_out
        out = torch.empty(b, n_out, dtype=torch.float32, device=device)
        sig = torch.empty(b, n_out, dtype=torch.float32, device=device) if want_sigmoid else None
        tile_flags = 0
        if records and self.stem_sparse and vg >= V and not (self.input_depth or self.render_depth):
            # the job flags of THIS step's (single) raster launch: same stream, consumed before the next launch on this slot's workspace;
            # 0 (dense walk) unless that launch ran in the compacted form with this shape -- the library keeps the record
            tile_flags = eng.raster_job_flags(self.renderer._ensure_db(), b * V, h, w, device, slot)
        bb.forward(x, b, h, w, out, sig, slot=slot, f32_mask=self._f32_mask(), tile_flags=tile_flags)
        if ev is not None:
            ev[2].record()
        return dict(TCO_n=TCO_n, tCR=tCR, TCV_O=TCV_O, KV_crop=KV_crop, K_crop=K_main, boxes_rend=boxes_rend, boxes_crop=boxes_crop, out=out,
                    sigmoid=sig, render_time=render_time, events=ev)

    @staticmethod
    def step_times(events) -> Tuple[float, float]:
        """(render_s, model_s) of one step from the events `_step(events=True)` recorded; the caller must have synchronised."""
        if events is None:
            return 0.0, 0.0
        return events[0].elapsed_time(events[1]) / 1000.0, events[1].elapsed_time(events[2]) / 1000.0

    def _prep_images(self, images: torch.Tensor) -> torch.Tensor:
        if not self.input_depth:
            images = images[:, :3]  # pose_rigid.py:511-513, :669-671
        return images

    # -- reference API -----------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, images: torch.Tensor, K: torch.Tensor, labels: List[str], TCO: torch.Tensor, n_iterations: int = 1,
                random_ambient_light: bool = False, im_ids: Optional[torch.Tensor] = None,
                materialize: bool = True, slot: int = 0, cuda_timer: bool = False) -> Dict[str, PosePredictorOutput]:
        """Same contract as the reference forward (pose_rigid.py:498-604).  Engine extensions: `im_ids` lets several rows
        share one observation frame (images is then [n_im,C,H,W] and K stays per-row); `materialize=False` skips the
        clones of the crops/renders (they stay valid only until the next step); `cuda_timer=True` attaches the step's HIP
        events to `timing_dict["events"]` (resolve with `PosePredictor.step_times` after a synchronise)."""
        if random_ambient_light:
            raise NotImplementedError("random_ambient_light is a training-time augmentation")
        images = self._prep_images(images)
        bsz = TCO.shape[0]
        assert TCO.shape == (bsz, 4, 4) and K.shape == (bsz, 3, 3) and len(labels) == bsz
        device = TCO.device
        if im_ids is None:
            assert images.shape[0] == bsz
            im_ids = torch.arange(bsz, dtype=torch.int32, device=device)
        if (self.predict_pose_update and 0 < bsz <= self.graph_rows and not materialize and not cuda_timer and not eng.profiling()
                and self.render_dtype == torch.float32):
            graphed = self._forward_graphed(images, K, labels, TCO, n_iterations, im_ids, slot)
            if graphed is not None:
                return graphed
        outputs: Dict[str, PosePredictorOutput] = dict()
        TCO_input = TCO
        nin = self._n_input_channels
        for n in range(n_iterations):
            st = self._step(images, im_ids, K, labels, TCO_input, want_sigmoid=False, slot=slot, events=cuda_timer)
            K_crop = st["K_crop"]   # crop_inputs' intrinsics (== KV_crop[:, 0] unless remove_TCO_rendering)
            if self.predict_pose_update:
                TCO_output = eng.pose_update(st["TCO_n"], K_crop, st["out"], st["tCR"], 9)
                network_outputs = {"pose": st["out"]}
                renderings_logits = torch.empty(bsz, self.n_rendered_views, dtype=TCO.dtype, device=device)
            else:
                TCO_output = st["TCO_n"].clone()
                network_outputs = {"renderings_logits": st["out"]}
                renderings_logits = st["out"]
            renders = images_crop = None
            if materialize:
                images_crop = self._nchw_view(bsz, 0, nin, slot).clone()
                renders = self._nchw_view(bsz, nin, self.backbone.n_inputs, slot).clone()
            outputs[f"iteration={n + 1}"] = PosePredictorOutput(
                renders=renders, images_crop=images_crop, TCO_input=st["TCO_n"], TCO_output=TCO_output, TCV_O_input=st["TCV_O"],
                tCR=st["tCR"], labels=labels, K=K, K_crop=K_crop, KV_crop=st["KV_crop"], network_outputs=network_outputs,
                boxes_rend=st["boxes_rend"], boxes_crop=st["boxes_crop"], renderings_logits=renderings_logits,
                timing_dict={"render": st["render_time"], "events": st["events"]})
            TCO_input = TCO_output
        return outputs

    # -- hipGraph path of small refiner calls -----------------------------------------------------------------------
    def _graph_widths(self):
        V = self.n_rendered_views
        return (("TCO_n", 16), ("TCO_out", 16), ("K_crop", 9), ("KV_crop", 9 * V), ("boxes_rend", 4), ("boxes_crop", 4), ("out", 9),
                ("tCR", 3), ("TCV_O", 16 * V))

    def _forward_graphed(self, images, K, labels, TCO, n_iterations, im_ids, slot) -> Optional[Dict[str, PosePredictorOutput]]:
        """The same call as the loop in forward(), captured once as a hipGraph (torch.cuda.CUDAGraph = hipStreamBeginCapture /
        hipGraphLaunch around this library's launches) and replayed: the first call of a shape runs eagerly (it also warms every
        lazily-sized workspace), the second captures, later ones copy the inputs into the graph's static buffers, replay, and clone the
        packed per-iteration results.  Returns None while the shape is still in its eager warm-up call."""
        device = TCO.device
        b = TCO.shape[0]
        key = (b, n_iterations, slot, tuple(images.shape), str(device), self._mv_mode)
        packed = self._packed(images)
        ids = self._ids(labels, device)
        g = self._graphs.get(key)
        if g is None:
            seen = self._graph_seen.get(key, 0)
            self._graph_seen[key] = seen + 1
            if seen == 0:
                return None
            if len(self._graphs) >= 32:
                for old in list(self._graphs):
                    self._drop_graph(old)
            g = self._graphs[key] = _RefinerGraph(self, b, n_iterations, slot, packed, device)
        res = g.run(TCO, K, im_ids, ids, packed)
        V = self.n_rendered_views
        outputs: Dict[str, PosePredictorOutput] = dict()
        for n in range(n_iterations):
            f, o = {}, 0
            for name, wd in self._graph_widths():
                f[name] = res[n, :, o:o + wd]
                o += wd
            outputs[f"iteration={n + 1}"] = PosePredictorOutput(
                renders=None, images_crop=None, TCO_input=f["TCO_n"].reshape(b, 4, 4), TCO_output=f["TCO_out"].reshape(b, 4, 4),
                TCV_O_input=f["TCV_O"].reshape(b, V, 4, 4), tCR=f["tCR"], labels=labels, K=K, K_crop=f["K_crop"].reshape(b, 3, 3),
                KV_crop=f["KV_crop"].reshape(b, V, 3, 3), network_outputs={"pose": f["out"]}, boxes_rend=f["boxes_rend"],
                boxes_crop=f["boxes_crop"], renderings_logits=torch.empty(b, V, dtype=TCO.dtype, device=device),
                timing_dict={"render": 0.0, "events": None})
        return outputs

    @torch.no_grad()
    def forward_coarse(self, images: torch.Tensor, K: torch.Tensor, labels: List[str], TCO_input: torch.Tensor,
                       cuda_timer: bool = False, return_debug_data: bool = False,
                       im_ids: Optional[torch.Tensor] = None, slot: int = 0, defer_timing: bool = False) -> Dict[str, Any]:
        """pose_rigid.py:634-708: logits/scores [b,1] of each hypothesis."""
        assert self.predict_rendered_views_logits, "Method only valid if coarse classification model"
        images = self._prep_images(images)
        bsz = TCO_input.shape[0]
        assert TCO_input.shape == (bsz, 4, 4) and K.shape == (bsz, 3, 3) and len(labels) == bsz
        if im_ids is None:
            assert images.shape[0] == bsz
            im_ids = torch.arange(bsz, dtype=torch.int32, device=TCO_input.device)
        st = self._step(images, im_ids, K, labels, TCO_input, want_sigmoid=True, slot=slot, events=cuda_timer, mv_mode=0)   # forward_coarse: KV_crop = K_crop (pose_rigid.py:683-685)
        # cuda_timer=True: `events` lets the caller resolve DEVICE render / model times once per stage (PoseEstimator does, after
        # its single synchronisation); a direct caller gets them here at the price of a synchronise, like the reference's
        # CudaTimer.end() (training/utils.py:224-264).  cuda_timer=False: model_time = 0.0 as in the reference, render_time =
        # host time to enqueue the (asynchronous) render launch.
        out = {"logits": st["out"], "scores": st["sigmoid"], "time": 0.0, "render_time": st["render_time"], "model_time": 0.0,
               "TCO_n": st["TCO_n"], "K_crop": st["K_crop"], "boxes_rend": st["boxes_rend"], "boxes_crop": st["boxes_crop"],
               "events": st["events"]}
        if cuda_timer and not defer_timing:
            torch.cuda.synchronize()
            out["render_time"], out["model_time"] = self.step_times(st["events"])
            out["time"] = out["model_time"]
        if return_debug_data:
            nin = self._n_input_channels
            out["images_crop"] = self._nchw_view(bsz, 0, nin, slot).clone()
            out["renders"] = self._nchw_view(bsz, nin, self.backbone.n_inputs, slot).clone()
        return out

    @torch.no_grad()
    def net_forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """pose_rigid.py:314-334 on an already assembled NCHW input [b, n_inputs, h, w] (API-compat path: one layout copy)."""
        bb = self._backbone_engine()
        b, c, h, w = x.shape
        assert c == self.backbone.n_inputs
        buf = eng.padded_nhwc(b, h, w, bb.c_in_p, bb.in_border, x.device)
        eng.padded_view(buf, b, h, w, bb.c_in_p, bb.in_border)[..., :c] = x.permute(0, 2, 3, 1)
        out = torch.empty(b, bb.n_out, dtype=torch.float32, device=x.device)
        bb.forward(buf, b, h, w, out)
        return {"pose" if self.predict_pose_update else "renderings_logits": out}

    @torch.no_grad()
    def forward_coarse_tensor(self, x: torch.Tensor, cuda_timer: bool = False) -> Dict[str, Union[torch.Tensor, float]]:
        assert self.predict_rendered_views_logits, "Method only valid if coarse classification model"
        logits = self.net_forward(x)["renderings_logits"]
        return {"logits": logits, "scores": torch.sigmoid(logits), "time": 0.0}

    @torch.no_grad()
    def crop_inputs(self, images: torch.Tensor, K: torch.Tensor, TCO: torch.Tensor, tCR: torch.Tensor, labels: List[str]):
        """pose_rigid.py:180-247 -> (images_cropped, K_crop, boxes_rend, boxes_crop).  tCR must be TCO's translation
        (the only way the hot path calls it)."""
        bsz = TCO.shape[0]
        pts_ids, _ = self._ids(labels, TCO.device)
        h, w = self.render_size
        _, _, _, KV, brend, bcrop = eng.pose_prepare(TCO, K, pts_ids, self.mesh_db.sampled_points(2000), 2000, 200, 1, 0,
                                                     tuple(images.shape[-2:]), (h, w), 1.4)
        C = images.shape[1]
        out = torch.empty(bsz, h, w, C, dtype=torch.float32, device=TCO.device)
        eng.crop_roi_align(images, torch.arange(bsz, dtype=torch.int32, device=TCO.device), bcrop, h, w, out, h * w * C, w * C, C, 0)
        return out.permute(0, 3, 1, 2), KV[:, 0], brend, bcrop

    @property
    def input_rgb_dims(self) -> List[int]:
        return self._input_rgb_dims

    @property
    def input_depth_dims(self) -> List[int]:
        return self._input_depth_dims

    @property
    def render_rgb_dims(self) -> List[int]:
        return self._render_rgb_dims

    @property
    def render_depth_dims(self) -> List[int]:
        return self._render_depth_dims

    @torch.no_grad()
    def compute_crops_multiview(self, images: torch.Tensor, K: torch.Tensor, TCV_O: torch.Tensor, tCR: torch.Tensor,
                                labels: List[str]) -> torch.Tensor:
        """pose_rigid.py:249-303 -> K_crop [bsz, n_views, 3, 3] of the virtual cameras (200 sampled points per view).  On the hot path
        this is part of pose_prepare; the stand-alone call requires tCR == translation of TCV_O, which is how the reference calls it."""
        bsz, n_views = TCV_O.shape[:2]
        assert tCR.shape == (bsz, n_views, 3) and TCV_O.shape == (bsz, n_views, 4, 4) and K.shape == (bsz, 3, 3)
        if (tCR - TCV_O[..., :3, 3]).abs().max().item() > 1e-6:
            raise NotImplementedError("compute_crops_multiview: anchor points other than the views' translations")
        pts_ids, _ = self._ids([l for l in labels for _ in range(n_views)], TCV_O.device)
        _, _, _, KV, _, _ = eng.pose_prepare(TCV_O.flatten(0, 1), K.unsqueeze(1).repeat(1, n_views, 1, 1).flatten(0, 1), pts_ids,
                                             self.mesh_db.sampled_points(2000), 200, 200, 1, 0, tuple(images.shape[-2:]), self.render_size, 1.4)
        return KV[:, 0].reshape(bsz, n_views, 3, 3)

    @torch.no_grad()
    def normalize_depth(self, depth: torch.Tensor, tCR: torch.Tensor) -> torch.Tensor:
        """pose_rigid.py:466-496: depth [B, ..., H, W] normalised by the anchor depth tCR[:, 2]; returns a new tensor"""
        mode_name = self.depth_normalization_type
        if mode_name == "tCR_center_obj_diam":
            raise NotImplementedError("Not yet implemented")
        if mode_name not in eng.DEPTH_NORM_MODES:
            raise ValueError(f"Unknown depth_normalization_type = {mode_name}")
        out = depth.detach().to(dtype=torch.float32).contiguous().clone()
        mode = eng.DEPTH_NORM_MODES[mode_name]
        if mode and out.numel():
            B, H, W = out.shape[0], out.shape[-2], out.shape[-1]
            k = out.numel() // (B * H * W)
            eng.normalize_depth(out, B * k, H, W, 0, 1, [0], tCR.to(out.device, torch.float32).repeat_interleave(k, dim=0), mode)
        return out

    @torch.no_grad()
    def normalize_images(self, images: torch.Tensor, renders: torch.Tensor, tCR: torch.Tensor, images_inplace: bool = False,
                         renders_inplace: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """pose_rigid.py:410-464 (the hot path normalises in place inside the CNN input buffer instead)"""
        if not images_inplace:
            images = images.clone()
        if not renders_inplace:
            renders = renders.clone()
        if self.input_depth:
            assert images.shape[1] == 4, "images must have C=4 channels if input_depth=True"
            images[:, self._input_depth_dims] = self.normalize_depth(images[:, self._input_depth_dims], tCR)
        if self.render_depth:
            dims = [self._render_depth_dims[0] + self._n_single_render_channels * v for v in range(self.n_rendered_views)]
            renders[:, dims] = self.normalize_depth(renders[:, dims], tCR)
        return images, renders

    @torch.no_grad()
    def update_pose(self, TCO: torch.Tensor, K_crop: torch.Tensor, pose_outputs: torch.Tensor, tCR: torch.Tensor) -> torch.Tensor:
        assert pose_outputs.shape[-1] == 9
        return eng.pose_update(TCO, K_crop.contiguous(), pose_outputs, tCR, 9)

    @torch.no_grad()
    def render_images_multiview(self, labels: List[str], TCV_O: torch.Tensor, KV: torch.Tensor, random_ambient_light: bool = False):
        """pose_rigid.py:336-408: [bsz, n_views*n_channels, H, W]"""
        bsz, n_views = TCV_O.shape[:2]
        labels_mv = [l for l in labels for _ in range(n_views)]
        data = self.renderer.render(labels=labels_mv, TCO=TCV_O.flatten(0, 1), K=KV.flatten(0, 1), render_mask=False,
                                    resolution=self.render_size, render_normals=self.render_normals, render_depth=self.render_depth,
                                    light_datas=[self._lights() for _ in labels_mv])
        cat = [data.rgbs] + ([data.normals] if self.render_normals else []) + ([data.depths] if self.render_depth else [])
        renders = torch.cat(cat, dim=1)
        return renders.view(bsz, n_views, renders.shape[1], *renders.shape[-2:]).flatten(1, 2)
