"""Thin torch-tensor front-end over the C-ABI (include/mp_engine.h).

PyTorch is used for device memory and streams only; every computation below is a
call into libmp_engine.so.  All functions enqueue on torch's current HIP stream and
never synchronise.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, EngineError, Lights, MeshDesc, NamedTensor, check

RASTER_NORMALS = 1
RASTER_DEPTH = 2
RASTER_NORMALS_GL = 4
RASTER_MSAA4 = 16   # 4x multisampling = the reference renderer's configuration (panda3d_scene_renderer.py:73-74)
RASTER_F16 = 32     # "fp16 renders": the output tensor holds binary16 elements (set from `out.dtype`, never by hand)
RASTER_XREC = 64

BACKBONE_KINDS = {"vanilla_resnet34": 0, "resnet34": 1, "resnet18": 2}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev_f32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise EngineError("engine tensors must live on the GPU")
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t


def _dev_i32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise EngineError("engine tensors must live on the GPU")
    if t.dtype != torch.int32 or not t.is_contiguous():
        t = t.to(torch.int32).contiguous()
    return t


def clock_probe(ms_target: float = 20.0) -> Dict[str, float]:
    """Effective shader clock (MHz) under fp32-MFMA load and the probe loop's own TFLOP/s (synchronises; mp_clock_probe)."""
    mhz, tf = C.c_double(0), C.c_double(0)
    check(_lib.load().mp_clock_probe(float(ms_target), C.byref(mhz), C.byref(tf), _stream()))
    return {"shader_mhz": mhz.value, "mfma_tflops": tf.value}


def conv_clock(reset: bool = True) -> float:
    """Effective shader clock (MHz) inside the convolution kernels since the last reset (synchronises; mp_conv_clock_read)."""
    mhz = C.c_double(0)
    check(_lib.load().mp_conv_clock_read(C.byref(mhz), 1 if reset else 0))
    return mhz.value


def conv_wino_stats(reset: bool = True) -> Tuple[float, float]:
    """(algorithmic = direct-convolution FLOPs, FLOPs actually executed) of the Winograd launches since the last reset"""
    a, b = C.c_double(0), C.c_double(0)
    check(_lib.load().mp_conv_wino_stats(C.byref(a), C.byref(b), 1 if reset else 0))
    return a.value, b.value


_PROFILING = False


def profiling() -> bool:
    """True between profile_begin() and profile_end() (the per-launch event profiler is on: graph capture is then avoided)"""
    return _PROFILING


def profile_begin() -> None:
    global _PROFILING
    check(_lib.load().mp_profile_begin())
    _PROFILING = True


def profile_end() -> Dict[str, Dict[str, float]]:
    """Stop the per-launch event profiler and return {kernel: {launches, ms, flops, bytes, executed, peak_tflops}} (synchronises):
    flops = algorithmic work, executed = FLOPs issued on the matrix pipe whose dense peak is peak_tflops (mp_profile_query_ex)."""
    global _PROFILING
    lib = _lib.load()
    check(lib.mp_profile_end())
    _PROFILING = False
    out: Dict[str, Dict[str, float]] = {}
    i = 0
    while True:
        name = C.create_string_buffer(128)
        n, ms, fl, by, ex, pk = C.c_int64(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
        rc = lib.mp_profile_query_ex(i, name, 128, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by), C.byref(ex), C.byref(pk))
        if rc == 1:
            break
        check(rc)
        out[name.value.decode()] = {"launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value, "executed": ex.value,
                                    "peak_tflops": pk.value}
        i += 1
    return out


def device_info() -> Tuple[int, int, str]:
    lib = _lib.load()
    n_cu, lds = C.c_int(0), C.c_int(0)
    name = C.create_string_buffer(64)
    check(lib.mp_device_info(C.byref(n_cu), C.byref(lds), name, 64))
    return n_cu.value, lds.value, name.value.decode()


# --------------------------------------------------------------------------- #
class MeshDB:
    """Device-resident meshes (mp_mesh_db).  `meshes`: list of dicts with float32 arrays
    vertices [V,3] (metres), normals [V,3], colors [V,3] in [0,1], int32 faces [T,3]; optionally uvs [T,3,2] and
    texture_mips (list of uint32 [h_l, w_l] RGBA8 levels) for UV-textured objects."""

    def __init__(self, meshes: Sequence[Dict[str, np.ndarray]]):
        lib = _lib.load()
        self._keep = []
        descs = (MeshDesc * len(meshes))()
        for i, m in enumerate(meshes):
            v = np.ascontiguousarray(m["vertices"], dtype=np.float32)
            n = np.ascontiguousarray(m["normals"], dtype=np.float32)
            c = np.ascontiguousarray(m["colors"], dtype=np.float32)
            f = np.ascontiguousarray(m["faces"], dtype=np.int32)
            assert v.shape == n.shape == c.shape and v.shape[1] == 3 and f.shape[1] == 3
            self._keep += [v, n, c, f]
            descs[i] = MeshDesc(v.ctypes.data, n.ctypes.data, c.ctypes.data, f.ctypes.data, v.shape[0], f.shape[0])
        h = C.c_void_p()
        check(lib.mp_mesh_db_create(descs, len(meshes), C.byref(h)))
        self.handle = h
        self.n = len(meshes)
        self.max_vertices = lib.mp_mesh_db_max_vertices(h)
        for i, m in enumerate(meshes):  # optional UV texture: per-corner uvs [T,3,2] + RGBA8 mip chain (mesh_io.build_mip_chain)
            if m.get("uvs") is not None and m.get("texture_mips") is not None:
                uv = np.ascontiguousarray(m["uvs"], dtype=np.float32)
                mips = m["texture_mips"]
                assert uv.shape == (np.asarray(m["faces"]).shape[0], 3, 2)
                th, tw = mips[0].shape[:2]
                flat = np.ascontiguousarray(np.concatenate([lv.reshape(-1) for lv in mips]).astype(np.uint32))
                check(lib.mp_mesh_db_set_texture(h, i, uv.ctypes.data, flat.ctypes.data, tw, th, len(mips)))
        self._keep = []
        self._ws: Dict[int, torch.Tensor] = {}

    def radius(self, i: int) -> float:
        return _lib.load().mp_mesh_db_radius(self.handle, i)

    def workspace(self, n_views: int, h: int, w: int, device, slot: int = 0) -> torch.Tensor:
        """scratch for the per-view tile lists of a launch; one per `slot` (= concurrent HIP stream)"""
        need = _lib.load().mp_raster_workspace_bytes(self.handle, n_views, h, w)
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need or ws.device != torch.device(device):
            self._ws[slot] = ws = torch.empty(max(need, 1), dtype=torch.uint8, device=device)
        return ws

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().mp_mesh_db_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_lights(ambient=(1.0, 1.0, 1.0), point_dirs=(), point_colors=(), point_offsets=None) -> Lights:
    """point light i sits at dir_i * 10 * (bounding radius of the mesh) + offset_i in the object frame"""
    L = Lights()
    L.ambient[:] = ambient
    L.n_point = len(point_dirs)
    for i, (d, c) in enumerate(zip(point_dirs, point_colors)):
        L.point_dir[i][:] = d
        L.point_color[i][:] = c
        L.point_offset[i][:] = point_offsets[i] if point_offsets is not None else (0.0, 0.0, 0.0)
    return L


class PackedObservation:
    """Observation frames repacked [n_im,H,W,4] on the device (mp_pack_observation_nhwc4) for the fused crop of raster_render."""

    def __init__(self, images: torch.Tensor):
        images = _dev_f32(images)
        self.n_im, self.C, self.H, self.W = (int(v) for v in images.shape)
        self.data = torch.empty(self.n_im, self.H, self.W, 4, dtype=torch.float32, device=images.device)
        check(_lib.load().mp_pack_observation_nhwc4(images.data_ptr(), self.n_im, self.C, self.H, self.W, self.data.data_ptr(), _stream()))


def raster_render(db: MeshDB, mesh_ids: torch.Tensor, TCO: torch.Tensor, K: torch.Tensor, h: int, w: int, flags: int,
                  lights: Lights, out: torch.Tensor, stride_v: int, stride_y: int, stride_x: int, c_rgb: int, c_normals: int,
                  c_depth: int, out_offset_floats: int = 0, views_per_item: int = 1, stride_view: int = 0, slot: int = 0,
                  crop=None, xrec=None) -> None:
    """Render n views into `out` (float32 or float16 device tensor) at the given ELEMENT strides / offset; a float16 `out`
    selects MP_RASTER_F16 (values rounded to nearest-even binary16 as they are stored -- the "fp16 renders" mode).
    crop = (images [n_im,C,H,W], im_ids [n_items], boxes [n_items,4], c0): also roi_align-crop every item's observation into channels
    c0.. of its pixels in the same launch (mp_raster_render_crop).  xrec = (f32_mask, tCR, depth_mode) with a bfloat16 `out`: stem records
    of a model WITH depth channels, depth normalised in the launch (mp_raster_render_xrec)."""
    lib = _lib.load()
    n = int(TCO.shape[0])
    mesh_ids = _dev_i32(mesh_ids)
    TCO = _dev_f32(TCO)
    K = _dev_f32(K)
    assert out.dtype in (torch.float32, torch.float16, torch.bfloat16) and out.is_cuda
    es = out.element_size()
    flags = (flags | RASTER_F16) if out.dtype == torch.float16 else (flags & ~RASTER_F16)
    # a bfloat16 `out` = the stem RECORDS of the exact-piece stem convolution (MP_RASTER_XREC; strides / offset in bf16 elements)
    flags = (flags | RASTER_XREC) if out.dtype == torch.bfloat16 else (flags & ~RASTER_XREC)
    ws = db.workspace(n, h, w, out.device, slot)
    if crop is not None:
        images, im_ids, boxes, c0 = crop
        if isinstance(images, PackedObservation):
            nhwc4, n_im, Cc, H, W, images = 1, images.n_im, images.C, images.H, images.W, images.data
        else:
            images = _dev_f32(images)
            nhwc4 = 0
            n_im, Cc, H, W = images.shape
        im_ids, boxes = _dev_i32(im_ids), _dev_f32(boxes)
        assert boxes.shape[0] * views_per_item == n and im_ids.shape[0] == boxes.shape[0]
        if xrec is not None:   # records with depth channels: (f32_mask, tCR [n_items, 3], depth mode) -> mp_raster_render_xrec
            assert out.dtype == torch.bfloat16 and c0 == 0
            f32_mask, tCR, depth_mode = xrec
            tCR = _dev_f32(tCR) if tCR is not None else None
            check(lib.mp_raster_render_xrec(db.handle, mesh_ids.data_ptr(), TCO.data_ptr(), K.data_ptr(), n, h, w, flags & ~RASTER_XREC,
                                            C.byref(lights), out.data_ptr() + es * out_offset_floats, stride_v, views_per_item, stride_view,
                                            stride_y, stride_x, c_rgb, c_normals, c_depth, ws.data_ptr(), ws.numel(), images.data_ptr(), nhwc4,
                                            n_im, Cc, H, W, im_ids.data_ptr(), boxes.data_ptr(), int(f32_mask) & 0xFFFFFFFF, _ptr(tCR),
                                            int(depth_mode), _stream()))
            return
        check(lib.mp_raster_render_crop(db.handle, mesh_ids.data_ptr(), TCO.data_ptr(), K.data_ptr(), n, h, w, flags, C.byref(lights),
                                        out.data_ptr() + es * out_offset_floats, stride_v, views_per_item, stride_view, stride_y, stride_x,
                                        c_rgb, c_normals, c_depth, ws.data_ptr(), ws.numel(), images.data_ptr(), nhwc4, n_im, Cc, H, W,
                                        im_ids.data_ptr(), boxes.data_ptr(), c0, _stream()))
        return
    check(lib.mp_raster_render(db.handle, mesh_ids.data_ptr(), TCO.data_ptr(), K.data_ptr(), n, h, w, flags, C.byref(lights),
                               out.data_ptr() + es * out_offset_floats, stride_v, views_per_item, stride_view, stride_y, stride_x, c_rgb,
                               c_normals, c_depth, ws.data_ptr(), ws.numel(), _stream()))


def raster_job_flags(db: MeshDB, n_views: int, h: int, w: int, device, slot: int = 0) -> int:
    """Device address of the job flags the LAST compacted raster launch on this database's workspace (`slot`) wrote: one byte per
    (item, 8x8-pixel tile), 0 = no view of the item reaches the tile (mp_raster_job_flags).  Valid on the same stream until the next
    raster launch on that slot; 0 if unavailable."""
    ws = db.workspace(n_views, h, w, device, slot)
    return int(_lib.load().mp_raster_job_flags(db.handle, ws.data_ptr(), n_views, h, w) or 0)


def crop_roi_align(images: torch.Tensor, im_ids: torch.Tensor, boxes: torch.Tensor, out_h: int, out_w: int, out: torch.Tensor,
                   stride_b: int, stride_y: int, stride_x: int, c0: int, out_offset_floats: int = 0) -> None:
    lib = _lib.load()
    images = _dev_f32(images)
    n_im, Cc, H, W = images.shape
    check(lib.mp_crop_roi_align(images.data_ptr(), n_im, Cc, H, W, _dev_i32(im_ids).data_ptr(), _dev_f32(boxes).data_ptr(),
                                int(boxes.shape[0]), out_h, out_w, out.data_ptr() + 4 * out_offset_floats, stride_b, stride_y,
                                stride_x, c0, _stream()))


def normalize_depth(x: torch.Tensor, b: int, h: int, w: int, border: int, Cp: int, channels: Sequence[int], tCR: torch.Tensor,
                    mode: int) -> None:
    lib = _lib.load()
    ch = (C.c_int32 * len(channels))(*channels)
    fn = lib.mp_normalize_depth_f16 if x.dtype == torch.float16 else lib.mp_normalize_depth
    check(fn(x.data_ptr(), b, h, w, border, Cp, ch, len(channels), _dev_f32(tCR).data_ptr(), mode, _stream()))


DEPTH_NORM_MODES = {None: 0, "none": 0, "tCR_scale": 1, "tCR_scale_clamp_center": 2, "tCR_center_clamp": 3}


# --------------------------------------------------------------------------- #
def padded_nhwc(n: int, h: int, w: int, c: int, border: int, device, slack: Optional[int] = None,
                dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Zero-initialised flat buffer holding a padded-NHWC tensor + read slack: the conv's last 32-float K chunk may run
    past the kernel window into the next padded row (its weights are zero there, the memory only has to be readable).
    dtype float16 = the half-precision CNN input of the "fp16 renders" mode (same element geometry)."""
    if slack is None:
        slack = (w + 2 * border) * c + 64
    return torch.zeros(n * (h + 2 * border) * (w + 2 * border) * c + slack, dtype=dtype, device=device)


def padded_view(buf: torch.Tensor, n: int, h: int, w: int, c: int, border: int) -> torch.Tensor:
    """[n, h, w, c] view of the interior of a padded-NHWC buffer."""
    full = buf[: n * (h + 2 * border) * (w + 2 * border) * c].view(n, h + 2 * border, w + 2 * border, c)
    return full[:, border : border + h, border : border + w, :]


def conv_pack_weights(w_oihw: np.ndarray, cin_p: int, scale: Optional[np.ndarray] = None) -> np.ndarray:
    lib = _lib.load()
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    Cout, Cin, KH, KW = w.shape
    n = lib.mp_conv_packed_floats(cin_p, Cout, KH, KW)
    out = np.empty(n, dtype=np.float32)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    check(lib.mp_conv_pack_weights(w.ctypes.data, Cout, Cin, KH, KW, cin_p, None if sc is None else sc.ctypes.data, out.ctypes.data))
    return out


def conv2d_nhwc(x: torch.Tensor, N: int, H: int, W: int, Cp: int, in_border: int, w_packed: torch.Tensor,
                bias: Optional[torch.Tensor], Cout: int, K: int, stride: int, pad: int, y: Optional[torch.Tensor], out_border: int,
                residual: Optional[torch.Tensor] = None, relu: bool = False, y_act: Optional[torch.Tensor] = None,
                act_scale: Optional[torch.Tensor] = None, act_shift: Optional[torch.Tensor] = None,
                splitk_ws: Optional[torch.Tensor] = None) -> None:
    """`splitk_ws` (fp32 scratch): lets launches whose tile grid cannot fill the chip split the K loop (deterministic two-pass).
    A float16 `x` (same padded-NHWC geometry) selects the half-precision input path (mp_conv_desc.x_f16; Cout <= 64 only)."""
    lib = _lib.load()
    d = ConvDesc()
    assert x.dtype in (torch.float32, torch.float16)
    d.x_f16 = int(x.dtype == torch.float16)
    if splitk_ws is not None:
        d.d_splitk_ws, d.splitk_ws_floats = splitk_ws.data_ptr(), splitk_ws.numel()
    d.d_x, d.N, d.H, d.W, d.C, d.in_border = x.data_ptr(), N, H, W, Cp, in_border
    d.d_w, d.d_bias = w_packed.data_ptr(), _ptr(bias)
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, K, K, stride, pad
    d.d_y, d.out_border, d.d_residual, d.relu = _ptr(y), out_border, _ptr(residual), int(relu)
    d.d_y_act, d.d_act_scale, d.d_act_shift = _ptr(y_act), _ptr(act_scale), _ptr(act_shift)
    check(lib.mp_conv2d_nhwc(C.byref(d), _stream()))


def conv_wino_pack_weights(w_oihw: np.ndarray, cin_p: int, scale: Optional[np.ndarray] = None) -> np.ndarray:
    """Winograd-transformed weights U = G g G^T of a 3x3 layer in MFMA fragment order (mp_conv_wino_pack_weights)"""
    lib = _lib.load()
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    Cout, Cin, KH, KW = w.shape
    assert KH == 3 and KW == 3
    out = np.empty(lib.mp_conv_wino_packed_floats(cin_p, Cout), dtype=np.float32)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    check(lib.mp_conv_wino_pack_weights(w.ctypes.data, Cout, Cin, cin_p, None if sc is None else sc.ctypes.data, out.ctypes.data))
    return out


def conv3x3_wino_nhwc(x: torch.Tensor, N: int, H: int, W: int, Cp: int, in_border: int, u_packed: torch.Tensor,
                      bias: Optional[torch.Tensor], Cout: int, y: Optional[torch.Tensor], out_border: int,
                      residual: Optional[torch.Tensor] = None, relu: bool = False, y_act: Optional[torch.Tensor] = None,
                      act_scale: Optional[torch.Tensor] = None, act_shift: Optional[torch.Tensor] = None) -> None:
    """Fused Winograd F(2x2, 3x3) form of a 3x3 / stride-1 / pad-1 convolution (mp_conv3x3_wino_nhwc).  `x` must carry
    (W + 2 * in_border + 1) * Cp floats of readable slack behind the tensor when H or W is odd."""
    d = ConvDesc()
    d.d_x, d.N, d.H, d.W, d.C, d.in_border = x.data_ptr(), N, H, W, Cp, in_border
    d.d_bias = _ptr(bias)
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, 3, 3, 1, 1
    d.d_y, d.out_border, d.d_residual, d.relu = _ptr(y), out_border, _ptr(residual), int(relu)
    d.d_y_act, d.d_act_scale, d.d_act_shift = _ptr(y_act), _ptr(act_scale), _ptr(act_shift)
    if u_packed.dtype == torch.uint8:   # the three-bf16-piece blob: the exact-piece kernel (mp_conv3x3_wino_bf16_nhwc)
        check(_lib.load().mp_conv3x3_wino_bf16_nhwc(C.byref(d), u_packed.data_ptr(), _stream()))
        return
    check(_lib.load().mp_conv3x3_wino_nhwc(C.byref(d), u_packed.data_ptr(), _stream()))


def conv_wino_bf16_pack_weights(w_oihw: np.ndarray, cin_p: int, scale: Optional[np.ndarray] = None) -> np.ndarray:
    """U = G g G^T of a 3x3 layer split into three exact bf16 pieces, MFMA fragment order (mp_conv_wino_bf16_pack_weights); uint8 blob"""
    lib = _lib.load()
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    Cout, Cin, KH, KW = w.shape
    assert KH == 3 and KW == 3
    out = np.empty(lib.mp_conv_wino_bf16_packed_bytes(cin_p, Cout), dtype=np.uint8)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    check(lib.mp_conv_wino_bf16_pack_weights(w.ctypes.data, Cout, Cin, cin_p, None if sc is None else sc.ctypes.data, out.ctypes.data))
    return out


def conv_wino_bf16_telemetry(on: bool) -> bool:
    """in-kernel clock telemetry of the bf16 Winograd kernel (every 64th workgroup, six global atomics): off by default; returns the
    previous setting (mp_conv_wino_bf16_telemetry)"""
    return bool(_lib.load().mp_conv_wino_bf16_telemetry(int(bool(on))))


def conv_wino_bf16_stats(reset: bool = True) -> Tuple[float, float]:
    """(algorithmic = direct-convolution FLOPs, executed bf16 FLOPs) of the bf16x9 Winograd launches since the last reset"""
    a, b = C.c_double(0), C.c_double(0)
    check(_lib.load().mp_conv_wino_bf16_stats(C.byref(a), C.byref(b), 1 if reset else 0))
    return a.value, b.value


def conv_wino_bf16_clock(reset: bool = True) -> Tuple[float, float]:
    """(effective shader clock in MHz, shader cycles per 16-channel step) inside the K loops of the bf16x9 Winograd launches"""
    a, b = C.c_double(0), C.c_double(0)
    check(_lib.load().mp_conv_wino_bf16_clock(C.byref(a), C.byref(b), 1 if reset else 0))
    return a.value, b.value


def leading_mask(n_f32: int) -> int:
    """fp32-kind channel mask of a record whose first `n_f32` channels are the fp32-kind ones"""
    return (1 << int(n_f32)) - 1


def conv_stem_pack_weights(w_oihw: np.ndarray, n_f32: int, scale: Optional[np.ndarray] = None, f32_mask: Optional[int] = None) -> np.ndarray:
    """three exact bf16 pieces of every stem weight (BN scale and, for the integer channels, 1/255 folded in) in MFMA fragment order
    (mp_conv_stem_pack_weights[_mask]); the first `n_f32` input channels -- or, with `f32_mask`, the channels whose bit is set -- are
    fp32-kind, the others 8-bit integers.  Returns a uint8 blob."""
    lib = _lib.load()
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    Cout, Cin, KH, KW = w.shape
    assert KH == KW
    mask = leading_mask(n_f32) if f32_mask is None else int(f32_mask)
    nf = bin(mask).count("1")
    out = np.empty(lib.mp_conv_stem_packed_bytes(KH, nf, Cin - nf, Cout), dtype=np.uint8)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    check(lib.mp_conv_stem_pack_weights_mask(w.ctypes.data, Cout, Cin, KH, mask, None if sc is None else sc.ctypes.data, out.ctypes.data))
    return out


def xrec_elements(n_f32: int, n_u8: int) -> int:
    return int(_lib.load().mp_xrec_elements(n_f32, n_u8))


def conv_stem_pack_weights_sparse(w_oihw: np.ndarray, n_f32: int, scale: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
    """the piece blob of the stem's BACKGROUND-TILE walk (only the record chunks that hold fp32-kind pieces; mp_conv_stem_pack_weights_sparse);
    None if this record has no such form (nothing to skip)"""
    lib = _lib.load()
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    Cout, Cin, KH, KW = w.shape
    n = lib.mp_conv_stem_sparse_packed_bytes(KH, n_f32, Cin - n_f32, Cout)
    if n == 0:
        return None
    out = np.empty(n, dtype=np.uint8)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    check(lib.mp_conv_stem_pack_weights_sparse(w.ctypes.data, Cout, Cin, KH, n_f32, None if sc is None else sc.ctypes.data, out.ctypes.data))
    return out


def conv_stem_bg_stats(reset: bool = True) -> Tuple[float, float]:
    """(workgroups that took the background-tile walk, all workgroups of such launches) counted while the event profiler was active"""
    a, b = C.c_double(), C.c_double()
    check(_lib.load().mp_conv_stem_bg_stats(C.byref(a), C.byref(b), int(reset)))
    return a.value, b.value


def conv_stem_xrec(xrec: torch.Tensor, N: int, H: int, W: int, c_real: int, n_f32: int, in_border: int, w_pieces: torch.Tensor,
                   bias: Optional[torch.Tensor], Cout: int, K: int, pad: int, y: Optional[torch.Tensor], out_border: int, relu: bool = False,
                   y_pool: Optional[torch.Tensor] = None, pool_border: int = 1, w_sparse: Optional[torch.Tensor] = None,
                   tile_flags: Optional[torch.Tensor] = None) -> None:
    """stride-2 stem convolution of a bfloat16 record tensor (mp_conv_stem_xrec); with `y_pool` the 3x3 / stride-2 / pad-1 max pool of its
    output is written too (mp_conv_stem_xrec_pool; `y` may then be None); with `w_sparse` + `tile_flags` (uint8 [N, ceil(H/8), ceil(W/8)],
    0 = the tile's integer channels are all 0) workgroups over background take the short walk (mp_conv_stem_xrec_sparse)"""
    assert xrec.dtype == torch.bfloat16 and w_pieces.dtype == torch.uint8
    d = ConvDesc()
    d.d_x, d.N, d.H, d.W, d.C, d.c_real, d.in_border = xrec.data_ptr(), N, H, W, (c_real + 3) // 4 * 4, c_real, in_border
    d.d_bias = _ptr(bias)
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, K, K, 2, pad
    d.d_y, d.out_border, d.relu = _ptr(y), out_border, int(relu)
    if w_sparse is not None and tile_flags is not None:
        assert tile_flags.dtype == torch.uint8 and tile_flags.is_cuda and w_sparse.dtype == torch.uint8
        check(_lib.load().mp_conv_stem_xrec_sparse(C.byref(d), w_pieces.data_ptr(), w_sparse.data_ptr(), n_f32, tile_flags.data_ptr(),
                                                   _ptr(y_pool), pool_border, _stream()))
        return
    if y_pool is not None:
        check(_lib.load().mp_conv_stem_xrec_pool(C.byref(d), w_pieces.data_ptr(), n_f32, y_pool.data_ptr(), pool_border, _stream()))
        return
    check(_lib.load().mp_conv_stem_xrec(C.byref(d), w_pieces.data_ptr(), n_f32, _stream()))


def conv2d_plan(N: int, H: int, W: int, Cp: int, in_border: int, Cout: int, K: int, stride: int, pad: int, n_cu: int,
                ws_floats: int = 0, x_f16: bool = False) -> Dict[str, int]:
    """How mp_conv2d_nhwc would lay this launch out on `n_cu` CUs (host-only, no GPU work): mode 0 single pass, 1 every tile
    split along K, 2 full rounds + split-K tail (half-precision inputs always run single pass)."""
    d = ConvDesc()
    d.x_f16 = int(x_f16)
    dummy = 0x1000  # the planner only tests pointers for NULL
    d.d_x, d.N, d.H, d.W, d.C, d.in_border = dummy, N, H, W, Cp, in_border
    d.d_w, d.Cout, d.KH, d.KW, d.stride, d.pad = dummy, Cout, K, K, stride, pad
    d.d_y, d.out_border = dummy, 1
    if ws_floats:
        d.d_splitk_ws, d.splitk_ws_floats = dummy, ws_floats
    out = (C.c_int32 * 5)()
    check(_lib.load().mp_conv2d_plan(C.byref(d), n_cu, out))
    return dict(zip(("mode", "k_split", "chunks_per_split", "n_main", "m_begin"), out))


def maxpool3x3s2(x, N, H, W, Cc, in_border, y, out_border, y_act=None, sc=None, sh=None) -> None:
    check(_lib.load().mp_maxpool3x3s2(x.data_ptr(), N, H, W, Cc, in_border, _ptr(y), out_border, _ptr(y_act), _ptr(sc), _ptr(sh),
                                      _stream()))


def bn_relu_nhwc(x, N, H, W, Cc, border, y_act, sc, sh) -> None:
    """y_act = relu(x * sc[c] + sh[c]) on the interior of a padded NHWC map (mp_bn_relu_nhwc: the WideResNets' first pre-activation
    behind the stem's fused max pool)"""
    check(_lib.load().mp_bn_relu_nhwc(x.data_ptr(), N, H, W, Cc, border, y_act.data_ptr(), sc.data_ptr(), sh.data_ptr(), _stream()))


def pool_fc_heads(x, N, H, W, Cc, in_border, fc_w, fc_b, n_feat, head_w, head_b, n_out, feat, out, sigmoid) -> None:
    check(_lib.load().mp_pool_fc_heads(x.data_ptr(), N, H, W, Cc, in_border, _ptr(fc_w), _ptr(fc_b), n_feat, head_w.data_ptr(),
                                       head_b.data_ptr(), n_out, _ptr(feat), out.data_ptr(), _ptr(sigmoid), _stream()))


class Backbone:
    """mp_backbone: whole CNN + head resident on the device, one call per forward."""

    def __init__(self, kind: str, c_in: int, head: str, n_out: int, state_dict: Dict[str, torch.Tensor]):
        lib = _lib.load()
        width = 1
        if kind.startswith("resnet34_width="):   # training/pose_models_cfg.py:114-116: WideResNet34(width=int(...))
            width, kind = int(kind.split("resnet34_width=")[1]), "resnet34"
        if kind not in BACKBONE_KINDS:
            raise EngineError(f"unknown backbone '{kind}' (pose_models_cfg.py:106-118 supports {list(BACKBONE_KINDS)} and resnet34_width=N)")
        self.width = width
        keep = []
        items = []
        for k, v in state_dict.items():
            if not torch.is_tensor(v) or not v.dtype.is_floating_point:
                continue
            a = np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32)
            keep.append(a)
            items.append((k.encode(), a))
        arr = (NamedTensor * len(items))()
        for i, (k, a) in enumerate(items):
            arr[i] = NamedTensor(k, a.ctypes.data, a.size)
        h = C.c_void_p()
        check(lib.mp_backbone_create_wide(BACKBONE_KINDS[kind], width, c_in, 0 if head == "pose" else 1, n_out, arr, len(items), C.byref(h)))
        self.handle = h
        self.kind, self.c_in, self.n_out = kind, c_in, n_out
        self.c_in_p = lib.mp_backbone_input_channels_padded(h)
        self.in_border = lib.mp_backbone_input_border(h)
        self._ws: Dict[int, torch.Tensor] = {}
        self._xrec_len: Dict[int, int] = {}   # fp32-kind channel mask -> record length of the prepared stem blob (0: no exact-piece form)

    def workspace(self, batch: int, h: int, w: int, device, slot: int = 0) -> torch.Tensor:
        need = _lib.load().mp_backbone_workspace_bytes(self.handle, batch, h, w)
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need or ws.device != torch.device(device):
            self._ws.pop(slot, None)
            self._ws[slot] = ws = torch.empty(need, dtype=torch.uint8, device=device)
            # a NEW allocation: its borders are not zero even if the executor has seen this address before
            check(_lib.load().mp_backbone_workspace_reset(self.handle, ws.data_ptr()))
        return ws

    def flops(self, batch: int, h: int, w: int) -> float:
        return _lib.load().mp_backbone_flops(self.handle, batch, h, w)

    def xrec_elements(self, n_f32: int = 3, f32_mask: Optional[int] = None) -> int:
        """Record length (bf16 elements per pixel) of the exact-piece stem input whose fp32-kind channels are the first `n_f32` (the
        observation crop) or the set bits of `f32_mask` (crop + depth channels of an RGBD model), all other input channels being 8-bit
        integers (renders); 0 = this stem has no such form.  This is also the PREPARE step (mp_backbone_xrec_prepare): the first call for
        a mask packs and uploads the stem's piece blob -- host work and a synchronous copy, so call it outside stream capture; `forward`
        only looks the blob up."""
        mask = leading_mask(n_f32) if f32_mask is None else int(f32_mask)
        if mask >> 32 or (self.c_in < 32 and mask >> self.c_in):
            return 0
        hit = self._xrec_len.get(mask)
        if hit is None:
            hit = self._xrec_len[mask] = int(_lib.load().mp_backbone_xrec_prepare(self.handle, mask))
        return hit

    def forward(self, x: torch.Tensor, batch: int, h: int, w: int, out: torch.Tensor, sigmoid: Optional[torch.Tensor] = None,
                feat: Optional[torch.Tensor] = None, slot: int = 0, n_f32: int = 3, f32_mask: Optional[int] = None,
                tile_flags: int = 0) -> None:
        """x: fp32 padded NHWC | float16 (same geometry, mp_backbone_forward_f16) | bfloat16 stem records whose fp32-kind channels are
        the first `n_f32` / the bits of `f32_mask` (what the rasteriser writes with MP_RASTER_XREC; mp_backbone_forward_xrec_mask).
        `tile_flags` = device address of the job flags of the raster launch that wrote the records (`raster_job_flags`): the stem takes
        the background-tile walk where it applies (mp_backbone_forward_xrec_sparse)."""
        ws = self.workspace(batch, h, w, x.device, slot)
        assert x.dtype in (torch.float32, torch.float16, torch.bfloat16)
        lib = _lib.load()
        if x.dtype == torch.bfloat16:
            mask = leading_mask(n_f32) if f32_mask is None else int(f32_mask)
            self.xrec_elements(f32_mask=mask)   # (prepared on first use; the C forward never allocates)
            if tile_flags:
                check(lib.mp_backbone_forward_xrec_sparse(self.handle, x.data_ptr(), mask & 0xFFFFFFFF, tile_flags, batch, h, w, out.data_ptr(),
                                                          _ptr(sigmoid), _ptr(feat), ws.data_ptr(), ws.numel(), _stream()))
                return
            check(lib.mp_backbone_forward_xrec_mask(self.handle, x.data_ptr(), mask & 0xFFFFFFFF, batch, h, w, out.data_ptr(), _ptr(sigmoid),
                                                    _ptr(feat), ws.data_ptr(), ws.numel(), _stream()))
            return
        fn = lib.mp_backbone_forward_f16 if x.dtype == torch.float16 else lib.mp_backbone_forward
        check(fn(self.handle, x.data_ptr(), batch, h, w, out.data_ptr(), _ptr(sigmoid), _ptr(feat), ws.data_ptr(), ws.numel(), _stream()))

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().mp_backbone_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------- #
def normalize_T(T: torch.Tensor) -> torch.Tensor:
    T = _dev_f32(T)
    out = torch.empty_like(T)
    check(_lib.load().mp_normalize_T(T.data_ptr(), T.shape[0], out.data_ptr(), _stream()))
    return out


def init_extents(points: torch.Tensor, R: torch.Tensor) -> torch.Tensor:
    points, R = _dev_f32(points), _dev_f32(R)
    n_mesh, n_pts, _ = points.shape
    ext = torch.empty(n_mesh, R.shape[0], 2, dtype=torch.float32, device=points.device)
    check(_lib.load().mp_init_extents(points.data_ptr(), n_mesh, n_pts, R.data_ptr(), R.shape[0], ext.data_ptr(), _stream()))
    return ext


def init_poses_from_boxes(boxes, K, mesh_ids, rot_ids, R, ext) -> torch.Tensor:
    boxes, K, R, ext = _dev_f32(boxes), _dev_f32(K), _dev_f32(R), _dev_f32(ext)
    b = boxes.shape[0]
    TCO = torch.empty(b, 4, 4, dtype=torch.float32, device=boxes.device)
    check(_lib.load().mp_init_poses_from_boxes(boxes.data_ptr(), K.data_ptr(), _dev_i32(mesh_ids).data_ptr(),
                                               _dev_i32(rot_ids).data_ptr(), R.data_ptr(), R.shape[0], ext.data_ptr(), b,
                                               TCO.data_ptr(), _stream()))
    return TCO


MV_MODES = {"TCO": 0, "1view_TCO": 0, "TCO+front_3views": 1, "TCO+front_1view": 2, "sphere_26views": 3}
MV_REMOVE_TCO, MV_INPLANE = 256, 512


def multiview_n_views(multiview: int) -> int:
    return int(_lib.load().mp_pose_multiview_n_views(int(multiview)))


def pose_prepare(TCO_in, K, mesh_ids, points, n_pts_main: int, n_pts_views: int, V: int, multiview: int, im_hw, out_hw,
                 lamb: float = 1.4, with_K_main: bool = False):
    """-> (TCO_n, tCR, TCV_O [b,V,4,4], KV_crop [b,V,3,3], boxes_rend, boxes_crop[, K_main [b,3,3]]); `multiview` = MV_MODES code
    | MV_REMOVE_TCO | MV_INPLANE (mp_pose_prepare_ex)."""
    TCO_in, K, points = _dev_f32(TCO_in), _dev_f32(K), _dev_f32(points)
    b = TCO_in.shape[0]
    dev = TCO_in.device
    f = dict(dtype=torch.float32, device=dev)
    TCO_n = torch.empty(b, 4, 4, **f)
    tCR = torch.empty(b, 3, **f)
    TCV_O = torch.empty(b, V, 4, 4, **f)
    KV = torch.empty(b, V, 3, 3, **f)
    boxes_rend = torch.empty(b, 4, **f)
    boxes_crop = torch.empty(b, 4, **f)
    K_main = torch.empty(b, 3, 3, **f) if with_K_main else None
    check(_lib.load().mp_pose_prepare_ex(TCO_in.data_ptr(), K.data_ptr(), _dev_i32(mesh_ids).data_ptr(), points.data_ptr(),
                                         points.shape[1], n_pts_main, n_pts_views, b, V, multiview, im_hw[0], im_hw[1], out_hw[0],
                                         out_hw[1], lamb, TCO_n.data_ptr(), tCR.data_ptr(), TCV_O.data_ptr(), KV.data_ptr(),
                                         boxes_rend.data_ptr(), boxes_crop.data_ptr(), _ptr(K_main), _stream()))
    if with_K_main:
        return TCO_n, tCR, TCV_O, KV, boxes_rend, boxes_crop, K_main
    return TCO_n, tCR, TCV_O, KV, boxes_rend, boxes_crop


def pose_update(TCO, K_crop, out9, tCR, k_stride_floats: int = 9) -> torch.Tensor:
    TCO, out9, tCR = _dev_f32(TCO), _dev_f32(out9), _dev_f32(tCR)
    assert K_crop.dtype == torch.float32 and K_crop.is_cuda
    out = torch.empty_like(TCO)
    check(_lib.load().mp_pose_update(TCO.data_ptr(), K_crop.data_ptr(), k_stride_floats, out9.data_ptr(), tCR.data_ptr(),
                                     TCO.shape[0], out.data_ptr(), _stream()))
    return out


def icp_refine(depth_meas: torch.Tensor, im_ids: torch.Tensor, depth_rend: torch.Tensor, K_images: torch.Tensor, K_rows: torch.Tensor,
               TCO: torch.Tensor, n_iterations: int = 100, n_levels: int = 4, tolerance: float = 0.05, n_min_points: int = 1000,
               user_masks: bool = False, association: str = "nn", return_iters: bool = False, masks: Optional[torch.Tensor] = None):
    """-> (TCO_refined [N,4,4], retval [N] int32 (0 ok / -1 input pose kept), residual [N]).  `user_masks`: the caller's masks
    are already applied to depth_meas; the 0.1 m measured-vs-rendered threshold mask is then not used (icp_refiner.py:249-250).
    association: "nn" = the reference's algorithm step for step (mp_icp_refine_nn: get_normal + OpenCV-style nearest-neighbour ICP);
    "projective" = the faster projective-association point-to-plane ICP (mp_icp_refine).  `masks` ("nn" only): the caller's per-frame
    masks [n_images,H,W], passed separately (depth_meas stays unmasked: the reference takes its normals from the whole frame)."""
    lib = _lib.load()
    depth_meas, depth_rend = _dev_f32(depth_meas), _dev_f32(depth_rend)
    K_images, K_rows, TCO = _dev_f32(K_images), _dev_f32(K_rows), _dev_f32(TCO)
    n_im, H, W = depth_meas.shape
    N = TCO.shape[0]
    assert depth_rend.shape == (N, H, W)
    dev = TCO.device
    out = torch.empty_like(TCO)
    retval = torch.empty(N, dtype=torch.int32, device=dev)
    residual = torch.empty(N, dtype=torch.float32, device=dev)
    if association == "nn":
        if user_masks:
            raise ValueError('association="nn" takes the caller\'s masks through `masks`, not pre-multiplied into the depth')
        if masks is not None:
            masks = (masks.to(dev) > 0).to(torch.uint8).contiguous()
            assert masks.shape == depth_meas.shape
        ws = torch.empty(lib.mp_icp_nn_workspace_bytes(n_im, N, H, W), dtype=torch.uint8, device=dev)
        iters = torch.zeros(N, 8, dtype=torch.int32, device=dev) if return_iters else None
        check(lib.mp_icp_refine_nn(depth_meas.data_ptr(), n_im, _dev_i32(im_ids).data_ptr(), depth_rend.data_ptr(), K_images.data_ptr(),
                                   K_rows.data_ptr(), TCO.data_ptr(), N, H, W, n_iterations, n_levels, tolerance, n_min_points, _ptr(masks),
                                   out.data_ptr(), retval.data_ptr(), residual.data_ptr(), _ptr(iters), ws.data_ptr(), ws.numel(), _stream()))
        return (out, retval, residual, iters) if return_iters else (out, retval, residual)
    if association != "projective":
        raise ValueError(f"association must be 'nn' or 'projective', got {association!r}")
    if masks is not None:
        raise ValueError('association="projective" takes masks pre-multiplied into depth_meas (user_masks=True)')
    ws = torch.empty(lib.mp_icp_workspace_bytes(n_im, N, H, W), dtype=torch.uint8, device=dev)
    check(lib.mp_icp_refine(depth_meas.data_ptr(), n_im, _dev_i32(im_ids).data_ptr(), depth_rend.data_ptr(), K_images.data_ptr(),
                            K_rows.data_ptr(), TCO.data_ptr(), N, H, W, n_iterations, n_levels, tolerance, n_min_points, int(user_masks), out.data_ptr(),
                            retval.data_ptr(), residual.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out, retval, residual


# --------------------------------------------------------------------------- #
class DetectorNet:
    """mp_detector: the Mask R-CNN (ResNet-50 + FPN) detection graph resident on the device, one call per image batch
    (csrc/detector.hip).  `state_dict` uses torchvision's keys (= a checkpoint of the reference's DetectorMaskRCNN)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], n_classes: int, min_size: int, max_size: int, **overrides):
        lib = _lib.load()
        self.cfg = _lib.DetectorConfig()
        check(lib.mp_detector_default_config(C.byref(self.cfg), n_classes, min_size, max_size))
        for k, v in overrides.items():
            if not hasattr(self.cfg, k):
                raise EngineError(f"unknown detector option '{k}'")
            cur = getattr(self.cfg, k)
            if hasattr(cur, "__len__"):
                cur[:] = list(v)
            else:
                setattr(self.cfg, k, v)
        keep, items = [], []
        for k, v in state_dict.items():
            if not torch.is_tensor(v) or not v.dtype.is_floating_point:
                continue
            a = np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32)
            keep.append(a)
            items.append((k.encode(), a))
        arr = (NamedTensor * len(items))()
        for i, (k, a) in enumerate(items):
            arr[i] = NamedTensor(k, a.ctypes.data, a.size)
        h = C.c_void_p()
        check(lib.mp_detector_create(C.byref(self.cfg), arr, len(items), C.byref(h)))
        self.handle = h
        self.n_classes = n_classes
        self._ws: Optional[torch.Tensor] = None

    @staticmethod
    def state_spec(n_classes: int) -> List[Tuple[str, Tuple[int, ...]]]:
        """[(state_dict key, shape)] the detector expects (host only: mp_detector_state_spec)"""
        lib = _lib.load()
        out, i = [], 0
        while True:
            name = C.create_string_buffer(160)
            shp, nd = (C.c_int64 * 4)(), C.c_int32(0)
            rc = lib.mp_detector_state_spec(n_classes, i, name, 160, shp, C.byref(nd))
            if rc == 1:
                return out
            check(rc)
            out.append((name.value.decode(), tuple(int(s) for s in shp[: nd.value])))
            i += 1

    def forward(self, images: torch.Tensor, with_masks: bool = True):
        """images [n,3,H,W] fp32 in [0,1] on the GPU -> (boxes [n,D,4], scores [n,D], labels [n,D] int32, counts [n] int32,
        masks [n,D,H,W] or None); nothing synchronises, entries past counts[i] are zero."""
        lib = _lib.load()
        images = _dev_f32(images)
        n, c, H, W = images.shape
        if c != 3:
            raise EngineError("the detector takes RGB images [n,3,H,W]")
        dev = images.device
        D = int(self.cfg.box_detections_per_img)
        need = lib.mp_detector_workspace_bytes(self.handle, n, H, W)
        if need == 0:
            raise EngineError(f"mp_detector_workspace_bytes failed for {n} x {H} x {W}: {lib.mp_last_error().decode()}")
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        boxes = torch.empty(n, D, 4, dtype=torch.float32, device=dev)
        scores = torch.empty(n, D, dtype=torch.float32, device=dev)
        labels = torch.empty(n, D, dtype=torch.int32, device=dev)
        counts = torch.empty(n, dtype=torch.int32, device=dev)
        masks = torch.empty(n, D, H, W, dtype=torch.float32, device=dev) if with_masks else None
        check(lib.mp_detector_forward(self.handle, images.data_ptr(), n, H, W, boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(),
                                      counts.data_ptr(), _ptr(masks), self._ws.data_ptr(), self._ws.numel(), _stream()))
        return boxes, scores, labels, counts, masks

    def debug_tensor(self, what: str) -> torch.Tensor:
        """a COPY of an intermediate of the last forward (parity tests): padded maps come back as their [n,h,w,c] interior"""
        lib = _lib.load()
        ptr, shp, border, rs, n_el = C.c_void_p(), (C.c_int64 * 4)(), C.c_int32(0), C.c_int64(0), C.c_int64(0)
        check(lib.mp_detector_debug_tensor(self.handle, what.encode(), C.byref(ptr), shp, C.byref(border), C.byref(rs), C.byref(n_el)))
        off = ptr.value - self._ws.data_ptr()
        is_int = what in ("proposal_counts", "f_cnt")
        flat = self._ws[off : off + 4 * n_el.value].view(torch.int32 if is_int else torch.float32).clone()
        s, b = [int(v) for v in shp], border.value
        if b:
            return flat.view(s[0], s[1] + 2 * b, s[2] + 2 * b, s[3])[:, b : b + s[1], b : b + s[2]].contiguous()
        if s[2] == 4:
            return flat.view(s[0], s[1], 4)
        if rs.value > 1:
            return flat.view(s[0], rs.value)[:, : s[1]].contiguous()
        return flat.view(s[0], s[1])

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().mp_detector_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
