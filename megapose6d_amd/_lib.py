"""ctypes binding of libmp_engine.so (the C-ABI in include/mp_engine.h).

The product path has NO fallback: if the shared library is missing or a call fails
this module raises.  Build with ``python __graft_entry__.py`` (or ``make -C megapose6d_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libmp_engine.so"


class EngineError(RuntimeError):
    pass


class MeshDesc(C.Structure):
    _fields_ = [
        ("h_vertices", C.c_void_p),
        ("h_normals", C.c_void_p),
        ("h_colors", C.c_void_p),
        ("h_faces", C.c_void_p),
        ("n_vertices", C.c_int32),
        ("n_faces", C.c_int32),
    ]


class Lights(C.Structure):
    _fields_ = [
        ("ambient", C.c_float * 3),
        ("n_point", C.c_int32),
        ("point_dir", (C.c_float * 3) * 8),
        ("point_color", (C.c_float * 3) * 8),
        ("point_offset", (C.c_float * 3) * 8),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("d_x", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
        ("in_border", C.c_int32),
        ("d_w", C.c_void_p),
        ("d_bias", C.c_void_p),
        ("Cout", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("d_y", C.c_void_p),
        ("out_border", C.c_int32),
        ("d_residual", C.c_void_p),
        ("relu", C.c_int32),
        ("d_y_act", C.c_void_p),
        ("d_act_scale", C.c_void_p),
        ("d_act_shift", C.c_void_p),
        ("c_real", C.c_int32),
        ("d_splitk_ws", C.c_void_p),
        ("splitk_ws_floats", C.c_int64),
        ("x_f16", C.c_int32),
    ]


class DetectorConfig(C.Structure):
    _fields_ = [
        ("n_classes", C.c_int32),
        ("min_size", C.c_int32), ("max_size", C.c_int32),
        ("image_mean", C.c_float * 3), ("image_std", C.c_float * 3),
        ("anchor_sizes", C.c_int32 * 5),
        ("aspect_ratios", C.c_float * 3),
        ("rpn_pre_nms_top_n", C.c_int32), ("rpn_post_nms_top_n", C.c_int32),
        ("rpn_nms_thresh", C.c_float), ("rpn_score_thresh", C.c_float), ("rpn_min_size", C.c_float),
        ("box_score_thresh", C.c_float), ("box_nms_thresh", C.c_float), ("box_min_size", C.c_float),
        ("box_detections_per_img", C.c_int32),
    ]


class NamedTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("h_data", C.c_void_p), ("numel", C.c_int64)]


_vp, _i, _i64, _f, _sz, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_uint32

# name -> (restype, argtypes); exactly the symbols declared in include/mp_engine.h
SIGNATURES = {
    "mp_version": (_i, []),
    "mp_last_error": (C.c_char_p, []),
    "mp_conv_clock_read": (_i, [C.POINTER(C.c_double), _i]),
    "mp_clock_probe": (_i, [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), _vp]),
    "mp_profile_begin": (_i, []),
    "mp_profile_end": (_i, []),
    "mp_profile_active": (_i, []),
    "mp_profile_query": (_i, [_i, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mp_profile_query_ex": (_i, [_i, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mp_device_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "mp_mesh_db_create": (_i, [C.POINTER(MeshDesc), _i, C.POINTER(_vp)]),
    "mp_mesh_db_set_texture": (_i, [_vp, _i, _vp, _vp, _i, _i, _i]),
    "mp_mesh_db_destroy": (_i, [_vp]),
    "mp_mesh_db_max_vertices": (_i, [_vp]),
    "mp_mesh_db_radius": (_f, [_vp, _i]),
    "mp_raster_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "mp_raster_render": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _u32, C.POINTER(Lights), _vp, _i64, _i, _i64, _i64, _i64, _i, _i, _i,
                              _vp, _sz, _vp]),
    "mp_raster_render_crop": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _u32, C.POINTER(Lights), _vp, _i64, _i, _i64, _i64, _i64, _i, _i, _i,
                                   _vp, _sz, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mp_raster_render_xrec": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _u32, C.POINTER(Lights), _vp, _i64, _i, _i64, _i64, _i64, _i, _i, _i,
                                   _vp, _sz, _vp, _i, _i, _i, _i, _i, _vp, _vp, _u32, _vp, _i, _vp]),
    "mp_raster_job_flags": (_vp, [_vp, _vp, _i, _i, _i]),
    "mp_pack_observation_nhwc4": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "mp_crop_roi_align": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i64, _i64, _i64, _i, _vp]),
    "mp_normalize_depth": (_i, [_vp, _i, _i, _i, _i, _i, C.POINTER(C.c_int32), _i, _vp, _i, _vp]),
    "mp_normalize_depth_f16": (_i, [_vp, _i, _i, _i, _i, _i, C.POINTER(C.c_int32), _i, _vp, _i, _vp]),
    "mp_conv_packed_floats": (_sz, [_i, _i, _i, _i]),
    "mp_conv_pack_weights": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mp_conv2d_nhwc": (_i, [C.POINTER(ConvDesc), _vp]),
    "mp_conv2d_plan": (_i, [C.POINTER(ConvDesc), _i, C.POINTER(C.c_int32)]),
    "mp_conv2d_kernel_name": (C.c_char_p, [C.POINTER(ConvDesc)]),
    "mp_conv_wino_packed_floats": (_sz, [_i, _i]),
    "mp_conv_wino_pack_weights": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mp_conv_wino_eligible": (_i, [C.POINTER(ConvDesc), _i]),
    "mp_conv3x3_wino_nhwc": (_i, [C.POINTER(ConvDesc), _vp, _vp]),
    "mp_conv_wino_stats": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), _i]),
    "mp_conv_wino_bf16_packed_bytes": (_sz, [_i, _i]),
    "mp_conv_wino_bf16_pack_weights": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mp_conv3x3_wino_bf16_nhwc": (_i, [C.POINTER(ConvDesc), _vp, _vp]),
    "mp_conv_wino_bf16_stats": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), _i]),
    "mp_conv_wino_bf16_clock": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), _i]),
    "mp_conv_wino_bf16_phases": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mp_conv_wino_bf16_telemetry": (_i, [_i]),
    "mp_xrec_elements": (_i, [_i, _i]),
    "mp_conv_stem_supported": (_i, [_i, _i, _i]),
    "mp_conv_stem_packed_bytes": (_sz, [_i, _i, _i, _i]),
    "mp_conv_stem_pack_weights": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "mp_conv_stem_pack_weights_mask": (_i, [_vp, _i, _i, _i, _u32, _vp, _vp]),
    "mp_conv_stem_sparse_chunks": (_i, [_i, _i, _i]),
    "mp_conv_stem_sparse_packed_bytes": (_sz, [_i, _i, _i, _i]),
    "mp_conv_stem_pack_weights_sparse": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "mp_conv_stem_xrec_sparse": (_i, [C.POINTER(ConvDesc), _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "mp_conv_stem_bg_stats": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), _i]),
    "mp_conv_stem_xrec": (_i, [C.POINTER(ConvDesc), _vp, _i, _vp]),
    "mp_conv_stem_xrec_pool": (_i, [C.POINTER(ConvDesc), _vp, _i, _vp, _i, _vp]),
    "mp_maxpool3x3s2": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "mp_bn_relu_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mp_pool_fc_heads": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "mp_backbone_create": (_i, [_i, _i, _i, _i, C.POINTER(NamedTensor), _i, C.POINTER(_vp)]),
    "mp_backbone_create_wide": (_i, [_i, _i, _i, _i, _i, C.POINTER(NamedTensor), _i, C.POINTER(_vp)]),
    "mp_backbone_destroy": (_i, [_vp]),
    "mp_backbone_input_channels_padded": (_i, [_vp]),
    "mp_backbone_input_border": (_i, [_vp]),
    "mp_backbone_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "mp_backbone_workspace_reset": (_i, [_vp, _vp]),
    "mp_backbone_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mp_backbone_forward_f16": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mp_backbone_xrec_elements": (_i, [_vp, _i]),
    "mp_backbone_forward_xrec": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mp_backbone_xrec_prepare": (_i, [_vp, _u32]),
    "mp_backbone_forward_xrec_mask": (_i, [_vp, _vp, _u32, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mp_backbone_forward_xrec_sparse": (_i, [_vp, _vp, _u32, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mp_backbone_flops": (C.c_double, [_vp, _i, _i, _i]),
    "mp_normalize_T": (_i, [_vp, _i, _vp, _vp]),
    "mp_init_extents": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    "mp_init_poses_from_boxes": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "mp_pose_prepare": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp,
                             _vp]),
    "mp_pose_multiview_n_views": (_i, [_i]),
    "mp_pose_prepare_ex": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_icp_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mp_icp_refine": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mp_icp_nn_max_points": (_i, []),
    "mp_icp_nn_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mp_icp_refine_nn": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mp_pose_update": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "mp_detector_default_config": (_i, [C.POINTER(DetectorConfig), _i, _i, _i]),
    "mp_detector_state_spec": (_i, [_i, _i, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(C.c_int32)]),
    "mp_detector_create": (_i, [C.POINTER(DetectorConfig), C.POINTER(NamedTensor), _i, C.POINTER(_vp)]),
    "mp_detector_destroy": (_i, [_vp]),
    "mp_detector_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "mp_detector_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mp_detector_debug_tensor": (_i, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(C.c_int32), C.POINTER(_i64), C.POINTER(_i64)]),
}

_lib = None


def load() -> C.CDLL:
    """Load libmp_engine.so and bind every symbol; raises EngineError if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("MP_ENGINE_LIB", LIB_PATH))
    if not path.is_file():
        raise EngineError(
            f"{path} not found: the HIP engine is not built. Run `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback."
        )
    try:
        lib = C.CDLL(str(path))
    except OSError as e:  # e.g. libamdhip64 missing
        raise EngineError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EngineError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().mp_last_error().decode("utf-8", "replace")
        raise EngineError(f"mp_engine error {rc}: {msg}")
