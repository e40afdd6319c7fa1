/* mp_engine.h -- C-ABI of the MI355X-native render-and-compare pose engine.
 *
 * This is the drop-in boundary for the reference's hot path.
 * The reference (megapose6d) is pure Python and has no FFI layer; the seam is three
 * duck-typed Python objects.  Each entry point below names the reference interface it
 * replaces (paths relative to /root/reference/src/megapose/).  INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *  - plain C, no torch types; every pointer named d_* is DEVICE memory (HBM), h_* is HOST.
 *  - mp_stream is a hipStream_t; all work is enqueued on it, nothing synchronises.
 *  - return value: 0 = ok, negative = error (message via mp_last_error()).
 *  - handles are opaque, thread-compatible (one thread per handle at a time).
 *  - outputs are caller-allocated.
 *  - images/activations inside the engine are fp32 NHWC with a zero border ("padded NHWC"):
 *    element (n, y, x, c) of a tensor with logical size H x W, border B, channels C lives at
 *    ((n*(H+2B) + y+B)*(W+2B) + x+B)*C + c.  Borders are zero and never written.
 */
#ifndef MP_ENGINE_H
#define MP_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mp_stream; /* hipStream_t */

#define MP_OK 0
#define MP_ERR_INVALID (-22)
#define MP_ERR_NOMEM (-12)
#define MP_ERR_HIP (-5)

int mp_version(void);
const char* mp_last_error(void);
/* Per-launch HIP-event profiler: between begin/end every instrumented kernel launch is bracketed by hipEvents recorded on
 * its launch stream.  mp_profile_query aggregates by kernel name: number of launches, summed duration (ms), and the summed
 * ALGORITHMIC flops / bytes of those launches (DESIGN.md states each kernel's per-unit figures).  Returns 1 past the end. */
int mp_profile_begin(void);
int mp_profile_end(void);
int mp_profile_active(void); /* 1 between begin and end */
int mp_profile_query(int idx, char* name, int name_len, int64_t* launches, double* total_ms, double* total_flops,
                     double* total_bytes);
/* the same + the FLOPs the launches EXECUTED on the matrix pipe (= the algorithmic figure for the direct kernels; 16/36 of it for the
 * fp32 Winograd kernel; 9 bf16 piece products per Winograd multiplication / 3 or 9 per stem multiplication for the exact-piece kernels)
 * and the dense peak (TFLOP/s) of the pipe they run on (157.3 fp32 MFMA, 2500 bf16 MFMA): executed / time / peak = MFMA utilisation */
int mp_profile_query_ex(int idx, char* name, int name_len, int64_t* launches, double* total_ms, double* total_flops,
                        double* total_bytes, double* total_executed_flops, double* peak_tflops);



/* number of CUs etc. of the current device; fails loudly when no gfx950 device is usable */
int mp_device_info(int* n_cus, int* lds_bytes, char* arch_name, int arch_name_len);

/* ------------------------------------------------------------------------------------ */
/* Mesh database: replaces the per-worker Panda3D model cache                            */
/* (panda3d_renderer/panda3d_scene_renderer.py:192-207 get_object_node) and the          */
/* BatchedMeshes point tensors (lib3d/rigid_mesh_database.py:90-130).                     */
/* ------------------------------------------------------------------------------------ */
typedef struct mp_mesh_db mp_mesh_db;

typedef struct {
  const float* h_vertices; /* [n_vertices,3] metres (RigidObject.scale and ypr offset applied) */
  const float* h_normals;  /* [n_vertices,3] unit vertex normals (object frame)                */
  const float* h_colors;   /* [n_vertices,3] albedo in [0,1] (uint8/255)                       */
  const int32_t* h_faces;  /* [n_faces,3]                                                      */
  int32_t n_vertices;
  int32_t n_faces;
} mp_mesh_desc;

int mp_mesh_db_create(const mp_mesh_desc* h_meshes, int n_meshes, mp_mesh_db** out);
/* UV texture of mesh `mesh_id` (replaces Panda3D's assimp/texture loading, panda3d_scene_renderer.py:192-207: the albedo of a
 * textured RigidObject): h_uvs = per-corner (u,v) [n_faces][3][2] float32 in the corner order of h_faces (v = 0 at the FIRST
 * texel row passed here, i.e. the host flips image rows so that v grows with the row index); h_texels = RGBA8 mip chain, level l
 * of size max(1,w>>l) x max(1,h>>l), levels concatenated (built on the host: megapose6d_amd.mesh_io.build_mip_chain).
 * Sampling: repeat wrap, trilinear (bilinear taps in the two mip levels around a per-pixel level of detail derived from the
 * analytic screen-space uv derivatives; texture-minfilter mipmap, panda3d_scene_renderer.py:71); albedo = vertex colour x
 * texel / 255.  16x anisotropic filtering (:72) is not reproduced. */
#define MP_TEX_MAX_LEVELS 15
int mp_mesh_db_set_texture(mp_mesh_db* db, int mesh_id, const float* h_uvs, const uint32_t* h_texels, int tex_w, int tex_h,
                           int n_levels);
int mp_mesh_db_destroy(mp_mesh_db* db);
int mp_mesh_db_max_vertices(const mp_mesh_db* db);
/* bounding-sphere radius (AABB centre) of mesh i, used for the point-light placement
 * (panda3d_scene_renderer.py:121-125 pos_fn) */
float mp_mesh_db_radius(const mp_mesh_db* db, int mesh_id);

/* ------------------------------------------------------------------------------------ */
/* Rasteriser: replaces Panda3dBatchRenderer.render                                       */
/* (panda3d_renderer/panda3d_batch_renderer.py:217-282; worker_loop :89-150;              */
/*  Panda3dSceneRenderer.render_scene panda3d_scene_renderer.py:298-358).                 */
/* One object per view, pinhole K, near 0.1 m / far 10 m (types.py:63-64).                */
/* ------------------------------------------------------------------------------------ */
#define MP_RASTER_NORMALS 1u      /* render_normals=True: eye-normal LUT pass               */
#define MP_RASTER_DEPTH 2u        /* render_depth=True: metric z, 0 = background            */
#define MP_RASTER_NORMALS_GL 4u   /* eye space = GL (x right,y up,z back) instead of Panda  */
#define MP_RASTER_MSAA4 16u        /* 4x multisampling (the reference's configuration: framebuffer-multisample 1,
                                     multisamples 4, panda3d_scene_renderer.py:73-74): coverage and depth per sample of the
                                     standard 4-sample pattern, shading once per (pixel, piece) at the pixel centre, 8-bit
                                     per-sample colours averaged; without the flag: one sample at the pixel centre          */

#define MP_RASTER_F16 32u          /* "fp16 renders" (BASELINE.json configs[4]; the reference's output path, panda3d_batch_renderer.py:
                                     261-274, converts uint8 -> fp32): d_out points at IEEE binary16 elements instead of floats --
                                     same ELEMENT strides and channel numbers, every written channel (renders and, with
                                     mp_raster_render_crop, the observation crop) rounded to nearest-even.  The consumer is
                                     mp_backbone_forward_f16 / mp_conv_desc.x_f16.                                              */

#define MP_RASTER_XREC 64u         /* d_out points at the bf16 pixel RECORDS of the exact-piece stem convolution (mp_conv_stem_xrec):
                                     [x1,x2,x3 of every crop channel | the 8-bit integer k of every render channel (NOT divided by 255) |
                                     zero padding], mp_xrec_elements(C_crop, n_render_channels) elements per pixel.  mp_raster_render_crop
                                     (c0_crop = 0, no depth channel, one launch writes every channel of the record) or, with depth channels,
                                     mp_raster_render_xrec; stride_v /
                                     stride_y / stride_x count bf16 elements (stride_x = the record length), c_rgb / c_normals /
                                     stride_view stay logical channel numbers.  The values are the ones the fp32 output holds: k = the
                                     integer whose k / 255 the fp32 path stores, x1 + x2 + x3 = the fp32 crop value exactly.            */

typedef struct {
  float ambient[3];        /* sum of ambient light colours                                  */
  int32_t n_point;         /* number of point lights (<= 8)                                 */
  float point_dir[8][3];   /* light position (object frame) = dir * 10 * mesh radius + offset: the affine-in-the-radius  */
  float point_color[8][3]; /* form of Panda3dLightData.positioning_function (panda3d_scene_renderer.py:104-136,         */
  float point_offset[8][3];/* types.py:104-114); make_scene_lights: dir = +-axes, offset = 0                            */
} mp_lights;

/* scratch for the per-view tile lists of a launch of n_views views at h x w (<= 1024 x 1024) */
size_t mp_raster_workspace_bytes(const mp_mesh_db* db, int n_views, int h, int w);

/* d_out addressing: element (view v, y, x, channel c) at
 *   d_out[(v / views_per_item)*stride_v + (v % views_per_item)*stride_view + y*stride_y + x*stride_x + c]
 * (views_per_item = 1, stride_view = 0 for a plain batch; = n_rendered_views / channels-per-view when the views of
 * one hypothesis are folded into the channels of one CNN input row, models/pose_rigid.py:405-408).
 * c_rgb / c_normals / c_depth are the first channel of each group (negative = not written).
 * Values are uint8-quantised then /255 exactly as panda3d_batch_renderer.py:261-274
 * (depth is not quantised).  Non-finite TCO/K rows produce zeros (:109-135).  Triangles crossing the near plane are
 * clipped.  One launch writes at most 32 consecutive channels per pixel (c_lo .. c_hi over all groups / views / the crop),
 * and stride_x must be >= that run.                                                        */
int mp_raster_render(const mp_mesh_db* db, const int32_t* d_mesh_ids, const float* d_TCO /*[n,4,4]*/,
                     const float* d_K /*[n,3,3]*/, int n_views, int h, int w, uint32_t flags,
                     const mp_lights* h_lights, float* d_out, int64_t stride_v, int views_per_item,
                     int64_t stride_view, int64_t stride_y, int64_t stride_x, int c_rgb, int c_normals,
                     int c_depth, void* d_workspace, size_t workspace_bytes, mp_stream stream);

/* mp_raster_render + the observation crop of every item (mp_crop_roi_align semantics, boxes / im_ids per ITEM = n_views /
 * views_per_item) written by the same launch into channels c0_crop.. of the item's pixels: what PosePredictor.forward does per
 * iteration with crop_inputs (models/pose_rigid.py:180-247) + render_images_multiview (:336-408) + torch.cat (:567).  The wave
 * that rasterises an 8x8-pixel tile of the item's views also computes the roi_align of those pixels, so every pixel of the CNN
 * input leaves the chip once, as one contiguous record of all its channels. */
int mp_raster_render_crop(const mp_mesh_db* db, const int32_t* d_mesh_ids, const float* d_TCO, const float* d_K, int n_views,
                          int h, int w, uint32_t flags, const mp_lights* lights, float* d_out, int64_t stride_v,
                          int views_per_item, int64_t stride_view, int64_t stride_y, int64_t stride_x, int c_rgb,
                          int c_normals, int c_depth, void* d_workspace, size_t workspace_bytes,
                          const float* d_images /*[n_im,C,H,W], or [n_im,H,W,4] if images_nhwc4*/, int images_nhwc4, int n_im,
                          int C, int H, int W, const int32_t* d_im_ids, const float* d_boxes, int c0_crop, mp_stream stream);
/* The record form of that launch for models WITH depth channels (the RGBD refiner, training/pose_models_cfg.py:101-103: observation depth
 * + one rendered depth per view; BASELINE.json configs[2]): MP_RASTER_XREC is implied, d_out = bf16 records of
 * mp_xrec_elements(popcount(f32_mask), n_channels - popcount(f32_mask)) elements.  Bit c of f32_mask marks logical channel c as fp32-kind
 * (three exact bf16 pieces): the crop's channels and every depth channel must be marked, the rgb / normal channels must not.  Depth
 * channels -- the 4th crop channel and c_depth of every view -- are normalised BEFORE the split exactly as mp_normalize_depth does it on the
 * fp32 tensor (models/pose_rigid.py:466-496; depth_mode 0..3, d_tCR [n_items,3]: the row's object-centre translation, z = reference depth;
 * background depth 0 is normalised like any other value), so the record decodes to the fp32 tensor path's values bit for bit. */
int mp_raster_render_xrec(const mp_mesh_db* db, const int32_t* d_mesh_ids, const float* d_TCO, const float* d_K, int n_views,
                          int h, int w, uint32_t flags, const mp_lights* lights, void* d_out_records, int64_t stride_v,
                          int views_per_item, int64_t stride_view, int64_t stride_y, int64_t stride_x, int c_rgb,
                          int c_normals, int c_depth, void* d_workspace, size_t workspace_bytes,
                          const float* d_images, int images_nhwc4, int n_im, int C, int H, int W, const int32_t* d_im_ids,
                          const float* d_boxes, uint32_t f32_mask, const float* d_tCR, int depth_mode, mp_stream stream);
/* The job flags of the LAST mp_raster_render* launch on `d_workspace`: NULL unless that launch ran in the compacted form (default;
 * MP_RASTER_COMPACT=0 = direct form) with exactly this n_views, h and w (the library keeps a host-side record per workspace, so stale
 * or foreign bytes are never handed out); else the device pointer to [n_items][tiles_y = ceil(h / 8)][tiles_x = ceil(w / 8)] bytes, 0 = no
 * view of the item reaches that 8x8-pixel tile (its render channels are all background).  Consumer: mp_conv_stem_xrec_sparse /
 * mp_backbone_forward_xrec_sparse on the same stream, before the next raster launch on this workspace. */
const unsigned char* mp_raster_job_flags(const mp_mesh_db* db, const void* d_workspace, int n_views, int h, int w);
/* observation frames [n_im,C,H,W] (C = 3 | 4) -> [n_im,H,W,4] (4th channel 0 for RGB): one 16-byte load per roi_align tap in the
 * fused crop; done once per observation, not per step */
int mp_pack_observation_nhwc4(const float* d_images, int n_im, int C, int H, int W, float* d_out, mp_stream stream);

/* ------------------------------------------------------------------------------------ */
/* Crop: replaces lib3d/cropping.py:113-144 crop_images (torchvision.ops.roi_align,       */
/* sampling_ratio=4, aligned=False) incl. the RGBD validity rule (:131-142), reading the  */
/* observation by batch_im_id (no per-row gather, pose_estimator.py:389).                 */
/* ------------------------------------------------------------------------------------ */
int mp_crop_roi_align(const float* d_images /*[n_im,C,H,W] NCHW*/, int n_im, int C, int H, int W,
                      const int32_t* d_im_ids /*[b]*/, const float* d_boxes /*[b,4] x1,y1,x2,y2*/, int b,
                      int out_h, int out_w, float* d_out, int64_t stride_b, int64_t stride_y,
                      int64_t stride_x, int c0, mp_stream stream);

/* ------------------------------------------------------------------------------------ */
/* Depth normalisation: models/pose_rigid.py:466-496 normalize_depth                      */
/* mode: 0 none, 1 tCR_scale, 2 tCR_scale_clamp_center, 3 tCR_center_clamp                */
/* applied in place to `n_ch` channels (list h_channels) of a padded-NHWC tensor.         */
/* ------------------------------------------------------------------------------------ */
int mp_normalize_depth(float* d_x, int b, int h, int w, int border, int C, const int32_t* h_channels,
                       int n_ch, const float* d_tCR /*[b,3]*/, int mode, mp_stream stream);
/* the same on a half-precision padded-NHWC tensor (MP_RASTER_F16 output): read, normalise in fp32, round back to binary16 */
int mp_normalize_depth_f16(void* d_x_half, int b, int h, int w, int border, int C, const int32_t* h_channels,
                           int n_ch, const float* d_tCR /*[b,3]*/, int mode, mp_stream stream);

/* ------------------------------------------------------------------------------------ */
/* Convolution stack: replaces `self.backbone(x)` (models/pose_rigid.py:323) for           */
/* models/torchvision_resnet.py (vanilla_resnet34) and models/wide_resnet.py               */
/* (WideResNet18/34).  fp32 in, fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 out.             */
/* ------------------------------------------------------------------------------------ */
/* number of floats of the packed weight blob of one conv (Cin_p = padded input channels) */
size_t mp_conv_packed_floats(int Cin_p, int Cout, int KH, int KW);
/* host-side packing of OIHW weights with per-output-channel scale folded in (eval BN).    */
int mp_conv_pack_weights(const float* h_w_oihw, int Cout, int Cin, int KH, int KW, int Cin_p,
                         const float* h_scale /*[Cout] or NULL*/, float* h_packed);

typedef struct {
  const float* d_x;        /* padded NHWC input                                             */
  int32_t N, H, W, C;      /* logical input size; C = padded channel count (multiple of 4)   */
  int32_t in_border;
  const float* d_w;        /* packed weights (mp_conv_pack_weights)                          */
  const float* d_bias;     /* [Cout] or NULL                                                 */
  int32_t Cout, KH, KW, stride, pad;
  float* d_y;              /* padded NHWC output (may be NULL when only d_y_act is wanted)   */
  int32_t out_border;
  const float* d_residual; /* same geometry as d_y, or NULL                                  */
  int32_t relu;            /* y = relu(conv + bias + residual)                               */
  float* d_y_act;          /* optional second output relu(y*act_scale + act_shift)           */
  const float* d_act_scale;
  const float* d_act_shift;
  int32_t c_real;          /* real (unpadded) input channels, for the profiler's algorithmic FLOP count; 0 = C   */
  float* d_splitk_ws;      /* optional scratch: with it, launches whose tile grid cannot fill the chip (small batches) split  */
  int64_t splitk_ws_floats;/* the K loop over several workgroups and reduce deterministically (fixed order); NULL = never    */
  int32_t x_f16;           /* != 0: d_x holds IEEE binary16 values in the same padded-NHWC geometry (what MP_RASTER_F16 writes);    */
                           /* the kernel widens them to fp32 on the way into LDS, arithmetic and outputs stay fp32.  Cout <= 64     */
                           /* (the stem convolutions)                                                                              */
} mp_conv_desc;

int mp_conv2d_nhwc(const mp_conv_desc* desc, mp_stream stream);
/* Host-side launch plan of mp_conv2d_nhwc for a device with n_cu compute units (no GPU work; pointers in `desc` are only tested
 * for NULL): out5 = {mode, k_split, chunks_per_split, n_main_tiles, first_row_of_the_split_part}; mode 0 = one single-pass launch,
 * 1 = small grid, every tile split along K, 2 = whole rounds single-pass + split-K for the tiles of a half-empty last round. */
int mp_conv2d_plan(const mp_conv_desc* desc, int n_cu, int32_t* out5);

/* the name of the kernel instantiation mp_conv2d_nhwc would launch (for profiling)        */
const char* mp_conv2d_kernel_name(const mp_conv_desc* desc);

/* Fused Winograd F(2x2, 3x3) form of the 3x3 / stride-1 / pad-1 convolutions of the residual stages (same call sites as
 * mp_conv2d_nhwc: models/torchvision_resnet.py:74-120 BasicBlock conv1 / conv2, models/wide_resnet.py:29-56) -- 16 instead of 36
 * multiplications per (2x2 output tile, cin, cout); fp32 MFMA, fp32 transforms; same fused epilogue (bias, residual, ReLU, second
 * pre-activated output).  d_u = the blob of mp_conv_wino_pack_weights uploaded to the device; `desc->d_w` and the split-K fields
 * are ignored.  Needs C % 16 == 0, Cout % 64 == 0, in_border >= 1.  When H or W is odd the kernel reads (and discards) up to one
 * padded row + one pixel past the end of the input tensor: the caller provides that much readable slack (the backbone workspace
 * does).  mp_conv_wino_eligible: 1 if a layer qualifies AND its grid gives each of n_cu compute units a workgroup (small grids stay
 * on mp_conv2d_nhwc's split-K path).  See csrc/conv_wino.hip. */
size_t mp_conv_wino_packed_floats(int Cin_p, int Cout);
int mp_conv_wino_pack_weights(const float* h_w_oi33, int Cout, int Cin, int Cin_p, const float* h_scale /*[Cout] or NULL*/, float* h_packed);
int mp_conv_wino_eligible(const mp_conv_desc* desc, int n_cu);
int mp_conv3x3_wino_nhwc(const mp_conv_desc* desc, const float* d_u, mp_stream stream);

/* The same fused Winograd convolution with its multiplications on the bf16 MFMA through EXACT operand pieces (csrc/conv_wino_bf16.hip):
 * U = G g G^T and every fp32 fragment of V = B^T d B are split by truncation into three bf16 pieces (24 = 3 x 8 mantissa bits) and ALL
 * nine piece products are accumulated in fp32 -- every product is exact, the result differs from mp_conv3x3_wino_nhwc only in the order
 * of the fp32 additions -- at 9/16 of the fp32-MFMA matrix time.  Same descriptor, eligibility (mp_conv_wino_eligible) and read-slack
 * contract; d_u_pieces = the blob of mp_conv_wino_bf16_pack_weights.  Launch form: persistent (one workgroup per CU walks the 64-tile x
 * 64-channel units; MP_WINO_PERSIST=0 = one workgroup per unit); results are identical.  (Counters / clock telemetry: mp_engine_debug.h.) */
size_t mp_conv_wino_bf16_packed_bytes(int Cin_p, int Cout);
int mp_conv_wino_bf16_pack_weights(const float* h_w_oi33, int Cout, int Cin, int Cin_p, const float* h_scale /*[Cout] or NULL*/, void* h_packed);
int mp_conv3x3_wino_bf16_nhwc(const mp_conv_desc* desc, const void* d_u_pieces, mp_stream stream);

/* Stem convolution on the bf16 MFMA through EXACT operand pieces (csrc/conv_stem.hip; same call site as mp_conv2d_nhwc for the first
 * layer: models/torchvision_resnet.py:213-216, models/wide_resnet.py:65-67).  The render channels of the CNN input are 8-bit integers
 * k / 255 by the reference's contract (uint8 -> float, panda3d_batch_renderer.py:261-274): k is ONE bf16 exactly; the weights (BN scale
 * and 1/255 folded in) and the fp32 observation-crop channels are split by truncation into three bf16 pieces each (24 = 3 x 8 mantissa
 * bits), so every bf16 x bf16 product is exact in the fp32 accumulator and the result differs from the fp32 convolution in the order of
 * the fp32 additions and in one rounding per integer-channel weight (the folded 1/255) -- at 3/16 (9/16 for the fp32-kind channels) of
 * the fp32-MFMA time.
 * Input = "xrec": padded NHWC of bf16 RECORDS, mp_xrec_elements(n_f32, n_u8) = roundup8(3 n_f32 + n_u8) elements per pixel:
 *   [x1,x2,x3 of the first fp32-kind channel | .. | of the last | k of the first integer channel | .. | zero padding]   (what MP_RASTER_XREC writes).
 * mp_conv_stem_xrec: desc as for mp_conv2d_nhwc with d_x = the record tensor, C ignored, c_real = n_f32 + n_u8; KH = KW in {5, 7},
 * stride 2, Cout % 64 == 0, records of 16..48 elements, no residual / second output. */
int mp_xrec_elements(int n_f32, int n_u8);
int mp_conv_stem_supported(int KS, int n_f32, int n_u8);
size_t mp_conv_stem_packed_bytes(int KS, int n_f32, int n_u8, int Cout);
int mp_conv_stem_pack_weights(const float* h_w_oihw, int Cout, int Cin, int KS, int n_f32, const float* h_scale /*[Cout] or NULL*/,
                              void* h_packed);
/* the same for a record whose fp32-kind channels are not the leading ones: input channel c (< 32) is fp32-kind iff bit c of f32_mask is
 * set (an RGBD refiner: crop rgb + crop depth + one rendered depth per view = 0x8102040F for 32 channels, 4 views) */
int mp_conv_stem_pack_weights_mask(const float* h_w_oihw, int Cout, int Cin, int KS, uint32_t f32_mask, const float* h_scale,
                                   void* h_packed);
int mp_conv_stem_xrec(const mp_conv_desc* desc, const void* d_packed, int n_f32, mp_stream stream);
/* the same with the 3x3 / stride-2 / pad-1 max pool that follows the stem (models/torchvision_resnet.py:216) fused into the epilogue:
 * d_ypool = padded NHWC [N, (Ho-1)/2+1, (Wo-1)/2+1, Cout] with border pool_border; desc->relu must be set; desc->d_y may be NULL (the stem
 * map is then never written).  Windows that straddle the kernel's 8 x 16 tiles are combined with unsigned atomicMax on the (non-negative)
 * float bits: deterministic. */
int mp_conv_stem_xrec_pool(const mp_conv_desc* desc, const void* d_packed, int n_f32, float* d_ypool, int pool_border, mp_stream stream);

/* Background tiles (round 5).  A stem workgroup (8 x 16 output pixels) whose input patch holds no rendered geometry -- every integer
 * channel of every pixel 0: 47 % of a refiner step's tiles (measured, profiles/r05_stem_sparse_ab.txt) -- needs only the record chunks that hold fp32-kind pieces:
 * mp_conv_stem_sparse_chunks = ceil(3 n_f32 / 8) if that is fewer than the record's chunks and the fp32-kind channels are the leading
 * ones (no depth channels), else 0.  mp_conv_stem_xrec_sparse = mp_conv_stem_xrec[_pool] (d_ypool may be NULL) that takes, besides the
 * dense blob, the blob packed for that short walk and the rasteriser's job flags of the launch that wrote the records
 * (mp_raster_job_flags): such workgroups run 26 instead of 62 steps (7x7, 40-element records) and stage 2 of 5 chunks.  Skipped products
 * are exact zeros; the evaluated ones are grouped into MFMAs differently from the dense walk (order of the fp32 additions).
 * (mp_conv_stem_bg_stats, mp_engine_debug.h: how many workgroups took the short walk.) */
int mp_conv_stem_sparse_chunks(int KS, int n_f32, int n_u8);
size_t mp_conv_stem_sparse_packed_bytes(int KS, int n_f32, int n_u8, int Cout);
int mp_conv_stem_pack_weights_sparse(const float* h_w_oihw, int Cout, int Cin, int KS, int n_f32, const float* h_scale, void* h_packed);
int mp_conv_stem_xrec_sparse(const mp_conv_desc* desc, const void* d_packed, const void* d_packed_sparse, int n_f32,
                             const unsigned char* d_tile_flags, float* d_ypool, int pool_border, mp_stream stream);

/* 3x3 stride-2 pad-1 max pool on padded NHWC (input must be >= 0, i.e. post-ReLU).        */
int mp_maxpool3x3s2(const float* d_x, int N, int H, int W, int C, int in_border, float* d_y,
                    int out_border, float* d_y_act, const float* d_act_scale, const float* d_act_shift,
                    mp_stream stream);

/* y_act = relu(x * scale[c] + shift[c]) on the interior of a padded NHWC map (same border in and out): the pre-activation of a
 * WideResNet's first block (models/wide_resnet.py:29-44 bn1 / relu) when the max pool in front of it is fused into the stem
 * (mp_conv_stem_xrec_pool) -- bit-identical to the second output of mp_maxpool3x3s2. */
int mp_bn_relu_nhwc(const float* d_x, int N, int H, int W, int C, int border, float* d_y_act, const float* d_act_scale,
                    const float* d_act_shift, mp_stream stream);

/* global average pool (+ optional fc) + heads: models/torchvision_resnet.py:311-314 and  */
/* models/pose_rigid.py:326-333.  d_fc_w may be NULL (WideResNet: features = pooled).      */
int mp_pool_fc_heads(const float* d_x, int N, int H, int W, int C, int in_border, const float* d_fc_w,
                     const float* d_fc_b, int n_feat, const float* d_head_w, const float* d_head_b,
                     int n_out, float* d_feat /*[N,n_feat] or NULL*/, float* d_out /*[N,n_out]*/,
                     float* d_sigmoid /*[N,n_out] or NULL*/, mp_stream stream);

/* ------------------------------------------------------------------------------------ */
/* Backbone executor: whole network as one call (weights resident on the device).         */
/* ------------------------------------------------------------------------------------ */
typedef struct mp_backbone mp_backbone;
#define MP_BACKBONE_VANILLA_RESNET34 0
#define MP_BACKBONE_WIDE_RESNET34 1
#define MP_BACKBONE_WIDE_RESNET18 2

typedef struct {
  const char* name;    /* state_dict key, e.g. "backbone.layer1.0.conv1.weight"             */
  const float* h_data; /* host fp32, contiguous                                              */
  int64_t numel;
} mp_named_tensor;

/* state_dict layout = the reference checkpoints' (SURVEY.md App. F): backbone.*, pose_fc.*, */
/* views_logits_head.*.  head: 0 = pose_fc (9 outputs), 1 = views_logits_head (n_views).     */
int mp_backbone_create(int kind, int c_in, int head_kind, int n_head_out, const mp_named_tensor* h_state,
                       int n_tensors, mp_backbone** out);
/* the same with the WideResNet width multiplier of `resnet34_width=N` (training/pose_models_cfg.py:114-116, models/wide_resnet.py:62:
 * stage widths 64N .. 512N, features 512N); width = 1 for the released models */
int mp_backbone_create_wide(int kind, int width, int c_in, int head_kind, int n_head_out, const mp_named_tensor* state, int n_tensors,
                            mp_backbone** out);
int mp_backbone_destroy(mp_backbone* bb);
int mp_backbone_input_channels_padded(const mp_backbone* bb);
int mp_backbone_input_border(const mp_backbone* bb);
size_t mp_backbone_workspace_bytes(const mp_backbone* bb, int batch, int h, int w);
/* The executor zeroes a workspace's borders once per (pointer, batch, h, w) and then trusts them.  Call this whenever the
 * memory behind `d_workspace` is a NEW allocation (the owner knows; an allocator may hand an old address back after foreign
 * writes): the next forward on it re-zeroes.                                                                            */
int mp_backbone_workspace_reset(mp_backbone* bb, const void* d_workspace);
/* d_x: padded NHWC [batch, h, w, Cp] with border mp_backbone_input_border().                */
int mp_backbone_forward(mp_backbone* bb, const float* d_x, int batch, int h, int w, float* d_out,
                        float* d_sigmoid, float* d_feat, void* d_workspace, size_t workspace_bytes,
                        mp_stream stream);
/* the same forward on a half-precision input tensor (binary16 elements, same padded-NHWC geometry: what the rasteriser writes  */
/* with MP_RASTER_F16).  Only the stem convolution differs (it widens the halves on their way into LDS).                       */
int mp_backbone_forward_f16(mp_backbone* bb, const void* d_x_half, int batch, int h, int w, float* d_out,
                            float* d_sigmoid, float* d_feat, void* d_workspace, size_t workspace_bytes,
                            mp_stream stream);
/* the same forward on the bf16 stem RECORDS the rasteriser writes with MP_RASTER_XREC (n_f32 fp32-kind channels first, all other   */
/* input channels 8-bit integers): only the stem convolution differs (mp_conv_stem_xrec: exact bf16 pieces, 16x the fp32 MFMA rate). */
/* mp_backbone_xrec_elements: record length in bf16 elements for this backbone (packs the piece blob on first use), 0 = the stem has   */
/* no such form (records outside 16..48 elements): use mp_backbone_forward.  The tensor has the geometry    */
/* of the fp32 input (border mp_backbone_input_border()) with records of that many bf16 elements per pixel.                            */
/* PRECONDITION of every mp_backbone_forward_xrec*: mp_backbone_xrec_elements / mp_backbone_xrec_prepare was called for that mask before   */
/* (once, outside stream capture): the forwards only look the blob up and fail with MP_ERR_INVALID if it is missing -- they never allocate. */
int mp_backbone_xrec_elements(mp_backbone* bb, int n_f32);
int mp_backbone_forward_xrec(mp_backbone* bb, const void* d_xrec, int n_f32, int batch, int h, int w, float* d_out,
                             float* d_sigmoid, float* d_feat, void* d_workspace, size_t workspace_bytes,
                             mp_stream stream);
/* ... with the fp32-kind channels given as a mask (bit c = input channel c; depth channels of an RGBD model).  mp_backbone_xrec_prepare
 * packs and uploads the stem's piece blob for that mask (host work + a synchronous copy: call it once, outside stream capture, from one
 * thread) and returns the record length (0 = no exact-piece form); mp_backbone_forward_xrec_mask only uses a prepared blob and fails if
 * there is none -- it never allocates. */
int mp_backbone_xrec_prepare(mp_backbone* bb, uint32_t f32_mask);
int mp_backbone_forward_xrec_mask(mp_backbone* bb, const void* d_xrec, uint32_t f32_mask, int batch, int h, int w, float* d_out,
                                  float* d_sigmoid, float* d_feat, void* d_workspace, size_t workspace_bytes,
                                  mp_stream stream);
/* ... with the rasteriser's job flags of the launch that wrote the records (mp_raster_job_flags; NULL = dense): the stem takes the
 * background-tile walk where it applies (mp_conv_stem_xrec_sparse; the blob for it is packed by mp_backbone_xrec_prepare) */
int mp_backbone_forward_xrec_sparse(mp_backbone* bb, const void* d_xrec, uint32_t f32_mask, const unsigned char* d_tile_flags, int batch,
                                    int h, int w, float* d_out, float* d_sigmoid, float* d_feat, void* d_workspace,
                                    size_t workspace_bytes, mp_stream stream);
/* algorithmic conv+fc FLOPs of one forward at this batch (2*MACs, real channels only)       */
double mp_backbone_flops(const mp_backbone* bb, int batch, int h, int w);

/* ------------------------------------------------------------------------------------ */
/* Pose math (all fp32, one thread block per row)                                          */
/* ------------------------------------------------------------------------------------ */
/* lib3d/transform_ops.py:117-119 normalize_T (ortho6d Gram-Schmidt, rotations.py:25-40)    */
int mp_normalize_T(const float* d_T, int b, float* d_T_out, mp_stream stream);

/* per-(mesh, rotation) extents for TCO_init_from_boxes_autodepth_with_R                     */
/* (lib3d/cosypose_ops.py:198-208): d_ext[(mesh*n_rot + r)*2 + {0,1}] = max-min of x,y of R p */
int mp_init_extents(const float* d_points /*[n_mesh,n_pts,3]*/, int n_mesh, int n_pts,
                    const float* d_R /*[n_rot,3,3]*/, int n_rot, float* d_ext, mp_stream stream);
/* lib3d/cosypose_ops.py:169-218.  row i uses mesh d_mesh_ids[i], rotation d_rot_ids[i]      */
int mp_init_poses_from_boxes(const float* d_boxes /*[b,4]*/, const float* d_K /*[b,3,3]*/,
                             const int32_t* d_mesh_ids, const int32_t* d_rot_ids, const float* d_R,
                             int n_rot, const float* d_ext, int b, float* d_TCO /*[b,4,4]*/,
                             mp_stream stream);

/* One refiner/coarse "prepare" step for b rows x V views:                                  */
/*   TCO_n = normalize_T(TCO)                     (models/pose_rigid.py:524, :678)           */
/*   tCR   = TCO_n[:3,3]                          (:527-529)                                 */
/*   TCV_O = make_TCO_multiview(...)              (lib3d/multiview.py:165-246; App. A.5)     */
/*   boxes_rend/boxes_crop/K_crop from 2000 pts   (pose_rigid.py:180-247 crop_inputs)        */
/*   KV_crop from 200 pts for views >= 1          (:249-303; KV_crop[:,0] = K_crop :551-552) */
/* multiview: low byte = mode: 0 single view (V = 1), 1 "TCO+front_3views", 2 "TCO+front_1view", 3 "sphere_26views"            */
/* (lib3d/multiview.py:197-234), | MP_MV_REMOVE_TCO (remove_TCO_rendering: the TCO view is not in the list and KV_crop[:,0] is  */
/* NOT replaced by K_crop, models/pose_rigid.py:551-552), | MP_MV_INPLANE (views_inplane_rotations: every view 4x, rotated by   */
/* 0/90/180/270 degrees about the optical axis, multiview.py:236-245).  V must equal mp_pose_multiview_n_views(multiview).       */
/* mp_pose_prepare_ex additionally returns K_crop of crop_inputs ([b,3,3], what update_pose consumes) in d_K_main (may be NULL). */
#define MP_MV_REMOVE_TCO 256
#define MP_MV_INPLANE 512
int mp_pose_multiview_n_views(int multiview);
int mp_pose_prepare_ex(const float* d_TCO_in, const float* d_K, const int32_t* d_mesh_ids, const float* d_points, int n_pts_stride,
                       int n_pts_main, int n_pts_views, int b, int V, int multiview, int im_h, int im_w, int out_h, int out_w, float lamb,
                       float* d_TCO_n, float* d_tCR, float* d_TCV_O, float* d_KV_crop, float* d_boxes_rend, float* d_boxes_crop,
                       float* d_K_main /*[b,3,3] or NULL*/, mp_stream stream);
int mp_pose_prepare(const float* d_TCO_in /*[b,4,4]*/, const float* d_K /*[b,3,3]*/,
                    const int32_t* d_mesh_ids, const float* d_points /*[n_mesh,n_pts,3] sampled*/,
                    int n_pts_stride, int n_pts_main, int n_pts_views, int b, int V, int multiview,
                    int im_h, int im_w, int out_h, int out_w, float lamb,
                    float* d_TCO_n /*[b,4,4]*/, float* d_tCR /*[b,3]*/, float* d_TCV_O /*[b,V,4,4]*/,
                    float* d_KV_crop /*[b,V,3,3]*/, float* d_boxes_rend /*[b,4]*/,
                    float* d_boxes_crop /*[b,4]*/, mp_stream stream);

/* models/pose_rigid.py:305-312 update_pose + lib3d/cosypose_ops.py:33-58                    */
int mp_pose_update(const float* d_TCO /*[b,4,4]*/, const float* d_K_crop /*[b,3,3] (stride 9*kstride)*/,
                   int k_stride_floats, const float* d_out9 /*[b,9]*/, const float* d_tCR /*[b,3]*/, int b,
                   float* d_TCO_out, mp_stream stream);

/* ------------------------------------------------------------------------------------ */
/* Depth refiner (ICP): replaces inference/icp_refiner.py:128-175 icp_refinement +          */
/* :195-262 ICPRefiner.refine_poses (masks refiner_utils.py:30-56).  The reference's ICP    */
/* core is OpenCV-contrib ppf_match_3d_ICP (third party, parity unpinned); this is a        */
/* projective point-to-plane ICP with the same budget / acceptance rule (csrc/icp.hip).     */
/* d_depth_meas [n_images,H,W] metres (0 = invalid), d_depth_rend [n_rows,H,W] rendered at  */
/* d_TCO, d_K_images [n_images,3,3], d_K_rows [n_rows,3,3].  retval[n] = 0 ok / -1 kept.    */
/* user_masks != 0: the caller's segmentation masks have been applied to d_depth_meas and     */
/* the |measured - rendered| <= 0.1 m test is skipped (icp_refiner.py:249-250).               */
/* ------------------------------------------------------------------------------------ */
size_t mp_icp_workspace_bytes(int n_images, int n_rows, int H, int W);
int mp_icp_refine(const float* d_depth_meas, int n_images, const int32_t* d_im_ids, const float* d_depth_rend,
                  const float* d_K_images, const float* d_K_rows, const float* d_TCO, int n_rows, int H, int W,
                  int n_iterations, int n_levels, float tolerance, int n_min_points, int user_masks, float* d_TCO_out,
                  int32_t* d_retval, float* d_residual, void* d_workspace, size_t workspace_bytes, mp_stream stream);

/* The same refiner with the REFERENCE's algorithm, step for step (csrc/icp_nn.hip): get_normal (hole fill, Gaussian sigma 2, gradient,
 * inference/icp_refiner.py:37-95), getXYZ, masks / 1000-point rule, centroid pre-shift, and OpenCV's ppf_match_3d ICP as restated in
 * oracle/icp_opencv.py (mean / scale normalisation, 4 levels, exact nearest neighbours, median + 2.5 MAD rejection, one-to-one filter,
 * linearised point-to-plane solve, relative-change stop).  Arguments as mp_icp_refine (n_iterations = 100, n_levels = 4, tolerance = 0.05
 * are the reference's) except d_masks: the caller's per-frame masks [n_images][H][W] uint8 (icp_refiner.py:249-250: they replace the
 * threshold mask and only select points -- the measured depth is passed unmasked, its normals come from the whole frame) or NULL; d_iters (optional, [n_rows][8] int32) receives the iterations run per level.  At most
 * mp_icp_nn_max_points() mask pixels per object (more -> that object is rejected, retval -1). */
int mp_icp_nn_max_points(void);
size_t mp_icp_nn_workspace_bytes(int n_images, int n_rows, int H, int W);
int mp_icp_refine_nn(const float* d_depth_meas, int n_images, const int32_t* d_im_ids, const float* d_depth_rend, const float* d_K_images,
                     const float* d_K_rows, const float* d_TCO, int n_rows, int H, int W, int n_iterations, int n_levels, float tolerance,
                     int n_min_points, const unsigned char* d_masks, float* d_TCO_out, int32_t* d_retval, float* d_residual, int32_t* d_iters,
                     void* d_ws, size_t ws_bytes, mp_stream stream);

/* ------------------------------------------------------------------------------------ */
/* Detector network: replaces the torchvision Mask R-CNN    */
/* behind `self.model([image_n ...])` in inference/detector.py:92                          */
/* (models/mask_rcnn.py:23-46 = MaskRCNN(resnet_fpn_backbone("resnet50"), num_classes,     */
/* AnchorGenerator(((32,),(64,),(128,),(256,),(512,)), ((0.5,1,2),)*5), min/max_size);     */
/* all other hyper-parameters torchvision 0.12 defaults).  One call runs the whole         */
/* inference graph on the device: normalise + resize + pad, ResNet-50 + FPN (the fp32 MFMA */
/* convolution of this library, FrozenBatchNorm folded), RPN (top-k, decode, NMS), RoIAlign,*/
/* box head (the two FC layers run as 1x1 convolutions on the same kernel), per-class NMS, */
/* mask head and mask pasting.  No host synchronisation, deterministic (ties resolve to    */
/* the lower index).  csrc/detector.hip.                                                   */
/* ------------------------------------------------------------------------------------ */
typedef struct mp_detector mp_detector;

typedef struct {
  int32_t n_classes;               /* including the background class 0                                           */
  int32_t min_size, max_size;      /* GeneralizedRCNNTransform (cfg.input_resize: (480, 640) for the released detectors) */
  float image_mean[3], image_std[3];
  int32_t anchor_sizes[5];         /* one per pyramid level P2..P6                                                */
  float aspect_ratios[3];
  int32_t rpn_pre_nms_top_n, rpn_post_nms_top_n;   /* <= 1024 each                                              */
  float rpn_nms_thresh, rpn_score_thresh, rpn_min_size;
  float box_score_thresh, box_nms_thresh, box_min_size;
  int32_t box_detections_per_img;                  /* <= 1024                                                    */
} mp_detector_config;

/* torchvision's eval defaults: 1000 / 1000 / 0.7 / 0.0 / 1e-3, 0.05 / 0.5 / 1e-2 / 100, ImageNet mean / std, the reference's anchors */
int mp_detector_default_config(mp_detector_config* cfg, int n_classes, int min_size, int max_size);
/* The checkpoint layout the detector expects (torchvision's state_dict keys: backbone.body.*, backbone.fpn.*, rpn.head.*,
 * roi_heads.*): entry `idx` -> name, dims.  Returns 1 past the end.  Host only. */
int mp_detector_state_spec(int n_classes, int idx, char* name, int name_len, int64_t* shape4, int32_t* n_dims);
int mp_detector_create(const mp_detector_config* cfg, const mp_named_tensor* h_state, int n_tensors, mp_detector** out);
int mp_detector_destroy(mp_detector* det);
size_t mp_detector_workspace_bytes(const mp_detector* det, int n_images, int H, int W);
/* d_images [n,3,H,W] fp32 in [0,1] (what Detector.get_detections passes, detector.py:88-92).  Outputs, D = box_detections_per_img:
 * d_boxes [n,D,4] (x1,y1,x2,y2 in ORIGINAL image pixels), d_scores [n,D], d_labels [n,D] (category ids >= 1), d_counts [n]; entries
 * past the count are zero.  d_masks: NULL or [n,D,H,W] soft masks in [0,1] pasted into the original frame (roi_heads.py
 * paste_masks_in_image); the caller thresholds them (detector.py:106).  Images of one call share H x W. */
int mp_detector_forward(mp_detector* det, const float* d_images, int n_images, int H, int W, float* d_boxes, float* d_scores,
                        int32_t* d_labels, int32_t* d_counts, float* d_masks, void* d_workspace, size_t workspace_bytes,
                        mp_stream stream);
/* Parity taps (tests): after a forward, the device address + logical shape of an intermediate inside `d_workspace`:
 * "P2".."P6" padded-NHWC pyramid levels {n, h, w, 256} with border 1; "proposals" {n, post_nms_top_n, 4} + "proposal_counts" {n} (int32);
 * "class_logits" {n*post, padded 5*n_classes row: n_classes logits then 4*n_classes deltas}; "mask_logits" {n*D*14*14*4, padded n_classes}.
 * Returns MP_ERR_INVALID for an unknown name or before the first forward. */
int mp_detector_debug_tensor(const mp_detector* det, const char* what, const void** d_ptr, int64_t* shape4, int32_t* border,
                             int64_t* row_stride, int64_t* n_elements /* 4-byte elements of the whole buffer (borders / row padding included) */);

#ifdef __cplusplus
}
#endif
#endif /* MP_ENGINE_H */
