// conv_stem.hip -- the stem convolution (7x7 / 5x5, stride 2) on the bf16 MFMA through EXACT operand pieces, gfx950.
//
// Replaces the first convolution behind `self.backbone(x)` (reference: src/megapose/models/pose_rigid.py:323; layers
// src/megapose/models/torchvision_resnet.py:213-216 conv1/bn1/relu, src/megapose/models/wide_resnet.py:65-67) -- 24 % of a
// config-2 step on the fp32 MFMA (conv.hip, 0.75 of the fp32 matrix peak: nothing left to gain there).
//
// Why this is exact and not a reduced-precision mode.  All but three of the CNN input channels are RENDERS, and a render value is
// an 8-bit integer k / 255 by the reference's own contract (uint8 framebuffer -> float, panda3d_batch_renderer.py:261-274; the
// 4x multisample resolve rounds to 8 bits again).  k <= 255 has 8 significant bits = ONE bf16, exactly.  A weight w (eval-BN scale
// and the 1/255 folded in) is split by truncation into three bf16 pieces w = w1 + w2 + w3 EXACTLY (24 = 3 x 8 mantissa bits).
// Every product k * w_i then has <= 16 significant bits: exact in the MFMA's fp32 accumulator.  The three observation-crop
// channels (roi_align output: general fp32) are split the same way, x = x1 + x2 + x3, and get all 9 exact piece products.  So
//     y = sum_taps ( sum_i k w_i  |  sum_{i,j} x_i w_j )      accumulated in fp32
// differs from the fp32-MFMA convolution in the ORDER of fp32 additions (same error class as any re-association, far below the
// Winograd layers' 1e-6) and in ONE rounding per integer-channel weight: the 1/255 of a render value is folded into the weight,
// fl32(w * scale / 255) (one rounding of the exact product, formed in double), and multiplied with the integer k exactly, where the
// fp32 path -- like the reference -- multiplies fl32(k / 255) with fl32(w * scale): each such product differs by <= 1 ulp (the record
// tests bound the whole-network effect at 1e-5 of the feature scale).  v_mfma_f32_16x16x32_bf16 retires 16x the multiply-adds per cycle
// of v_mfma_f32_32x32x2_f32: 3 (9) bf16 products per fp32 product = 3/16 (9/16) of the matrix time.
//
// Input ("xrec", written by raster_tiles with MP_RASTER_XREC): padded NHWC of bf16 RECORDS, R = 8*Q elements per pixel (Q = 2 .. 6):
//   [x1,x2,x3 of the first fp32-kind channel | ... of the last | k of the first integer channel | ... | zero padding].
// fp32-kind = the observation crop and, for RGBD models, every (normalised) depth channel: mp_conv_stem_pack_weights_mask.
// Structure (MI355X-first, not a GEMM library shape):
//   * workgroup = 8 x 16 output pixels x 64 output channels, 4 waves, TWO workgroups per CU.  The input patch of the tile
//     ((2*8+KH-2) x (2*16+KW-2) pixel records, 62 KB at 7x7 / Q = 5) is staged in LDS ONCE; there is no im2col anywhere: a K slice
//     of 8 elements = one 16-byte chunk of one pixel record, so an A fragment of the 16x16x32 MFMA (lane = pixel l & 15, K group
//     l >> 4 = slice 4t + (l >> 4)) is ONE ds_read_b128 at  pixel(l & 15) + tap/chunk offset(slice)  -- one address VGPR per step,
//     the 8 M-tiles (output rows) of the tile are immediate offsets.  With stride 2 every pixel sits at an even 16-byte unit and
//     consecutive slices at odd distance, which is exactly what the b128 lane groups need to be conflict-free (Q = 5).
//   * wave w owns output channels 16w .. 16w+15 for all 128 pixels: 8 accumulators of 4 registers.  Its weight fragments come
//     from L2 STRAIGHT into registers in fragment order (host-packed [cb][wave][step][piece][lane][8 bf16], 3 KB per step, prefetched
//     two steps ahead): every weight byte is loaded by exactly one wave of the workgroup, no LDS write traffic.
//   * per step: 8 ds_read_b128 + 3 buffer loads + 24 MFMAs (8 M-tiles x 3 weight pieces, same A fragment).
//   * epilogue: bias (folded BN) + ReLU, the tile goes through the (now free) patch LDS and leaves as whole 256-byte pixel rows.
// Roofline: MFMA (bf16) bound; algorithmic work = 2 * MACs of the direct convolution over the real channels.
#include <cstdlib>
#include <vector>

#include "common.h"

namespace mp {
namespace stem {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16;   // output tile: 8 rows x 16 columns; M-tile j of the MFMA = output row j
constexpr int NCO = 64;          // output channels per workgroup (4 waves x 16)

struct Params {
  const unsigned char* __restrict__ x;   // xrec tensor [N][Hp][Wp][8Q] bf16
  const unsigned char* __restrict__ w;   // packed pieces
  const float* __restrict__ bias;
  float* __restrict__ y;
  int N, Ho, Wo;
  int Hp, Wp;            // padded input size in pixels
  int in_off;            // in_border - pad
  int Cout, Hop, Wop, out_border;
  int tiles_x, tiles_y, n_cb;
  int n_steps;           // even
  int relu;
  unsigned img_bytes;    // Hp * Wp * 16Q
  unsigned out_bytes;    // of ONE image of y: Hop * Wop * Cout * 4
  // fused 3x3 / stride-2 / pad-1 max pool (POOL instances): pooled output, padded NHWC with border pool_border
  float* __restrict__ ypool;
  int Hq, Wq, pool_border;
  // background tiles (SP instances): the rasteriser's per-(item, 8x8-pixel tile) job flags (0 = no view reaches the tile: every integer
  // channel of its pixels is 0) and the piece blob of the slices that hold fp32-kind pieces only (record chunks q < q_use)
  const unsigned char* __restrict__ tile_flags;   // [N][fl_ty][fl_tx]
  const unsigned char* __restrict__ w_sparse;
  int fl_tx, fl_ty, q_use, n_steps_sparse;
  int H, W, pad;         // input size in pixels, convolution padding (to place the patch in the flag grid)
  int count_bg;          // != 0: one atomic per background workgroup into g_stem_bg (mp_conv_stem_bg_stats; only while the event profiler runs)
};

__device__ unsigned long long g_stem_bg[2];   // workgroups that took the background-tile walk | workgroups of SP launches (counted launches only)

constexpr size_t LDS_PER_WG = 80 * 1024;   // two workgroups per CU share the 160 KB

template <int KS, int Q>
struct Geo {
  static constexpr int PH = 2 * (TH - 1) + KS, PW = 2 * (TW - 1) + KS;   // patch size in pixels
  // LDS pixel pitch in 16-byte chunks.  The 8 lanes a ds_read_b128 serves per clock are 8 consecutive output pixels = input pixels two apart:
  // their chunk addresses are 2 i QP (+ a common offset) and fall into 8 different 16-byte bank groups iff QP is ODD.  An even record
  // (Q = 2: the coarse net, Q = 4, Q = 6: RGBD) is therefore staged with one chunk of padding per pixel wherever the patch still fits two
  // workgroups per CU (round 4's Q = 2 instance ran with 2-way conflicts: SQ_LDS_BANK_CONFLICT 1.9e9 per launch); the 7x7 / Q = 6 patch
  // does not fit padded and keeps its 2-way conflict.
  static constexpr int QP = (Q % 2 == 0 && (size_t)(PH + 1) * ((PW * (Q + 1)) | 1) * 16 <= LDS_PER_WG) ? Q + 1 : Q;
  static constexpr int ROW16 = PW * QP;                                   // 16-byte chunks per patch row (incl. the per-pixel padding)
  static constexpr int PITCH16 = ROW16 | 1;                               // odd pitch (in chunks): row wraps keep the odd slice distance
  static constexpr int PITCH = PITCH16 * 16;
  static constexpr int KWQ = KS * Q;                                      // slices per kernel row
  static constexpr int S = KS * KWQ;                                      // slices in all
  static constexpr int T = ((S + 3) / 4 + 1) / 2 * 2;                     // steps of 4 slices, even
  static constexpr size_t PATCH = (size_t)(PH + 1) * PITCH;               // + one zero row: where the padding slices of the last steps point
  static constexpr size_t OUT_TILE = (size_t)TH * TW * NCO * 4;           // the output tile is staged in the same memory
  static constexpr size_t LDS = PATCH > OUT_TILE ? PATCH : OUT_TILE;
  static_assert(4 * T + 4 - S <= KWQ, "padding slices (and the look-ahead read past the last step) must stay inside the extra row");
  static_assert(LDS <= LDS_PER_WG, "the patch must leave room for a second workgroup on the CU");
};

// POOL: the 3x3 / stride-2 / pad-1 max pool that follows the stem (models/torchvision_resnet.py:216) is taken from the tile while it sits in
// LDS: a pooled pixel whose window lies inside the tile (21 of the 45 an 8 x 16 tile touches) is stored; one whose window straddles tiles
// is combined with unsigned atomicMax on the float bits (the values are post-ReLU, >= 0: the integer order IS the float order; max is
// associative and commutative, so the result is deterministic) into a position that pool_zero_kernel cleared beforehand.  The stem map
// itself is then never written (y may be NULL) and the separate pool kernel + its 2.9-GB read at 576 rows disappear.
// SP (round 5): BACKGROUND TILES.  47 % (measured, profiles/r05_stem_sparse_ab.txt) of a refiner step's stem tiles see no rendered geometry in any view (the object fills ~30 % of its
// crop): every integer (render) channel of every pixel of their input patch is 0, and 3 of the 5 record chunks contribute exact zeros.
// The rasteriser already knows which 8x8-pixel tiles no view reaches (raster_classify's job flags); a workgroup whose whole patch lies in
// such tiles walks only the slices of the chunks that hold fp32-kind pieces (q < q_use: 2 of 5 for the RGB refiner) with a piece blob
// packed for that walk: 26 instead of 62 steps, and it stages only those chunks.  Every product that is evaluated is exact and every
// skipped one is an exact zero; the grouping of the remaining products into MFMAs differs from the dense walk, i.e. the order of the fp32
// additions (the error class of every other launch-shape difference, bounded by the record tests at 1e-5 of the feature scale).
template <int KS, int Q, bool POOL, bool SP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_stem_bf16x3(Params p) {
  using G = Geo<KS, Q>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int cb = wg % p.n_cb; wg /= p.n_cb;
  const int tx = wg % p.tiles_x; wg /= p.tiles_x;
  const int ty = wg % p.tiles_y;
  const int n = wg / p.tiles_y;
  bool sparse = false;
  if constexpr (SP) {   // does any 8x8-pixel tile under the input patch hold geometry?  (<= 4 x 6 flag bytes, one per lane; wave-uniform result)
    const int r0 = max(2 * ty * TH - p.pad, 0) >> 3, r1 = min(2 * ty * TH - p.pad + G::PH - 1, p.H - 1) >> 3;
    const int c0 = max(2 * tx * TW - p.pad, 0) >> 3, c1 = min(2 * tx * TW - p.pad + G::PW - 1, p.W - 1) >> 3;
    const int nc = c1 - c0 + 1, cnt = (r1 - r0 + 1) * nc;
    unsigned char f = 0;
    if (lane < cnt) f = p.tile_flags[((size_t)n * p.fl_ty + r0 + lane / nc) * p.fl_tx + c0 + lane % nc];
    sparse = cnt <= 64 && __ballot(f != 0) == 0ull;
  }
  if constexpr (SP) {
    if (p.count_bg && tid == 0) {
      if (sparse) atomicAdd(&g_stem_bg[0], 1ull);
      if (blockIdx.x == 0) atomicAdd(&g_stem_bg[1], (unsigned long long)gridDim.x);
    }
  }
  const int QU = SP && sparse ? p.q_use : Q;                       // record chunks walked per pixel
  const int n_steps = SP && sparse ? p.n_steps_sparse : p.n_steps;

  // ---- patch: PH rows of ROW16 16-byte chunks, global -> registers -> LDS (rows beyond the image: range-checked to zero) ------------
  {
    const unsigned char* img = p.x + (size_t)n * p.img_bytes;
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, p.img_bytes, 0x00020000);
    const int row_bytes = p.Wp * (16 * Q);
    const int org = (2 * ty * TH + p.in_off) * row_bytes + (2 * tx * TW + p.in_off) * (16 * Q);
    constexpr int SRC16 = G::PW * Q;   // chunks per patch row in global memory (records are dense there)
    constexpr int N_CH = G::PH * SRC16, PER = (N_CH + 255) / 256;
    constexpr int HALF = (PER + 1) / 2;
    u32x4 v[HALF];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int k = 0; k < HALF; ++k) {
        const int c = tid + 256 * (h * HALF + k);
        const int row = c / SRC16, col = c - row * SRC16;
        const bool want = c < N_CH && (!SP || col % Q < QU);   // (background tile: only the chunks that are walked)
        v[k] = want ? __builtin_amdgcn_raw_buffer_load_b128(r_x, org + row * row_bytes + col * 16, 0, 0) : u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int k = 0; k < HALF; ++k) {
        const int c = tid + 256 * (h * HALF + k);
        const int row = c / SRC16, col = c - row * SRC16;
        const int lcol = G::QP == Q ? col : col + (col / Q) * (G::QP - Q);   // pixel * QP + q
        if (c < N_CH && (!SP || col % Q < QU)) *reinterpret_cast<u32x4*>(lds + row * G::PITCH + lcol * 16) = v[k];
      }
    }
    // (the padding chunk of every pixel is never read: a slice's chunk index is q < Q)
    for (int c = tid; c < G::PITCH16; c += 256) *reinterpret_cast<u32x4*>(lds + G::PH * G::PITCH + c * 16) = u32x4{0, 0, 0, 0};
  }

  // ---- weights: this wave's stream [step][piece][lane][16 B] ----------------------------------------------------------------------
  const size_t w_wave = ((size_t)cb * 4 + wave) * n_steps * 3072;
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)((SP && sparse ? p.w_sparse : p.w) + w_wave), 0, (unsigned)n_steps * 3072u, 0x00020000);
  const int w_voff = lane * 16;
  u32x4 b0[3], b1[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    b0[q] = __builtin_amdgcn_raw_buffer_load_b128(r_w, w_voff, q * 1024, 0);
    b1[q] = __builtin_amdgcn_raw_buffer_load_b128(r_w, w_voff, 3072 + q * 1024, 0);
  }

  // ---- A addressing: lane = (pixel column i, slice group g); slice s = 4 t + g -> (kernel row, chunk in the row run) ----------------
  const int i = lane & 15, g = lane >> 4;
  int s_r = g;                                   // s % KWQ (g < KWQ)
  int a_row = (2 * i) * (16 * G::QP);            // pixel (kernel row 0, column i); + PITCH per kernel row
  // slice s_r of a kernel row = (kw, q) = (s_r / Q, s_r % Q) sits at chunk kw * QP + q = s_r + kw * (QP - Q) of the row run
  // (SP: QU chunks per pixel are walked: kw = sr / QU by a multiply-shift that is exact for sr < 64, QU <= 6)
  const int kwq = SP ? KS * QU : G::KWQ;
  const int qu_m = 1024 / QU + 1;
  auto slice_off = [&](int sr) {
    if constexpr (SP) {
      const int kw = (sr * qu_m) >> 10;
      return (kw * G::QP + (sr - kw * QU)) * 16;
    } else {
      return G::QP == Q ? sr * 16 : (sr + (sr / Q) * (G::QP - Q)) * 16;
    }
  };
  int a_off = a_row + slice_off(s_r);
  auto advance = [&]() {                         // s += 4
    s_r += 4;
    if (s_r >= kwq) { s_r -= kwq; a_row += G::PITCH; }
    a_off = a_row + slice_off(s_r);
  };
  f32x4 acc[TH];
#pragma unroll
  for (int j = 0; j < TH; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a0[TH], a1[TH];

  __syncthreads();
#pragma unroll
  for (int j = 0; j < TH; ++j) a0[j] = *reinterpret_cast<const u32x4*>(lds + a_off + j * 2 * G::PITCH);
  advance();

  const int n_pairs = n_steps >> 1;
  for (int tp = 0; tp < n_pairs; ++tp) {
    const int w_next = (2 * tp + 2) * 3072;   // (past the end in the last pair: range-checked, never used)
    // even step: fragments a0 / b0; read a1 for the odd step
#pragma unroll
    for (int j = 0; j < TH; ++j) a1[j] = *reinterpret_cast<const u32x4*>(lds + a_off + j * 2 * G::PITCH);
    advance();
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bf16x8 bq = __builtin_bit_cast(bf16x8, b0[q]);
#pragma unroll
      for (int j = 0; j < TH; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0[j]), bq, acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) b0[q] = __builtin_amdgcn_raw_buffer_load_b128(r_w, w_voff, w_next + q * 1024, 0);
    // odd step: fragments a1 / b1; read a0 for the next even step
#pragma unroll
    for (int j = 0; j < TH; ++j) a0[j] = *reinterpret_cast<const u32x4*>(lds + a_off + j * 2 * G::PITCH);
    advance();
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bf16x8 bq = __builtin_bit_cast(bf16x8, b1[q]);
#pragma unroll
      for (int j = 0; j < TH; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1[j]), bq, acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) b1[q] = __builtin_amdgcn_raw_buffer_load_b128(r_w, w_voff, w_next + 3072 + q * 1024, 0);
  }

  // ---- epilogue: bias + ReLU, through LDS, whole pixel rows out ------------------------------------------------------------------------
  __syncthreads();   // every wave is done with the patch
  {
    const int co = cb * NCO + wave * 16 + i;
    const float bias = p.bias ? p.bias[co] : 0.f;
    float* tile = reinterpret_cast<float*>(lds);   // [128 pixels][64 channels]
#pragma unroll
    for (int j = 0; j < TH; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[j][r] + bias;
        if (p.relu) v = fmaxf(v, 0.f);
        tile[(j * TW + 4 * g + r) * NCO + wave * 16 + i] = v;   // C/D map: row (pixel) = 4 (lane >> 4) + r, column (channel) = lane & 15
      }
  }
  __syncthreads();
  if constexpr (POOL) {
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int Hqp = p.Hq + 2 * p.pool_border, Wqp = p.Wq + 2 * p.pool_border;
    const float4* tile4 = reinterpret_cast<const float4*>(lds);   // [128 pixels][16 groups of 4 channels]
    for (int item = tid; item < (TH / 2 + 1) * (TW / 2 + 1) * 16; item += 256) {
      const int part = item & 15, pp = item >> 4;
      const int ppy = pp / (TW / 2 + 1), ppx = pp - ppy * (TW / 2 + 1);
      const int PY = ty * (TH / 2) + ppy, PX = tx * (TW / 2) + ppx;
      if (PY >= p.Hq || PX >= p.Wq) continue;
      // the part of the window [2P-1, 2P+1] that is inside the image ...
      const int wr0 = max(2 * PY - 1, 0), wr1 = min(2 * PY + 1, p.Ho - 1), wc0 = max(2 * PX - 1, 0), wc1 = min(2 * PX + 1, p.Wo - 1);
      // ... and inside this tile
      const int r0 = max(wr0, oy0), r1 = min(wr1, oy0 + TH - 1), c0 = max(wc0, ox0), c1 = min(wc1, ox0 + TW - 1);
      if (r0 > r1 || c0 > c1) continue;
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = r0; r <= r1; ++r)
        for (int c = c0; c <= c1; ++c) {
          const float4 v = tile4[((r - oy0) * TW + (c - ox0)) * 16 + part];
          m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
      float* dst = p.ypool + ((((size_t)n * Hqp + PY + p.pool_border) * Wqp + PX + p.pool_border) * p.Cout + cb * NCO + part * 4);
      if (r0 == wr0 && r1 == wr1 && c0 == wc0 && c1 == wc1) {
        *reinterpret_cast<float4*>(dst) = m;
      } else {   // (& 0x7FFFFFFF: a -0.0 out of the ReLU must not outrank every positive value)
        unsigned* du = reinterpret_cast<unsigned*>(dst);
        atomicMax(du, __float_as_uint(m.x) & 0x7FFFFFFFu); atomicMax(du + 1, __float_as_uint(m.y) & 0x7FFFFFFFu);
        atomicMax(du + 2, __float_as_uint(m.z) & 0x7FFFFFFFu); atomicMax(du + 3, __float_as_uint(m.w) & 0x7FFFFFFFu);
      }
    }
  }
  if (!POOL || p.y != nullptr) {
    // (one resource per image: the batch may exceed 4 GB -- coarse launches of 1000 rows do)
    const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)n * (p.out_bytes / 4)), 0, p.out_bytes, 0x00020000);
    const int oy0 = ty * TH, ox0 = tx * TW;
#pragma unroll
    for (int k = 0; k < (TH * TW * NCO / 4) / 256; ++k) {
      const int c = tid + 256 * k;          // 16-byte chunk of the tile: pixel c / 16, channels 4 (c % 16) ..
      const int pix = c >> 4, part = c & 15;
      const int oy = oy0 + (pix >> 4), ox = ox0 + (pix & 15);
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds + c * 16);
      const int off = ((oy + p.out_border) * p.Wop + ox + p.out_border) * p.Cout + cb * NCO + part * 4;
      const unsigned voff = (oy < p.Ho && ox < p.Wo) ? (unsigned)off * 4u : 0xFFFFFFF0u;   // outside the image: dropped by the range check
      __builtin_amdgcn_raw_buffer_store_b128(v, r_y, (int)voff, 0, 0);
    }
  }
}

// clears the pooled positions that more than one tile contributes to (every pooled row 4k and pooled column 8k: the windows that straddle
// the 8 x 16 stem tiles), one thread per (such position, 4 channels): the grid enumerates only those positions -- first the rows 4k in
// full, then column 8k of the other rows (a third of the pooled map; round 4 launched a thread per position and returned from two thirds)
__host__ __device__ inline int pool_zero_positions(int Hq, int Wq) {
  const int rows4 = (Hq + TH / 2 - 1) / (TH / 2), cols8 = (Wq + TW / 2 - 1) / (TW / 2);
  return rows4 * Wq + (Hq - rows4) * cols8;
}
__global__ __launch_bounds__(256) void pool_zero_kernel(float* __restrict__ y, int N, int Hq, int Wq, int C, int border) {
  const int c4n = C / 4;
  const int P = pool_zero_positions(Hq, Wq);
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * P * c4n) return;
  const int c4 = (int)(idx % c4n);
  long t = idx / c4n;
  const int pos = (int)(t % P), n = (int)(t / P);
  const int rows4 = (Hq + TH / 2 - 1) / (TH / 2), cols8 = (Wq + TW / 2 - 1) / (TW / 2);
  int py, px;
  if (pos < rows4 * Wq) {
    py = (pos / Wq) * (TH / 2);
    px = pos % Wq;
  } else {   // the k-th row that is not a multiple of 4: k + k / 3 + 1
    const int q = pos - rows4 * Wq, k = q / cols8;
    py = k + k / (TH / 2 - 1) + 1;
    px = (q % cols8) * (TW / 2);
  }
  *reinterpret_cast<float4*>(y + ((((size_t)n * (Hq + 2 * border) + py + border) * (Wq + 2 * border) + px + border) * C + c4 * 4)) =
      make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int KS, int Q, bool POOL = false, bool SP = false>
int launch(const Params& p, hipStream_t s, double flops, double bytes, const char* name) {
  // executed on the bf16 pipe: every 16x16x32 MFMA of every step, three weight pieces (this counts the record padding, the step padding
  // and the nine products of the fp32-kind channels as what they cost)
  const double executed = 2.0 * (double)p.N * p.tiles_y * p.tiles_x * (TH * TW) * p.Cout * (double)p.n_steps * 32.0 * 3.0;
  using G = Geo<KS, Q>;
  static int attr_dev = -1;
  int dev = 0;
  MP_CHECK_HIP(hipGetDevice(&dev));
  if (attr_dev != dev) {
    MP_CHECK_HIP(hipFuncSetAttribute((const void*)conv_stem_bf16x3<KS, Q, POOL, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS));
    attr_dev = dev;
  }
  if (POOL) {
    ProfScope prof0("pool_zero", 0.0, 16.0 * p.N * (p.Hq / 4 + 1) * p.Wq * p.Cout, s);
    const long total = (long)p.N * pool_zero_positions(p.Hq, p.Wq) * (p.Cout / 4);
    hipLaunchKernelGGL(pool_zero_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p.ypool, p.N, p.Hq, p.Wq, p.Cout, p.pool_border);
  }
  ProfScope prof(name, flops, bytes, s, executed, 2500.0);
  hipLaunchKernelGGL((conv_stem_bf16x3<KS, Q, POOL, SP>), dim3((unsigned)((long)p.N * p.tiles_y * p.tiles_x * p.n_cb)), dim3(256), G::LDS, s, p);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

inline int n_steps(int KS, int Q) { return ((KS * KS * Q + 3) / 4 + 1) / 2 * 2; }

}  // namespace stem
}  // namespace mp

using namespace mp;

extern "C" int mp_xrec_elements(int n_f32, int n_u8) {
  if (n_f32 < 0 || n_u8 < 0 || n_f32 + n_u8 <= 0) return 0;
  return (3 * n_f32 + n_u8 + 7) / 8 * 8;
}

extern "C" int mp_conv_stem_supported(int KS, int n_f32, int n_u8) {
  const int R = mp_xrec_elements(n_f32, n_u8);
  return (KS == 7 || KS == 5) && R >= 16 && R <= 48 ? 1 : 0;
}

extern "C" size_t mp_conv_stem_packed_bytes(int KS, int n_f32, int n_u8, int Cout) {
  const int Q = mp_xrec_elements(n_f32, n_u8) / 8;
  return (size_t)(Cout / stem::NCO) * 4 * stem::n_steps(KS, Q) * 3072;
}

// exact truncation split of a float into three bf16 pieces (hi, mid, lo): v == hi + mid + lo
static void split3(float v, unsigned short out[3]) {
  unsigned vb, rb, qb;
  memcpy(&vb, &v, 4);
  const unsigned h = vb & 0xFFFF0000u;
  float hf; memcpy(&hf, &h, 4);
  const float r = v - hf;
  memcpy(&rb, &r, 4);
  const unsigned m = rb & 0xFFFF0000u;
  float mf; memcpy(&mf, &m, 4);
  const float q = r - mf;
  memcpy(&qb, &q, 4);
  out[0] = (unsigned short)(h >> 16); out[1] = (unsigned short)(m >> 16); out[2] = (unsigned short)(qb >> 16);
}

// packed[cb][wave][step][piece][lane][e]: lane = (cout = cb*64 + wave*16 + (lane & 15), slice group lane >> 4), slice s = 4 step + group
// = (kh, kw, chunk q of the pixel record), element e of the chunk = record slot 8q + e -> input channel; piece = w1 | w2 | w3 of
// w * scale (* 1/255 for the integer channels, one rounding of the exact product)
extern "C" int mp_conv_stem_pack_weights(const float* w, int Cout, int Cin, int KS, int n_f32, const float* scale, void* packed) {
  MP_REQUIRE(n_f32 >= 0 && n_f32 <= 32 && n_f32 <= Cin, "mp_conv_stem_pack_weights: bad n_f32");
  return mp_conv_stem_pack_weights_mask(w, Cout, Cin, KS, n_f32 >= 32 ? 0xFFFFFFFFu : (1u << n_f32) - 1u, scale, packed);
}

// the general record: input channel c is fp32-kind (three pieces) iff bit c of f32_mask is set -- e.g. an RGBD refiner's depth channels
// (observation depth + one rendered depth per view) -- and sits, in channel order, in front of the integer channels:
//   [x1,x2,x3 of every fp32-kind channel | k of every integer channel | zero padding]
// q_walk = record chunks per pixel the kernel's slice walk visits: Q for the dense walk; the chunks that hold fp32-kind pieces for the
// background-tile walk (mp_conv_stem_sparse_*)
static int pack_weights_walk(const float* w, int Cout, int Cin, int KS, uint32_t f32_mask, const float* scale, void* packed, int q_walk);

extern "C" int mp_conv_stem_pack_weights_mask(const float* w, int Cout, int Cin, int KS, uint32_t f32_mask, const float* scale, void* packed) {
  return pack_weights_walk(w, Cout, Cin, KS, f32_mask, scale, packed, 0);
}

// Background-tile form: a workgroup whose input patch holds no rendered geometry (all integer channels 0) walks only the record chunks
// that hold fp32-kind pieces: q_use = ceil(3 n_f32 / 8) of the Q chunks.  Defined only when the fp32-kind channels are the leading ones
// (no depth channels: a normalised background depth is not 0); 0 = no such form (nothing to skip, or not applicable).
extern "C" int mp_conv_stem_sparse_chunks(int KS, int n_f32, int n_u8) {
  if (!mp_conv_stem_supported(KS, n_f32, n_u8) || n_f32 <= 0) return 0;
  const int Q = mp_xrec_elements(n_f32, n_u8) / 8, q_use = (3 * n_f32 + 7) / 8;
  if (q_use >= Q) return 0;
  // the short walk runs 4 n_steps(KS, q_use) + 4 slices (one group ahead) over KS * KS * q_use real ones and ONE zero row of KS * q_use
  // slices behind them: a walk that needs more padding than that row (q_use = 1, i.e. n_f32 = 1 or 2: 11 > 7 / 5) would step past it
  // and read LDS outside the patch -- no background-tile form for such records (the pipeline's records have q_use = 2)
  const int pad = 4 * stem::n_steps(KS, q_use) + 4 - KS * KS * q_use;
  return pad <= KS * q_use ? q_use : 0;
}
extern "C" size_t mp_conv_stem_sparse_packed_bytes(int KS, int n_f32, int n_u8, int Cout) {
  const int q_use = mp_conv_stem_sparse_chunks(KS, n_f32, n_u8);
  return q_use ? (size_t)(Cout / stem::NCO) * 4 * stem::n_steps(KS, q_use) * 3072 : 0;
}
extern "C" int mp_conv_stem_pack_weights_sparse(const float* w, int Cout, int Cin, int KS, int n_f32, const float* scale, void* packed) {
  const int q_use = mp_conv_stem_sparse_chunks(KS, n_f32, Cin - n_f32);
  MP_REQUIRE(q_use > 0, "mp_conv_stem_pack_weights_sparse: this record has no background-tile form");
  return pack_weights_walk(w, Cout, Cin, KS, n_f32 >= 32 ? 0xFFFFFFFFu : (1u << n_f32) - 1u, scale, packed, q_use);
}

static int pack_weights_walk(const float* w, int Cout, int Cin, int KS, uint32_t f32_mask, const float* scale, void* packed, int q_walk) {
  MP_REQUIRE(Cin >= 1 && Cin <= 32 && (Cin == 32 || (f32_mask >> Cin) == 0u), "mp_conv_stem_pack_weights_mask: mask names channels >= Cin (<= 32 input channels)");
  const int n_f32 = __builtin_popcount(f32_mask), n_u8 = Cin - n_f32;
  MP_REQUIRE(w && packed && n_u8 >= 0 && Cout % stem::NCO == 0 && mp_conv_stem_supported(KS, n_f32, n_u8),
             "mp_conv_stem_pack_weights: bad arguments (Cout %% 64, KS 5 | 7, 16 <= record <= 48 elements)");
  int ch_of_f32[32], ch_of_u8[32];
  for (int c = 0, a = 0, b = 0; c < Cin; ++c) {
    if ((f32_mask >> c) & 1u) ch_of_f32[a++] = c; else ch_of_u8[b++] = c;
  }
  const int Q = q_walk > 0 ? q_walk : mp_xrec_elements(n_f32, n_u8) / 8;   // chunks walked per pixel (slot 8 q + e names the same channel either way)
  const int T = stem::n_steps(KS, Q), S = KS * KS * Q;
  unsigned short* out = (unsigned short*)packed;
  memset(out, 0, (size_t)(Cout / stem::NCO) * 4 * T * 3072);
  for (int co = 0; co < Cout; ++co) {
    const int cb = co / stem::NCO, wave = (co % stem::NCO) / 16, i = co % 16;
    const double sc = scale ? (double)scale[co] : 1.0;
    for (int s = 0; s < S; ++s) {
      const int t = s / 4, g = s % 4;
      const int kh = s / (KS * Q), rr = s % (KS * Q), kw = rr / Q, q = rr % Q;
      for (int e = 0; e < 8; ++e) {
        const int slot = 8 * q + e;
        int ch;
        double f;
        if (slot < 3 * n_f32) { ch = ch_of_f32[slot / 3]; f = sc; }
        else if (slot < 3 * n_f32 + n_u8) { ch = ch_of_u8[slot - 3 * n_f32]; f = sc / 255.0; }
        else continue;
        const float v = (float)((double)w[(((size_t)co * Cin + ch) * KS + kh) * KS + kw] * f);
        unsigned short pc[3];
        split3(v, pc);
        for (int piece = 0; piece < 3; ++piece)
          out[((((size_t)(cb * 4 + wave) * T + t) * 3 + piece) * 64 + (g * 16 + i)) * 8 + e] = pc[piece];
      }
    }
  }
  return MP_OK;
}

// d_x = xrec tensor (bf16 records, padded NHWC with border in_border); the other fields as mp_conv2d_nhwc; KH = KW in {5, 7},
// stride 2, Cout % 64 == 0, no residual / second output
static int stem_xrec_impl(const mp_conv_desc* d, const void* d_packed, int n_f32, float* d_ypool, int pool_border, mp_stream stream,
                          const void* d_packed_sparse = nullptr, const unsigned char* d_tile_flags = nullptr) {
  MP_REQUIRE(d && d->d_x && d_packed && (d->d_y || d_ypool), "mp_conv_stem_xrec: null pointer");
  MP_REQUIRE(!d_ypool || (d->relu && pool_border >= 0), "mp_conv_stem_xrec_pool: the fused max pool needs the ReLU (its atomicMax orders non-negative floats)");
  const int n_u8 = d->c_real - n_f32;
  MP_REQUIRE(d->KH == d->KW && d->stride == 2 && d->Cout % stem::NCO == 0 && !d->d_residual && !d->d_y_act && d->in_border >= d->pad &&
                 mp_conv_stem_supported(d->KH, n_f32, n_u8),
             "mp_conv_stem_xrec: unsupported layer (square 5x5 / 7x7, stride 2, Cout %% 64, c_real = real channels, 16 <= record <= 48 elements)");
  const int R = mp_xrec_elements(n_f32, n_u8), Q = R / 8;
  stem::Params p;
  p.x = (const unsigned char*)d->d_x; p.w = (const unsigned char*)d_packed; p.bias = d->d_bias; p.y = d->d_y;
  p.N = d->N;
  p.Ho = (d->H + 2 * d->pad - d->KH) / 2 + 1; p.Wo = (d->W + 2 * d->pad - d->KW) / 2 + 1;
  p.Hp = d->H + 2 * d->in_border; p.Wp = d->W + 2 * d->in_border; p.in_off = d->in_border - d->pad;
  p.Cout = d->Cout; p.Hop = p.Ho + 2 * d->out_border; p.Wop = p.Wo + 2 * d->out_border; p.out_border = d->out_border;
  p.tiles_x = ceil_div(p.Wo, stem::TW); p.tiles_y = ceil_div(p.Ho, stem::TH); p.n_cb = d->Cout / stem::NCO;
  p.n_steps = stem::n_steps(d->KH, Q);
  p.relu = d->relu;
  const long img = (long)p.Hp * p.Wp * 16 * Q, outb = (long)p.Hop * p.Wop * d->Cout * 4;   // per image: the batch index is applied in 64 bits
  MP_REQUIRE(img < (1L << 31) && outb < (1L << 31) && (long)d->N * p.tiles_y * p.tiles_x * p.n_cb < (1L << 31), "mp_conv_stem_xrec: image too large for 32-bit offsets");
  p.img_bytes = (unsigned)img; p.out_bytes = (unsigned)outb;
  p.ypool = d_ypool; p.pool_border = pool_border;
  p.Hq = (p.Ho + 2 - 3) / 2 + 1; p.Wq = (p.Wo + 2 - 3) / 2 + 1;
  p.H = d->H; p.W = d->W; p.pad = d->pad;
  p.tile_flags = nullptr; p.w_sparse = nullptr; p.fl_tx = p.fl_ty = 0; p.q_use = Q; p.n_steps_sparse = p.n_steps;
  const int q_use = mp_conv_stem_sparse_chunks(d->KH, n_f32, n_u8);
  const bool sp = d_packed_sparse && d_tile_flags && q_use > 0;
  p.count_bg = 0;
  if (sp) {   // the rasteriser's job flags of the launch that wrote these records: one byte per (image, 8x8-pixel tile)
    p.count_bg = mp_profile_active();
    p.tile_flags = d_tile_flags; p.w_sparse = (const unsigned char*)d_packed_sparse;
    p.fl_tx = ceil_div(d->W, 8); p.fl_ty = ceil_div(d->H, 8); p.q_use = q_use; p.n_steps_sparse = stem::n_steps(d->KH, q_use);
  }
  const double M = (double)d->N * p.Ho * p.Wo;
  const double flops = 2.0 * M * d->Cout * d->KH * d->KW * d->c_real;
  const double bytes = (double)d->N * img + (d->d_y ? 4.0 * M * d->Cout : 0.0) + (d_ypool ? 4.0 * (double)d->N * p.Hq * p.Wq * d->Cout : 0.0) +
                       (double)mp_conv_stem_packed_bytes(d->KH, n_f32, n_u8, d->Cout);
  hipStream_t s = (hipStream_t)stream;
#define MP_STEM_GO(KSV, QV)                                                                                                          \
  if (d->KH == KSV && Q == QV) {                                                                                                   \
    if (sp)                                                                                                                        \
      return d_ypool ? stem::launch<KSV, QV, true, true>(p, s, flops, bytes, "conv_stem_bf16x3+maxpool<" #KSV "x" #KSV ",Q" #QV ">")   \
                     : stem::launch<KSV, QV, false, true>(p, s, flops, bytes, "conv_stem_bf16x3<" #KSV "x" #KSV ",Q" #QV ">");       \
    return d_ypool ? stem::launch<KSV, QV, true>(p, s, flops, bytes, "conv_stem_bf16x3+maxpool<" #KSV "x" #KSV ",Q" #QV ">")         \
                   : stem::launch<KSV, QV, false>(p, s, flops, bytes, "conv_stem_bf16x3<" #KSV "x" #KSV ",Q" #QV ">");               \
  }
  MP_STEM_GO(7, 2) MP_STEM_GO(7, 3) MP_STEM_GO(7, 4) MP_STEM_GO(7, 5) MP_STEM_GO(7, 6)
  MP_STEM_GO(5, 2) MP_STEM_GO(5, 3) MP_STEM_GO(5, 4) MP_STEM_GO(5, 5) MP_STEM_GO(5, 6)
#undef MP_STEM_GO
  set_error("mp_conv_stem_xrec: no instance for %dx%d, record of %d elements", d->KH, d->KW, R);
  return MP_ERR_INVALID;
}

extern "C" int mp_conv_stem_xrec(const mp_conv_desc* d, const void* d_packed, int n_f32, mp_stream stream) {
  MP_REQUIRE(d && d->d_y, "mp_conv_stem_xrec: null output");
  return stem_xrec_impl(d, d_packed, n_f32, nullptr, 0, stream);
}

// the same + the 3x3 / stride-2 / pad-1 max pool of the stem output (models/torchvision_resnet.py:216) fused into the epilogue: d_ypool =
// padded NHWC [N][(Ho+1)/2 ...] with border pool_border (zero border, interior fully written); desc->d_y may be NULL (no stem map)
extern "C" int mp_conv_stem_xrec_pool(const mp_conv_desc* d, const void* d_packed, int n_f32, float* d_ypool, int pool_border, mp_stream stream) {
  MP_REQUIRE(d_ypool, "mp_conv_stem_xrec_pool: null pooled output");
  return stem_xrec_impl(d, d_packed, n_f32, d_ypool, pool_border, stream);
}

// mp_conv_stem_xrec[_pool] with the background-tile form: d_packed_sparse = the blob of mp_conv_stem_pack_weights_sparse, d_tile_flags =
// the job flags of the rasteriser launch that wrote the records (mp_raster_job_flags: one byte per (image, 8x8-pixel tile), 0 = no view
// reaches the tile).  d_ypool may be NULL (then desc->d_y must be set).
extern "C" int mp_conv_stem_xrec_sparse(const mp_conv_desc* d, const void* d_packed, const void* d_packed_sparse, int n_f32,
                                        const unsigned char* d_tile_flags, float* d_ypool, int pool_border, mp_stream stream) {
  MP_REQUIRE(d && (d->d_y || d_ypool), "mp_conv_stem_xrec_sparse: null output");
  return stem_xrec_impl(d, d_packed, n_f32, d_ypool, pool_border, stream, d_packed_sparse, d_tile_flags);
}

// background-tile statistics of the SP launches issued while the event profiler was active: workgroups that took the short walk, all
// workgroups of those launches (synchronises the device)
extern "C" int mp_conv_stem_bg_stats(double* background_wgs, double* total_wgs, int reset) {
  unsigned long long h[2] = {0, 0};
  MP_CHECK_HIP(hipDeviceSynchronize());
  MP_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(stem::g_stem_bg), sizeof(h)));
  if (background_wgs) *background_wgs = (double)h[0];
  if (total_wgs) *total_wgs = (double)h[1];
  if (reset) {
    const unsigned long long z[2] = {0, 0};
    MP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(stem::g_stem_bg), z, sizeof(z)));
  }
  return MP_OK;
}
