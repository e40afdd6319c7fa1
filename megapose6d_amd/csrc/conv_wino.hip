// conv_wino.hip -- 3x3 / stride-1 convolution as a FUSED Winograd F(2x2, 3x3) on the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces the cuDNN 3x3 convolutions of the residual stages behind `self.backbone(x)` (reference:
// src/megapose/models/pose_rigid.py:323; layers src/megapose/models/torchvision_resnet.py:74-120 (BasicBlock conv1/conv2),
// src/megapose/models/wide_resnet.py:29-56) -- 73 % of a refiner row's FLOPs.  The direct implicit GEMM (conv.hip) is MFMA-bound
// at 0.8-0.9 of the fp32 matrix peak; the only lever left is fewer multiplications: Y = A^T [ (G g G^T) o (B^T d B) ] A computes a
// 2x2 output tile from a 4x4 input patch with 16 instead of 36 multiplications per (cin, cout) pair (2.25x).
//
// Everything is fused into ONE kernel (an un-fused version loses: the 4x transformed tensor costs more HBM time than the MFMA time
// saved at 576 rows):
//   * workgroup = 64 output tiles (2x2 pixels each) x 64 output channels, 4 waves, ONE workgroup per CU; the 16 frequency planes
//     x 64 tiles x 64 channels = 65 536 accumulators fill the CU's accumulator registers (256 per lane).  Wave w owns the
//     frequency points of row w of the 4x4 grid (f = 4w .. 4w+3), for all tiles and channels: 4 x (2 x 2) MFMA tiles of 32x32.
//   * K loop over the input channels, 8 per step.  The pre-transformed weights U = G g G^T are packed on the host in MFMA
//     fragment order per (chunk, f) and go from L2 STRAIGHT into registers (each wave needs only its own f's: no LDS, no
//     redundancy); the input transform V = B^T d B is computed by half of the threads per step (they alternate) from 4x4 patches
//     read with 16-byte buffer loads two steps ahead, and handed to the MFMA waves through a double-buffered LDS tile
//     V[f][tile][8 + 4 pad] (conflict-free b128 fragment reads); one barrier per step, 64 MFMAs per wave between barriers.
//   * epilogue: each wave applies the row half of the output transform to its own accumulators ((m A)[w][0..1]), the four waves
//     exchange those through LDS, and every thread finishes A^T (.) for 4 channels of one tile: bias (folded BN) + residual + ReLU
//     (+ the second pre-activated output of the WideResNet blocks), 16-byte loads / stores.
// Numerics: fp32 throughout; F(2x2, 3x3) in fp32 is ~2x the rounding error of the direct sum (3-6e-7 of the activation scale
// on ResNet-shaped data) -- far inside the 1e-4 parity tolerance; results are deterministic.
// Roofline: MFMA-bound; it EXECUTES 16/36 of the direct algorithm's FLOPs, so its algorithmic (direct-equivalent) rate can exceed
// the matrix peak -- the profiler rows carry BOTH figures (mp_profile_query_ex): bench.py's `roofline.frac` is the executed rate / peak
// (<= 1 by construction), the direct-equivalent rate sits next to it.
#include <cstdlib>
#include <vector>

#include "wino_common.h"

namespace mp {


__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_wino_f32(WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;
  int* tile_tab = (int*)(smem + 4 * 2 * WT * WCOUT);   // [64][2]: output element offset of pixel (2ty, 2tx) (-1: no such tile), validity bits

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keeps the weight resource / role branches scalar
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int cb = wg % p.n_cblocks;      // channel block fastest: the workgroups that share an input tile set run together
  const int tg = wg / p.n_cblocks;
  const int tile0 = tg * WT;

  // ---- tile table (epilogue) -----------------------------------------------------------------------------------------------
  if (tid < WT) {
    const int t = tile0 + tid;
    int off = -1, bits = 0;
    if (t < p.n_tiles) {
      const int tx = t % p.tiles_x, r = t / p.tiles_x;
      const int ty = r % p.tiles_y, n = r / p.tiles_y;
      off = ((n * p.Hop + 2 * ty + p.out_border) * p.Wop + 2 * tx + p.out_border) * p.Cout;
      bits = ((2 * ty + 1 < p.Ho) ? 1 : 0) | ((2 * tx + 1 < p.Wo) ? 2 : 0);
    }
    tile_tab[2 * tid] = off;
    tile_tab[2 * tid + 1] = bits;
  }

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, -1, 0x00020000);
  const int row_bytes = p.Wp * p.C * 4, pix_bytes = p.C * 4;
  // weights: this wave's slice of chunk ch = u + (((cb * n_chunks + ch) * 16 + 4 * wave) * 2) * 256 floats, 8 KB contiguous
  const __amdgpu_buffer_rsrc_t u_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.u + ((size_t)cb * p.n_chunks * 16 + 4 * wave) * 512), 0, -1, 0x00020000);
  const int u_voff = lane * 16;
  constexpr int U_CHUNK_BYTES = 16 * 2 * 256 * 4;   // 32 KB per chunk (all 16 f)

  float4 patch[4][4];
  float4 U0[4][2], U1[4][2];   // weight fragments of the even / odd 8-channel half-steps

  // per step: 16 input channels.  Every thread transforms ONE (tile, 4-channel group) pair per step; no roles, no branches in the loop.
  const int ptile = tid >> 2, pc4 = tid & 3;
  int x_voff;
  {
    int t = tile0 + ptile;
    t = t < p.n_tiles ? t : p.n_tiles - 1;
    const int tx = t % p.tiles_x, r = t / p.tiles_x;
    const int ty = r % p.tiles_y, n = r / p.tiles_y;
    const size_t pix = ((size_t)n * p.Hp + (size_t)(2 * ty + p.in_off)) * p.Wp + (size_t)(2 * tx + p.in_off);
    x_voff = (int)((pix * p.C + pc4 * 4) * sizeof(float));   // < 2^31: checked on the host
  }
  // V[stage][f][tile][16 floats], the four 16-byte slots of a row XOR-swizzled by (tile >> 2) & 3: conflict-free for the
  // transform's ds_write_b128 (lanes = 16 tiles x 4 slots) and for the fragment ds_read_b128 (lanes = 32 tiles x 2 k-halves)
  float* vw = Vs + ptile * WCK + ((pc4 ^ ((ptile >> 2) & 3)) * 4);

#define WINO_LOAD_PATCH(ST)                                                                            \
  {                                                                                                    \
    const int cs_ = (ST) * (WCK * 4);                                                                  \
    _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                      \
      _Pragma("unroll") for (int b = 0; b < 4; ++b) patch[a][b] = buf4(x_rsrc, x_voff, cs_ + a * row_bytes + b * pix_bytes); \
  }
#define WINO_LOAD_U(DST, C8)                                                                           \
  {                                                                                                    \
    const int us_ = (C8) * U_CHUNK_BYTES;                                                              \
    _Pragma("unroll") for (int fi = 0; fi < 4; ++fi)                                                   \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) DST[fi][j] = buf4(u_rsrc, u_voff, us_ + (fi * 2 + j) * 1024); \
  }
// B^T d B of the thread's 4x4 patch (4 channels at once), written as 16 float4 into stage BUF
#define WINO_TRANSFORM(BUF)                                                                            \
  {                                                                                                    \
    float* vw_ = vw + (BUF) * WV_STAGE;                                                                \
    float4 t_[4][4];                                                                                   \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                    \
      t_[0][b] = f4sub(patch[0][b], patch[2][b]);                                                      \
      t_[1][b] = f4add(patch[1][b], patch[2][b]);                                                      \
      t_[2][b] = f4sub(patch[2][b], patch[1][b]);                                                      \
      t_[3][b] = f4sub(patch[1][b], patch[3][b]);                                                      \
    }                                                                                                  \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                    \
      *reinterpret_cast<float4*>(vw_ + (a * 4 + 0) * (WT * WCK)) = f4sub(t_[a][0], t_[a][2]);          \
      *reinterpret_cast<float4*>(vw_ + (a * 4 + 1) * (WT * WCK)) = f4add(t_[a][1], t_[a][2]);          \
      *reinterpret_cast<float4*>(vw_ + (a * 4 + 2) * (WT * WCK)) = f4sub(t_[a][2], t_[a][1]);          \
      *reinterpret_cast<float4*>(vw_ + (a * 4 + 3) * (WT * WCK)) = f4sub(t_[a][1], t_[a][3]);          \
    }                                                                                                  \
  }
  auto f4add = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
  auto f4sub = [](float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };

  f32x16 acc[4][2][2];
#pragma unroll
  for (int fi = 0; fi < 4; ++fi)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fi][i][j][r] = 0.f;

  // ---- prologue: V of step 0, patch of step 1 in flight, U of half-step 0 ---------------------------------------------------
  const int ns = p.n_steps;
  WINO_LOAD_U(U0, 0)
  WINO_LOAD_PATCH(0)
  WINO_TRANSFORM(0)
  WINO_LOAD_PATCH(ns > 1 ? 1 : 0)
  __syncthreads();

  // fragment read position: lane (tile row = lane & 31 (+ 32), k half = lane >> 5), slot swizzle as above
  const int fsw = ((lane & 31) >> 2) & 3;
  const float* vr = Vs + ((4 * wave) * WT + (lane & 31)) * WCK;
  const int fo0 = (((lane >> 5)) ^ fsw) * 4, fo1 = ((2 + (lane >> 5)) ^ fsw) * 4;   // float offsets of the two 8-channel halves
#define WINO_STEP(AF, UU, Q)                                                                            \
  acc[fi][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[0].Q, UU[fi][0].Q, acc[fi][0][0], 0, 0, 0);   \
  acc[fi][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[0].Q, UU[fi][1].Q, acc[fi][0][1], 0, 0, 0);   \
  acc[fi][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[1].Q, UU[fi][0].Q, acc[fi][1][0], 0, 0, 0);   \
  acc[fi][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[1].Q, UU[fi][1].Q, acc[fi][1][1], 0, 0, 0);
#define WINO_READ_FRAGS(AF, FI, FO)                                                                     \
  AF[0] = *reinterpret_cast<const float4*>(vb + (FI) * (WT * WCK) + (FO));                              \
  AF[1] = *reinterpret_cast<const float4*>(vb + (FI) * (WT * WCK) + 32 * WCK + (FO));
#define WINO_SB __builtin_amdgcn_sched_barrier(0);
// transform slice (a, col): one of the 16 frequency planes of the thread's patch -> one ds_write_b128.  Slice (a, 0) also forms the
// row combination t[a][0..3] of the patch (B^T d); the column combination (. B) is 4 VALU per plane.
#define WINO_SLICE(A, COL)                                                                              \
  {                                                                                                     \
    if ((COL) == 0) {                                                                                   \
      _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                     \
        trow[b] = (A) == 0 ? f4sub(patch[0][b], patch[2][b]) : (A) == 1 ? f4add(patch[1][b], patch[2][b])  \
                : (A) == 2 ? f4sub(patch[2][b], patch[1][b]) : f4sub(patch[1][b], patch[3][b]);         \
    }                                                                                                   \
    const float4 v_ = (COL) == 0 ? f4sub(trow[0], trow[2]) : (COL) == 1 ? f4add(trow[1], trow[2])       \
                    : (COL) == 2 ? f4sub(trow[2], trow[1]) : f4sub(trow[1], trow[3]);                   \
    *reinterpret_cast<float4*>(vwn + ((A) * 4 + (COL)) * (WT * WCK)) = v_;                              \
  }
  float4 A0[2], A1[2], trow[4];
  for (int st = 0; st < ns; ++st) {
    const int buf = st & 1;
    const float* vb = vr + buf * WV_STAGE;
    float* vwn = vw + (buf ^ 1) * WV_STAGE;
    const int cs_patch = (st + 2 < ns ? st + 2 : ns - 1) * (WCK * 4);
    const int us_next = (st + 1 < ns ? 2 * st + 2 : 2 * st) * U_CHUNK_BYTES;
    // Hand-placed step (sched_barrier pins the order; in-order issue lets ~14 issue slots ride under every 64-cycle MFMA):
    //   half 0 (channels 0-7, weights U0): after every 4 MFMAs one transform slice of the NEXT step's patch (8-20 VALU + 1 ds_write_b128)
    //   half 1 (channels 8-15, weights U1): after every 4 MFMAs one patch request of the step after next (+ the next U0 fragments)
    // Fragments are read one group ahead.  The last step harmlessly transforms / re-loads its own data into idle buffers.
    WINO_LOAD_U(U1, 2 * st + 1)
    WINO_READ_FRAGS(A0, 0, fo0)
    WINO_SB
#define WINO_GROUP_H0(FI, ACUR, ANEXT, NFI, NFO)                                                        \
  {                                                                                                     \
    constexpr int fi = FI;                                                                              \
    WINO_READ_FRAGS(ANEXT, NFI, NFO)                                                                    \
    WINO_STEP(ACUR, U0, x) WINO_SB WINO_SLICE(FI, 0) WINO_SB                                            \
    WINO_STEP(ACUR, U0, y) WINO_SB WINO_SLICE(FI, 1) WINO_SB                                            \
    WINO_STEP(ACUR, U0, z) WINO_SB WINO_SLICE(FI, 2) WINO_SB                                            \
    WINO_STEP(ACUR, U0, w) WINO_SB WINO_SLICE(FI, 3) WINO_SB                                            \
  }
    WINO_GROUP_H0(0, A0, A1, 1, fo0)
    WINO_GROUP_H0(1, A1, A0, 2, fo0)
    WINO_GROUP_H0(2, A0, A1, 3, fo0)
    WINO_GROUP_H0(3, A1, A0, 0, fo1)
#define WINO_PLOAD(A, B) patch[A][B] = buf4(x_rsrc, x_voff, cs_patch + (A) * row_bytes + (B) * pix_bytes);
#define WINO_ULOAD(FI, J) U0[FI][J] = buf4(u_rsrc, u_voff, us_next + ((FI) * 2 + (J)) * 1024);
#define WINO_GROUP_H1(FI, ACUR, ANEXT, NFI, LAST)                                                       \
  {                                                                                                     \
    constexpr int fi = FI;                                                                              \
    if (!(LAST)) { WINO_READ_FRAGS(ANEXT, NFI, fo1) }                                                   \
    WINO_STEP(ACUR, U1, x) WINO_SB WINO_PLOAD(FI, 0) WINO_ULOAD(FI, 0) WINO_SB                          \
    WINO_STEP(ACUR, U1, y) WINO_SB WINO_PLOAD(FI, 1) WINO_ULOAD(FI, 1) WINO_SB                          \
    WINO_STEP(ACUR, U1, z) WINO_SB WINO_PLOAD(FI, 2) WINO_SB                                            \
    WINO_STEP(ACUR, U1, w) WINO_SB WINO_PLOAD(FI, 3) WINO_SB                                            \
  }
    WINO_GROUP_H1(0, A0, A1, 1, false)
    WINO_GROUP_H1(1, A1, A0, 2, false)
    WINO_GROUP_H1(2, A0, A1, 3, false)
    WINO_GROUP_H1(3, A1, A0, 0, true)
    __syncthreads();
  }
#undef WINO_GROUP_H0
#undef WINO_GROUP_H1
#undef WINO_PLOAD
#undef WINO_ULOAD
#undef WINO_SLICE
#undef WINO_READ_FRAGS
#undef WINO_SB
#undef WINO_STEP
#undef WINO_LOAD_PATCH
#undef WINO_LOAD_U
#undef WINO_TRANSFORM

  // ---- epilogue -----------------------------------------------------------------------------------------------------------------
  // item = (tile, 4 channels): thread t owns channels (t & 15) * 4 of the tiles (t >> 4) + 16 * it, it = 0..3.  Straight-line code
  // through buffer instructions: a missing tile / row / column gets a byte offset beyond num_records (load 0, store dropped).
  // (0) the residual values (16 x 16 bytes per thread) are requested BEFORE the exchange, so their HBM latency hides behind it
  const int n = cb * WCOUT + (tid & 15) * 4;
  constexpr unsigned WOOB = 0xFFFFFFF0u;
  unsigned voff[4][4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int tl = it * 16 + (tid >> 4);
    const int off = tile_tab[2 * tl], bits = tile_tab[2 * tl + 1];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const bool ok = off >= 0 && (!ii || (bits & 1)) && (!jj || (bits & 2));
        voff[it][ii * 2 + jj] = ok ? (unsigned)(off + (ii * p.Wop + jj) * p.Cout + n) * 4u : WOOB;
      }
  }
  const int out_bytes = p.out_bytes;
  u32x4 res[4][4];
  if (p.residual) {
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, out_bytes, 0x00020000);
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int k = 0; k < 4; ++k) res[it][k] = __builtin_amdgcn_raw_buffer_load_b128(r_res, voff[it][k], 0, 0);
  }
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bias = *reinterpret_cast<const float4*>(p.bias + n);
  if (p.y_act) {
    sc = *reinterpret_cast<const float4*>(p.act_scale + n);
    sh = *reinterpret_cast<const float4*>(p.act_shift + n);
  }
  // (1) row half of the output transform on the wave's own accumulators: with m[w][c] = acc[c] (c = fi),
  //     s[w][0] = m0 + m1 + m2,  s[w][1] = m1 - m2 - m3   -> S[w][jj][tile][cout] in LDS (the K loop ended with a barrier)
  float* S = smem;
  {
    float* sw = S + (size_t)wave * (2 * WT * WCOUT) + ((lane >> 5) * 4) * WCOUT + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m0 = acc[0][i][j][r], m1 = acc[1][i][j][r], m2 = acc[2][i][j][r], m3 = acc[3][i][j][r];
          const int row = i * 32 + (r & 3) + 8 * (r >> 2);
          sw[row * WCOUT + j * 32] = (m0 + m1) + m2;
          sw[WT * WCOUT + row * WCOUT + j * 32] = (m1 - m2) - m3;
        }
  }
  __syncthreads();
  // (2) column half + fused epilogue: Y[0][jj] = s[0][jj] + s[1][jj] + s[2][jj], Y[1][jj] = s[1][jj] - s[2][jj] - s[3][jj]
  const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.y ? out_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc((void*)p.y_act, 0, p.y_act ? out_bytes : 0, 0x00020000);
  const bool has_res = p.residual != nullptr, relu = p.relu != 0, has_act = p.y_act != nullptr;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int tl = it * 16 + (tid >> 4);
    float4 s4[4][2];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) s4[w][jj] = *reinterpret_cast<const float4*>(S + ((size_t)(w * 2 + jj) * WT + tl) * WCOUT + (tid & 15) * 4);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        float4 v;
        if (ii == 0) {
          v.x = (s4[0][jj].x + s4[1][jj].x) + s4[2][jj].x; v.y = (s4[0][jj].y + s4[1][jj].y) + s4[2][jj].y;
          v.z = (s4[0][jj].z + s4[1][jj].z) + s4[2][jj].z; v.w = (s4[0][jj].w + s4[1][jj].w) + s4[2][jj].w;
        } else {
          v.x = (s4[1][jj].x - s4[2][jj].x) - s4[3][jj].x; v.y = (s4[1][jj].y - s4[2][jj].y) - s4[3][jj].y;
          v.z = (s4[1][jj].z - s4[2][jj].z) - s4[3][jj].z; v.w = (s4[1][jj].w - s4[2][jj].w) - s4[3][jj].w;
        }
        v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
        if (has_res) {
          const u32x4 rr = res[it][ii * 2 + jj];
          v.x += __uint_as_float(rr.x); v.y += __uint_as_float(rr.y); v.z += __uint_as_float(rr.z); v.w += __uint_as_float(rr.w);
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        u32x4 o;
        o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
        __builtin_amdgcn_raw_buffer_store_b128(o, r_y, voff[it][ii * 2 + jj], 0, 0);
        if (has_act) {
          u32x4 a;
          a.x = __float_as_uint(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f)); a.y = __float_as_uint(fmaxf(fmaf(v.y, sc.y, sh.y), 0.f));
          a.z = __float_as_uint(fmaxf(fmaf(v.z, sc.z, sh.z), 0.f)); a.w = __float_as_uint(fmaxf(fmaf(v.w, sc.w, sh.w), 0.f));
          __builtin_amdgcn_raw_buffer_store_b128(a, r_act, voff[it][ii * 2 + jj], 0, 0);
        }
      }
  }
}

}  // namespace mp

using namespace mp;

// The 3x3 / stride-1 / pad-1 layers this kernel takes: channel counts that tile (8 input channels per step, 64 output channels per
// workgroup) and enough tiles to give every CU a workgroup (small grids stay on the direct kernel's split-K path).
extern "C" int mp_conv_wino_eligible(const mp_conv_desc* d, int n_cu) {
  if (!d || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->x_f16) return 0;
  if (d->C % WCK != 0 || d->Cout % WCOUT != 0 || d->in_border < 1) return 0;   // 16 input channels per step, 64 output channels per workgroup
  const long tiles = (long)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2);
  const long wgs = ((tiles + WT - 1) / WT) * (d->Cout / WCOUT);
  // the kernels address with 32-bit byte offsets: larger tensors (wide backbones at full batch) stay on the direct kernel
  const long Hp = d->H + 2 * d->in_border, Wp = d->W + 2 * d->in_border;
  const long in_bytes = ((long)d->N * Hp + 2) * Wp * d->C * 4;
  const long out_elems = (long)d->N * (d->H + 2 * d->out_border) * (d->W + 2 * d->out_border) * d->Cout;
  if (tiles >= (1L << 30) || in_bytes >= (1L << 31) || out_elems >= (1L << 29)) return 0;
  // Grid threshold: below it the direct kernel's split-K path takes the layer.  MP_WINO_MIN_WGS overrides it (A/B runs; see DESIGN.md 5:
  // the per-rank refiner batch of an 8-GPU run at the released K = 5 is 40 rows = 200 / 104 workgroups for the 256- / 512-channel layers)
  static const long min_wgs_env = getenv("MP_WINO_MIN_WGS") ? atol(getenv("MP_WINO_MIN_WGS")) : -1;
  // default: a quarter of the CUs.  Measured (profiles/r05_emulated_rank_of_8.txt): the 40-row refiner share of an 8-GPU run at K = 5 has 200 / 104
  // workgroups on its 256- / 512-channel layers; with the round-4 threshold (one workgroup per CU) they fell to split-K: 27.9 ms per share,
  // 21.1 ms at n_cu / 2, 18.5 ms at n_cu / 4 -- and the 5-row K = 5 call on one GPU gets SLOWER below that (53.5 ms at 16 vs 50.6 at 64)
  const long min_wgs = min_wgs_env >= 0 ? min_wgs_env : (long)(n_cu / 4);
  return wgs >= min_wgs ? 1 : 0;
}

static double g_wino_direct = 0.0, g_wino_executed = 0.0;   // host-side totals over the launches since the last reset
extern "C" int mp_conv_wino_stats(double* direct_flops, double* executed_flops, int reset) {
  if (direct_flops) *direct_flops = g_wino_direct;
  if (executed_flops) *executed_flops = g_wino_executed;
  if (reset) g_wino_direct = g_wino_executed = 0.0;
  return MP_OK;
}

extern "C" size_t mp_conv_wino_packed_floats(int Cin_p, int Cout) { return (size_t)16 * Cin_p * Cout; }

// U = G g G^T per (cout, cin), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], computed in double, stored in MFMA fragment order:
// packed[cb][chunk][f][j][lane][q] = U_f[cin = chunk*8 + (lane >> 5)*4 + q][cout = cb*64 + j*32 + (lane & 31)]
extern "C" int mp_conv_wino_pack_weights(const float* w, int Cout, int Cin, int Cin_p, const float* scale, float* packed) {
  MP_REQUIRE(w && packed && Cin_p >= Cin && Cin_p % WCK == 0 && Cout % WCOUT == 0, "mp_conv_wino_pack_weights: bad arguments (Cin_p %% 16, Cout %% 64)");
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int n_chunks = Cin_p / 8, n_cb = Cout / WCOUT;
  memset(packed, 0, mp_conv_wino_packed_floats(Cin_p, Cout) * sizeof(float));
  for (int n = 0; n < Cout; ++n) {
    const double s = scale ? (double)scale[n] : 1.0;
    const int cb = n / WCOUT, j = (n % WCOUT) / 32, nl = n % 32;
    for (int c = 0; c < Cin; ++c) {
      const float* g = w + ((size_t)n * Cin + c) * 9;
      double t[4][3], U[4][4];
      for (int a = 0; a < 4; ++a)
        for (int k = 0; k < 3; ++k) t[a][k] = G[a][0] * g[0 * 3 + k] * s + G[a][1] * g[1 * 3 + k] * s + G[a][2] * g[2 * 3 + k] * s;
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) U[a][b] = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
      const int ch = c / 8, kq = (c % 8) / 4, q = c % 4;
      const int lane = kq * 32 + nl;
      for (int f = 0; f < 16; ++f)
        packed[(((((size_t)cb * n_chunks + ch) * 16 + f) * 2 + j) * 64 + lane) * 4 + q] = (float)U[f / 4][f % 4];
    }
  }
  (void)n_cb;
  return MP_OK;
}

extern "C" int mp_conv3x3_wino_nhwc(const mp_conv_desc* d, const float* d_u, mp_stream stream) {
  MP_REQUIRE(d && d->d_x && d_u && (d->d_y || d->d_y_act), "mp_conv3x3_wino_nhwc: null pointer");
  MP_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1, "mp_conv3x3_wino_nhwc: 3x3 / stride 1 / pad 1 only");
  MP_REQUIRE(d->C % WCK == 0 && d->Cout % WCOUT == 0 && d->in_border >= 1, "mp_conv3x3_wino_nhwc: C %% 16, Cout %% 64, in_border >= 1");
  MP_REQUIRE(!d->d_y_act || (d->d_act_scale && d->d_act_shift), "mp_conv3x3_wino_nhwc: y_act needs scale/shift");
  WinoParams p;
  p.x = d->d_x; p.u = d_u; p.bias = d->d_bias; p.residual = d->d_residual; p.act_scale = d->d_act_scale; p.act_shift = d->d_act_shift;
  p.y = d->d_y; p.y_act = d->d_y_act;
  p.N = d->N; p.Ho = d->H; p.Wo = d->W;
  p.Hp = d->H + 2 * d->in_border; p.Wp = d->W + 2 * d->in_border; p.C = d->C;
  p.in_off = d->in_border - 1;
  p.Cout = d->Cout;
  p.Hop = d->H + 2 * d->out_border; p.Wop = d->W + 2 * d->out_border; p.out_border = d->out_border;
  p.tiles_x = (d->W + 1) / 2; p.tiles_y = (d->H + 1) / 2;
  const long n_tiles = (long)d->N * p.tiles_x * p.tiles_y;
  const long in_bytes = ((long)d->N * p.Hp + 2) * p.Wp * p.C * 4, out_elems = (long)d->N * p.Hop * p.Wop * d->Cout;
  MP_REQUIRE(n_tiles < (1L << 30) && in_bytes < (1L << 31) && out_elems < (1L << 29), "mp_conv3x3_wino_nhwc: tensor too large for 32-bit offsets");
  p.out_bytes = (int)(out_elems * 4);
  p.n_tiles = (int)n_tiles;
  p.n_chunks = d->C / 8;
  p.n_steps = d->C / WCK;
  p.relu = d->relu;
  p.telemetry = 0;
  p.n_cblocks = d->Cout / WCOUT;
  int dev = 0;
  MP_CHECK_HIP(hipGetDevice(&dev));
  static int attr_dev = -1;
  if (attr_dev != dev) {   // (per device: the attribute does not travel with the process)
    MP_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_wino_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO_LDS_BYTES));
    attr_dev = dev;
  }
  const long n_wg = ((n_tiles + WT - 1) / WT) * p.n_cblocks;
  hipStream_t s = (hipStream_t)stream;
  // profiler row: the ALGORITHMIC work of the layer as SURVEY.md 8d defines it (2 * MACs of the direct convolution over the real
  // channels; bytes = input + weights + output once).  What the kernel EXECUTES is 16 multiplications per 2x2 tile and (cin, cout)
  // pair -- 16/36 of that for even sizes; both totals are accumulated for mp_conv_wino_stats (bench.py reports the executed rate as
  // the kernel's MFMA utilisation, which is <= 1 by construction, next to the algorithmic rate, which may exceed the matrix peak).
  const double c_real = d->c_real > 0 ? d->c_real : d->C;
  const double direct = 2.0 * 9.0 * (double)d->N * d->H * d->W * c_real * d->Cout, executed = 2.0 * 16.0 * (double)n_tiles * c_real * d->Cout;
  g_wino_direct += direct;
  g_wino_executed += executed;
  ProfScope prof("conv3x3_wino_f32<64t,64c>", direct, 4.0 * ((double)d->N * d->H * d->W * (d->C + d->Cout) + 16.0 * d->C * d->Cout), s, executed, 157.3);
  hipLaunchKernelGGL(conv3x3_wino_f32, dim3((unsigned)n_wg), dim3(256), WINO_LDS_BYTES, s, p);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}
