// conv.hip -- fp32 implicit-GEMM convolution on MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces the cuDNN convolutions behind `self.backbone(x)`
// (reference: src/megapose/models/pose_rigid.py:323; layers in
//  src/megapose/models/torchvision_resnet.py:74-120,181-316 and src/megapose/models/wide_resnet.py:29-111).
//
// Formulation ("row-run implicit GEMM"): activations are padded NHWC, so for a fixed kernel row kh the
// (kw, c) taps of one output pixel are ONE contiguous run of KW*C floats in memory, and the zero border
// makes bounds checks unnecessary.  GEMM view:  M = N*Ho*Wo output pixels, N = Cout, K = KH * (KW*C).
// The K loop walks (kh, 32-float chunk of the run); A tiles are gathered row-by-row (128-B contiguous
// pieces), B tiles come from a pre-packed weight blob laid out in exactly the loop order.  Both are
// staged through LDS (double-buffered, register prefetch), read back as 16-B fragments, and fed to the
// 32x32x2 fp32 MFMA, whose k-order within a 8-float group is permuted identically for A and B (free).
// Epilogue fuses bias (folded eval-BN), residual add, ReLU and an optional second "pre-activated"
// output relu(y*s+t) for pre-activation (WideResNet) blocks.
//
// Roofline: MFMA-bound (fp32 matrix peak 157.3 TFLOP/s, MI355X_MICROARCH.md).
#include <cstdlib>

#include <algorithm>
#include <string>
#include <vector>

#include "common.h"

namespace mp {

constexpr int BK = 32;        // floats of K per chunk
constexpr int LDS_LD = BK + 4;  // padded LDS row (36 floats): conflict-free 16-B fragment reads

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
  const float* __restrict__ x;
  const float* __restrict__ w;
  const float* __restrict__ bias;
  const float* __restrict__ residual;
  const float* __restrict__ act_scale;
  const float* __restrict__ act_shift;
  float* __restrict__ y;
  float* __restrict__ y_act;
  int M;               // N*Ho*Wo
  int Ho, Wo;
  int Hp, Wp, C;       // padded input geometry
  int in_off;          // in_border - pad
  int stride;
  int Cout;
  int Hop, Wop, out_border;
  int KH;
  int run;             // KW*C floats: contiguous (kw, c) taps of one kernel row
  int n_chunks;        // ceil(KH*run / BK): the K loop walks the concatenated row runs
  int relu;
  int n_mblocks, n_nblocks;
  // split-K (small M: too few tiles to fill 256 CUs): blockIdx.y owns chunks [y*chunks_per_split, ...) and writes raw partial
  // sums to partial[y][M][Cout]; conv_splitk_reduce adds them in a fixed order and applies the epilogue (deterministic)
  float* partial;
  int k_split, chunks_per_split;
  int tile_begin;    // first linear tile id of this launch (the tail launch of a "full rounds + split-K tail" pair starts later)
  int m_part_begin;  // first output row held by `partial` (rows before it belong to the single-pass launch)
};

// Clock telemetry (always on, ~free): every 64th workgroup adds the shader cycles (s_memtime) and the 100 MHz real-time ticks
// (s_memrealtime) it spent in its K loop; mp_conv_clock_read turns the sums into the effective shader clock the convolutions ran at.
// The fp32-MFMA peak scales with that clock, and under real (bit-toggling) operands this part sustains ~2.15 GHz, not 2.4.
__device__ unsigned long long g_conv_clk[2];
#ifdef MP_RASTER_PROF   // scripts/microbench build only: where a wave's cycles go inside one K-loop iteration: 0 load issue,
// 1 MFMA groups 0-1, 2 wait for the global loads + LDS writes, 3 MFMA groups 2-3, 4 barrier
__device__ unsigned long long g_conv_seg[8];
#define CPROF(slot)                                                    \
  {                                                                    \
    const unsigned long long cprof_n = __builtin_readcyclecounter();   \
    cprof_acc[slot] += cprof_n - cprof_t;                              \
    cprof_t = cprof_n;                                                 \
  }
#else
#define CPROF(slot)
#endif

// Fused epilogue through buffer instructions: one scalar resource per tensor based at the tile's first output row, one 32-bit
// byte offset per (lane, tile row); rows past M and channels past Cout get an offset beyond num_records, which the hardware's range
// check turns into "load 0 / drop the store" -- no per-element branches, no 64-bit lane addresses.
// (Requesting the residual tile before the K loop, to hide its latency, was measured: 2 % slower -- the loads compete with the first
// chunks and 32-64 registers stay live across the loop.)
constexpr unsigned EPI_WINDOW = 0x40000000u;   // 1 GB window from the tile's first row: a tile spans a few image rows

template <int TM>
__device__ __forceinline__ void conv_row_offsets(const int* row_off, int row0, int n_first, int base, unsigned (&voff)[TM][16]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = row_off[row0 + i * 32 + (r & 3) + 8 * (r >> 2)];
      voff[i][r] = o >= 0 ? (unsigned)(o - base + n_first) * 4u : EPI_WINDOW;
    }
}

template <int TM, int TN, bool RES, bool RELU, bool ACT>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const f32x16 (&acc)[TM][TN], const int* row_off, int row0, int n_first) {
  const int base = __builtin_amdgcn_readfirstlane(row_off[0]);   // row 0 of a launched tile always exists
  unsigned voff[TM][16];
  conv_row_offsets<TM>(row_off, row0, n_first, base, voff);
  float res[TM][TN][16];
  if (RES) {   // every residual value of the wave's tile is requested before the first one is used: one exposed latency per tile
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual + base), 0, (int)EPI_WINDOW, 0x00020000);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bool n_ok = n_first + j * 32 < p.Cout;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          res[i][j][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_res, n_ok ? voff[i][r] + j * 128 : EPI_WINDOW, 0, 0));
    }
  }
  const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y ? p.y + base : nullptr), 0, p.y ? (int)EPI_WINDOW : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_act =
      __builtin_amdgcn_make_buffer_rsrc((void*)(ACT ? p.y_act + base : nullptr), 0, ACT ? (int)EPI_WINDOW : 0, 0x00020000);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n_first + j * 32;
    const bool n_ok = n < p.Cout;
    const float bias = (p.bias && n_ok) ? p.bias[n] : 0.f;
    float sc = 1.f, sh = 0.f;
    if (ACT && n_ok) {
      sc = p.act_scale[n];
      sh = p.act_shift[n];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned vo = n_ok ? voff[i][r] + j * 128 : EPI_WINDOW;
        float v = acc[i][j][r] + bias;
        if (RES) v += res[i][j][r];
        if (RELU) v = fmaxf(v, 0.f);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r_y, vo, 0, 0);   // (a null y has num_records = 0: dropped)
        if (ACT) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaxf(fmaf(v, sc, sh), 0.f)), r_act, vo, 0, 0);
      }
    }
  }
}

// LDS-transposed epilogue with 16-byte stores (Cout % 4 == 0).  The MFMA C layout gives a lane ONE column of 16 scattered rows, so the
// direct epilogue above needs TM*TN*16 dword stores (and as many dword residual loads) per wave -- store-ISSUE-bound: 9 % of an
// 18-chunk layer-1 tile.  Here each wave parks its WM x WN accumulator tile in its own slice of the (now idle) A/B staging LDS
// (row stride = WN floats: conflict-free for both the ds_write_b32 column writes and the ds_read_b128 row reads) and walks it back in
// rows: a lane owns 4 consecutive channels of one output pixel -> one buffer_load_b128 of the residual, one buffer_store_b128 of the
// result (4x fewer VMEM instructions, whole 128/256-byte row segments per 8/16 lanes).  Same arithmetic per element, same masking
// through the buffer range check.  The wave only touches its own slice: no workgroup barrier.
template <int TM, int TN, bool RES, bool RELU, bool ACT>
__device__ __forceinline__ void conv_epilogue_vec(const ConvParams& p, const f32x16 (&acc)[TM][TN], const int* row_off, float* stage,
                                                  int wave_row0, int n_wave0) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int LDW = TN * 32;              // floats per staged row
  constexpr int LPR = TN * 8;               // lanes per row (one float4 each)
  constexpr int RPI = 64 / LPR;             // rows per iteration
  constexpr int NIT = TM * 32 / RPI;        // iterations
  const int lane = threadIdx.x & 63;
  {
    float* st = stage + ((lane >> 5) * 4) * LDW + (lane & 31);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[(i * 32 + (r & 3) + 8 * (r >> 2)) * LDW + j * 32] = acc[i][j][r];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int base = __builtin_amdgcn_readfirstlane(row_off[0]);
  const int lr = lane / LPR, lc = (lane % LPR) * 4;
  const int n = n_wave0 + lc;
  const bool n_ok = n < p.Cout;
  unsigned voff[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int o = row_off[wave_row0 + it * RPI + lr];
    voff[it] = (o >= 0 && n_ok) ? (unsigned)(o - base + n) * 4u : EPI_WINDOW;
  }
  u32x4 res[NIT];
  if (RES) {
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual + base), 0, (int)EPI_WINDOW, 0x00020000);
#pragma unroll
    for (int it = 0; it < NIT; ++it) res[it] = __builtin_amdgcn_raw_buffer_load_b128(r_res, voff[it], 0, 0);
  }
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias && n_ok) bias = *reinterpret_cast<const float4*>(p.bias + n);
  if (ACT && n_ok) {
    sc = *reinterpret_cast<const float4*>(p.act_scale + n);
    sh = *reinterpret_cast<const float4*>(p.act_shift + n);
  }
  const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y ? p.y + base : nullptr), 0, p.y ? (int)EPI_WINDOW : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_act =
      __builtin_amdgcn_make_buffer_rsrc((void*)(ACT ? p.y_act + base : nullptr), 0, ACT ? (int)EPI_WINDOW : 0, 0x00020000);
  const float* rd = stage + lr * LDW + lc;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    float4 v = *reinterpret_cast<const float4*>(rd + it * RPI * LDW);
    v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
    if (RES) {
      v.x += __uint_as_float(res[it].x); v.y += __uint_as_float(res[it].y);
      v.z += __uint_as_float(res[it].z); v.w += __uint_as_float(res[it].w);
    }
    if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    u32x4 o;
    o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(o, r_y, voff[it], 0, 0);
    if (ACT) {
      u32x4 a;
      a.x = __float_as_uint(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f)); a.y = __float_as_uint(fmaxf(fmaf(v.y, sc.y, sh.y), 0.f));
      a.z = __float_as_uint(fmaxf(fmaf(v.z, sc.z, sh.z), 0.f)); a.w = __float_as_uint(fmaxf(fmaf(v.w, sc.w, sh.w), 0.f));
      __builtin_amdgcn_raw_buffer_store_b128(a, r_act, voff[it], 0, 0);
    }
  }
}

// waves_per_eu(2,2): LDS already limits residency to 2 workgroups per CU (= 2 waves per SIMD); telling the compiler so lets it
// keep the prefetch registers live across the MFMA block instead of spilling them to scratch to chase a higher occupancy.
// VARIANT: 1 = the LDS store of the next chunk sits under the LAST MFMA group, | 256 = under the 3rd of 4; | 4096 = the two-chunks-ahead
// pipeline (barrier before the last group); | 8192 = buffer loads (scalar resource + 32-bit offsets) instead of global loads;
// | 16384 = persistent workgroups: the launch has one workgroup per resident slot (2 per CU) and each walks the tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... (p.k_split carries the tile count of the launch) -- no workgroup dispatch between tiles.
// AHALF: the INPUT tensor holds IEEE binary16 values (the "fp16 renders" CNN input the rasteriser writes with MP_RASTER_F16): the A
// tile is fetched as 8-byte pieces (4 halves), stays packed in registers across the MFMA block and is widened to fp32 on its way
// into LDS -- everything after the LDS store (fragments, MFMA, epilogue) is the fp32 path unchanged.
template <int BM, int BN, int WM, int WN, int VARIANT = 0, bool RAGGED = false, bool SPLITK = false, bool AHALF = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_nhwc_f32_mfma(ConvParams p) {
  constexpr int NBUF = 2;
  constexpr int LDT = LDS_LD;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_LD4 = BM / 32;  // float4 loads per thread for the A tile
  constexpr int B_LD4 = BN / 32;

  const bool clk_sample = threadIdx.x == 0 && (blockIdx.x & 63) == 0;
  unsigned long long clk_c0 = 0, clk_r0 = 0;
  if (clk_sample) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                          // [NBUF][BM][LDT]
  float* Bs = smem + NBUF * BM * LDT;        // [NBUF][BN][LDT]
  int* row_off = (int*)(Bs + NBUF * BN * LDT);  // [BM] output element offset of each tile row (-1: none)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / (BN / WN);
  const int wn = wave % (BN / WN);

  constexpr bool PERSIST = (VARIANT & 16384) != 0;
  constexpr bool VEC_EPI = (VARIANT & 65536) != 0;   // LDS-transposed epilogue with 16-byte stores (conv_epilogue_vec)
  static_assert(4 * WM * WN <= NBUF * (BM + BN) * LDT, "the epilogue stages the accumulators in the A/B buffers");
  static_assert(!PERSIST || !SPLITK, "persistent workgroups are a single-pass launch mode");
  for (int tile = blockIdx.x; PERSIST ? tile < p.k_split : true; tile += gridDim.x) {   // (not PERSIST: exactly one trip, the loop folds away)
  if constexpr (PERSIST) {
    if (clk_sample) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
  }
  const int lb = (PERSIST ? xcd_remap(tile, p.k_split) : xcd_remap(blockIdx.x, gridDim.x)) + p.tile_begin;
  const int nblk = lb % p.n_nblocks;
  const int mblk = lb / p.n_nblocks;
  const int m0 = mblk * BM;
  const int n0 = nblk * BN;

  // ---- per-thread A row pointers (one output pixel per row) -------------------------------
  const int a_c4 = tid & 7;   // which float4 of the 32-float chunk
  const int a_r0 = tid >> 3;  // 0..31
  const float* a_ptr[A_LD4];
#pragma unroll
  for (int i = 0; i < A_LD4; ++i) {
    int m = m0 + a_r0 + 32 * i;
    m = m < p.M ? m : p.M - 1;
    const int wo = m % p.Wo;
    const int t = m / p.Wo;
    const int ho = t % p.Ho;
    const int n = t / p.Ho;
    const size_t pix = ((size_t)n * p.Hp + (size_t)(ho * p.stride + p.in_off)) * p.Wp + (size_t)(wo * p.stride + p.in_off);
    a_ptr[i] = p.x + pix * p.C;
  }
  // output offsets of this tile's rows
  for (int r = tid; r < BM; r += 256) {
    const int m = m0 + r;
    int off = -1;
    if (m < p.M) {
      const int wo = m % p.Wo;
      const int t = m / p.Wo;
      const int ho = t % p.Ho;
      const int n = t / p.Ho;
      off = (((n * p.Hop) + ho + p.out_border) * p.Wop + wo + p.out_border) * p.Cout;
    }
    row_off[r] = off;
  }

  const int a_col = a_c4 * 4;  // this thread's float offset in a chunk row
  const float* b_ptr = p.w + (size_t)nblk * p.n_chunks * (BN * BK) + (tid >> 3) * BK + a_col;
  // VARIANT bit 8192: buffer_load_dwordx4 (scalar resource + 32-bit lane offset + scalar chunk offset) instead of global_load_dwordx4
  // with 64-bit lane addresses.  The A resource starts at the tile's first pixel (addresses ascend with the row index, the tile spans
  // a few image rows), the B resource at this n-block's packed weights; neither range check can trigger (num_records = 2^32 - 1).
  constexpr bool BUFLD = (VARIANT & 8192) != 0;
  static_assert(!AHALF || BUFLD, "the half-precision input path is written for the buffer-load variants");
  constexpr int A_ES = AHALF ? 2 : 4;   // bytes per input element
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  int a_voff[A_LD4] = {0, 0, 0, 0};
  int b_voff = 0;
  __amdgpu_buffer_rsrc_t a_rsrc, b_rsrc;
  if constexpr (BUFLD) {
    const int mb = m0 < p.M ? m0 : p.M - 1;
    const int wo = mb % p.Wo;
    const int t = mb / p.Wo;
    const int ho = t % p.Ho;
    const int n = t / p.Ho;
    const size_t pix = ((size_t)n * p.Hp + (size_t)(ho * p.stride + p.in_off)) * p.Wp + (size_t)(wo * p.stride + p.in_off);
    const float* a_base = p.x + pix * p.C;   // (AHALF: p.x really points at halves; a_ptr / a_base only serve as element counters)
    if constexpr (AHALF)
      a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const _Float16*>(p.x) + pix * p.C), 0, -1, 0x00020000);
    else
      a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, -1, 0x00020000);
    b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (size_t)nblk * p.n_chunks * (BN * BK)), 0, -1, 0x00020000);
#pragma unroll
    for (int i = 0; i < A_LD4; ++i) a_voff[i] = (int)((a_ptr[i] - a_base) * A_ES);
    b_voff = ((tid >> 3) * BK + a_col) * 4;
  }
  const int row_stride = p.Wp * p.C;  // floats between successive kh rows

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Straight-line software pipeline (no lambdas / no conditional loads: hipcc otherwise parks the prefetch registers in
  // scratch and waits for the loads right away).  The loads of chunk c+1 are issued before the MFMAs of chunk c and written
  // to the other LDS buffer after them; the last iteration harmlessly re-loads the last chunk.
// (explicit scalars, not arrays: the arrays only become registers after loop unrolling, and the sched_barrier intrinsic
//  that pins the prefetch is a memory barrier to the optimiser, which would leave them in scratch)
#define MP_LD4(P) (*reinterpret_cast<const float4*>(P))
#define MP_BUF4(R, V, S) ([&] { const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(R, V, S, 0);              \
    return make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); }())
#define MP_CONV_LOAD(AOFF, BP)                                                                                     \
  if constexpr (BUFLD) {                                                                                           \
    /* RAGGED: the run position is per lane (vector offset); otherwise only a_col is, the rest rides in the scalar offset */ \
    const int av_ = RAGGED ? (AOFF) * A_ES : a_col * A_ES;                                                         \
    const int as_ = RAGGED ? 0 : a_su * A_ES;   /* a_su / b_su: the wave-uniform parts of AOFF / BP, tracked separately */ \
    const int bs_ = b_su * 4;                                                                                      \
    if constexpr (AHALF) {   /* 4 halves per lane and tile row: widened when they are written to LDS */            \
      ah0 = __builtin_amdgcn_raw_buffer_load_b64(a_rsrc, a_voff[0] + av_, as_, 0);                                 \
      ah1 = __builtin_amdgcn_raw_buffer_load_b64(a_rsrc, a_voff[1] + av_, as_, 0);                                 \
      ah2 = __builtin_amdgcn_raw_buffer_load_b64(a_rsrc, a_voff[2] + av_, as_, 0);                                 \
      ah3 = __builtin_amdgcn_raw_buffer_load_b64(a_rsrc, a_voff[3] + av_, as_, 0);                                 \
    } else {                                                                                                       \
      a0 = MP_BUF4(a_rsrc, a_voff[0] + av_, as_);                                                                  \
      a1 = MP_BUF4(a_rsrc, a_voff[1] + av_, as_);                                                                  \
      a2 = MP_BUF4(a_rsrc, a_voff[2] + av_, as_);                                                                  \
      a3 = MP_BUF4(a_rsrc, a_voff[3] + av_, as_);                                                                  \
    }                                                                                                              \
    b0 = MP_BUF4(b_rsrc, b_voff, bs_);                                                                             \
    b1 = MP_BUF4(b_rsrc, b_voff, bs_ + 4096);                                                                      \
    if constexpr (B_LD4 > 2) {                                                                                     \
      b2 = MP_BUF4(b_rsrc, b_voff, bs_ + 8192);                                                                    \
      b3 = MP_BUF4(b_rsrc, b_voff, bs_ + 12288);                                                                   \
    }                                                                                                              \
  } else {                                                                                                         \
    a0 = MP_LD4(a_ptr0 + (AOFF));                                                                                  \
    a1 = MP_LD4(a_ptr1 + (AOFF));                                                                                  \
    a2 = MP_LD4(a_ptr2 + (AOFF));                                                                                  \
    a3 = MP_LD4(a_ptr3 + (AOFF));                                                                                  \
    b0 = MP_LD4((BP));                                                                                             \
    b1 = MP_LD4((BP) + 1024);                                                                                      \
    if constexpr (B_LD4 > 2) {                                                                                     \
      b2 = MP_LD4((BP) + 2048);                                                                                    \
      b3 = MP_LD4((BP) + 3072);                                                                                    \
    }                                                                                                              \
  }
#define MP_ST4(P, V) (*reinterpret_cast<float4*>(P) = (V))
#define MP_H4(V) ([&] { const f16x4 h_ = __builtin_bit_cast(f16x4, V);                                             \
    return make_float4((float)h_.x, (float)h_.y, (float)h_.z, (float)h_.w); }())
#define MP_CONV_STORE(BUF)                                                        \
  {                                                                               \
    float* as_w = As + (BUF) * BM * LDT + a_r0 * LDT + a_c4 * 4;            \
    float* bs_w = Bs + (BUF) * BN * LDT + (tid >> 3) * LDT + (tid & 7) * 4; \
    if constexpr (AHALF) {                                                        \
      a0 = MP_H4(ah0);                                                            \
      a1 = MP_H4(ah1);                                                            \
      a2 = MP_H4(ah2);                                                            \
      a3 = MP_H4(ah3);                                                            \
    }                                                                             \
    MP_ST4(as_w, a0);                                                             \
    MP_ST4(as_w + 32 * LDT, a1);                                               \
    MP_ST4(as_w + 64 * LDT, a2);                                               \
    MP_ST4(as_w + 96 * LDT, a3);                                               \
    MP_ST4(bs_w, b0);                                                             \
    MP_ST4(bs_w + 32 * LDT, b1);                                               \
    if constexpr (B_LD4 > 2) {                                                    \
      MP_ST4(bs_w + 64 * LDT, b2);                                             \
      MP_ST4(bs_w + 96 * LDT, b3);                                             \
    }                                                                             \
  }
  static_assert(A_LD4 == 4 && (B_LD4 == 2 || B_LD4 == 4), "staging code is written for BM = 128, BN in {64, 128}");
  const float* a_ptr0 = a_ptr[0];
  const float* a_ptr1 = a_ptr[1];
  const float* a_ptr2 = a_ptr[2];
  const float* a_ptr3 = a_ptr[3];
  float4 a0, a1, a2, a3, b0, b1, b2, b3;
  u32x2 ah0, ah1, ah2, ah3;   // AHALF: the packed halves of the prefetched A pieces

  // per-thread position of its float4 inside the concatenated K axis: j = offset in the current kernel row's run,
  // aoff = element offset from the pixel's first tap (run % 4 == 0, so a float4 never straddles two rows)
  // RAGGED (run % 32 != 0, the 7x7 / 5x5 stems): per-lane bookkeeping.  Otherwise chunks never straddle kernel rows and
  // the bump is wave-uniform (scalar registers), which is what the 3x3 / 1x1 layers use.
  int j = a_col, aoff = a_col;
  int ju = 0;  // uniform run position (non-RAGGED)
  int a_su = 0, b_su = 0;  // buffer-load variant: uniform float offsets of the current chunk (A: non-RAGGED only)
  const float* bp = b_ptr;
  int c_begin = 0, c_end = p.n_chunks;
  if constexpr (SPLITK) {
    c_begin = blockIdx.y * p.chunks_per_split;
    c_end = min(p.n_chunks, c_begin + p.chunks_per_split);
    bp += (size_t)c_begin * (BN * BK);
    b_su = c_begin * (BN * BK);
    if constexpr (RAGGED) {
      const int jj = a_col + c_begin * BK;
      const int kh = jj / p.run;
      j = jj - kh * p.run;
      aoff = kh * row_stride + j;
    } else {
      const int kh = (c_begin * BK) / p.run;
      ju = c_begin * BK - kh * p.run;
      aoff = a_col + kh * row_stride + ju;
      a_su = kh * row_stride + ju;
    }
  } else if constexpr (RAGGED) {
    while (j >= p.run) { j -= p.run; aoff += row_stride - p.run; }
  }
  MP_CONV_LOAD(aoff, bp)
  MP_CONV_STORE(0)
  __syncthreads();

  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 4;
  const int row_wrap = row_stride - p.run;
#ifdef MP_RASTER_PROF
  unsigned long long cprof_acc[5] = {0, 0, 0, 0, 0};
  unsigned long long cprof_t = __builtin_readcyclecounter();
#endif
// advance the load position to the next chunk (guarded by COND; the loads themselves are unconditional, the last one is repeated)
#define MP_CONV_ADVANCE(COND)                                                                                      \
  if (COND) {                                                                                                      \
    bp += BN * BK;                                                                                                 \
    b_su += BN * BK;                                                                                               \
    aoff += BK;                                                                                                    \
    a_su += BK;                                                                                                    \
    if constexpr (RAGGED) {                                                                                        \
      j += BK;                                                                                                     \
      while (j >= p.run) { /* crossed into the next kernel row(s) (runs shorter than BK wrap more than once) */    \
        j -= p.run;                                                                                                \
        aoff += row_wrap;                                                                                          \
      }                                                                                                            \
    } else {                                                                                                       \
      ju += BK;                                                                                                    \
      if (ju == p.run) {                                                                                           \
        ju = 0;                                                                                                    \
        aoff += row_wrap;                                                                                          \
        a_su += row_wrap;                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
  }
  if constexpr ((VARIANT & 4096) != 0) {
    // Two-chunks-ahead pipeline, ONE barrier per chunk placed before the LAST MFMA group:
    //   registers G hold chunk c+1 (requested a whole chunk ago), fragments of k-group g+1 are read from LDS under the 16 MFMAs of
    //   group g -- across the chunk boundary too: the barrier sits where (i) every wave has written chunk c+1 to the other buffer
    //   (under group 0) and (ii) every wave has READ its last fragments of chunk c (requested under group 2), so after it the other
    //   buffer may be read (group 0 of chunk c+1, under group 3 of chunk c) and this buffer may be overwritten (top of chunk c+1).
    static_assert(BK == 32, "four k-groups per chunk");
    MP_CONV_ADVANCE(c_begin + 1 < c_end)
    MP_CONV_LOAD(aoff, bp)
    float4 afr[2][TM], bfr[2][TN];
    {
      const float* as0 = As + (wm * WM + frag_row) * LDT + frag_k;
      const float* bs0 = Bs + (wn * WN + frag_row) * LDT + frag_k;
#pragma unroll
      for (int i = 0; i < TM; ++i) afr[0][i] = *reinterpret_cast<const float4*>(as0 + i * 32 * LDT);
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) bfr[0][jn] = *reinterpret_cast<const float4*>(bs0 + jn * 32 * LDT);
    }
#define MP_FRAG_READ(SLOT, AS, BS, KOFF)                                                                          \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) afr[SLOT][i] = *reinterpret_cast<const float4*>((AS) + i * 32 * LDT + (KOFF));  \
  _Pragma("unroll") for (int jn = 0; jn < TN; ++jn) bfr[SLOT][jn] = *reinterpret_cast<const float4*>((BS) + jn * 32 * LDT + (KOFF));
#define MP_MFMA_ROW(SLOT, I)                                                                                       \
  _Pragma("unroll") for (int jn = 0; jn < TN; ++jn) {                                                              \
    acc[I][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[SLOT][I].x, bfr[SLOT][jn].x, acc[I][jn], 0, 0, 0);       \
    acc[I][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[SLOT][I].y, bfr[SLOT][jn].y, acc[I][jn], 0, 0, 0);       \
    acc[I][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[SLOT][I].z, bfr[SLOT][jn].z, acc[I][jn], 0, 0, 0);       \
    acc[I][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[SLOT][I].w, bfr[SLOT][jn].w, acc[I][jn], 0, 0, 0);       \
  }
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
      const int buf = (chunk - c_begin) & 1;
      const float* as = As + buf * BM * LDT + (wm * WM + frag_row) * LDT + frag_k;
      const float* bs = Bs + buf * BN * LDT + (wn * WN + frag_row) * LDT + frag_k;
      const float* as_n = As + (buf ^ 1) * BM * LDT + (wm * WM + frag_row) * LDT + frag_k;
      const float* bs_n = Bs + (buf ^ 1) * BN * LDT + (wn * WN + frag_row) * LDT + frag_k;
      // group 0 (+ chunk c+1: registers -> other buffer; chunk c+2: global -> registers)
      MP_FRAG_READ(1, as, bs, 8)
      __builtin_amdgcn_sched_barrier(0);
      MP_MFMA_ROW(0, 0)
      __builtin_amdgcn_sched_barrier(0);
      MP_CONV_STORE(buf ^ 1)
      __builtin_amdgcn_sched_barrier(0);
      MP_MFMA_ROW(0, 1)
      __builtin_amdgcn_sched_barrier(0);
      MP_CONV_ADVANCE(chunk + 2 < c_end)
      MP_CONV_LOAD(aoff, bp)
      __builtin_amdgcn_sched_barrier(0);
      // group 1
      MP_FRAG_READ(0, as, bs, 16)
      __builtin_amdgcn_sched_barrier(0);
      MP_MFMA_ROW(1, 0)
      MP_MFMA_ROW(1, 1)
      __builtin_amdgcn_sched_barrier(0);
      // group 2
      MP_FRAG_READ(1, as, bs, 24)
      __builtin_amdgcn_sched_barrier(0);
      MP_MFMA_ROW(0, 0)
      MP_MFMA_ROW(0, 1)
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      // group 3, with the first fragments of the next chunk
      MP_FRAG_READ(0, as_n, bs_n, 0)
      __builtin_amdgcn_sched_barrier(0);
      MP_MFMA_ROW(1, 0)
      MP_MFMA_ROW(1, 1)
      __builtin_amdgcn_sched_barrier(0);
    }
#undef MP_FRAG_READ
#undef MP_MFMA_ROW
  } else
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const int buf = (chunk - c_begin) & 1;
    MP_CONV_ADVANCE(chunk + 1 < c_end)
    MP_CONV_LOAD(aoff, bp)
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMA block (the scheduler otherwise sinks it to the end)
    CPROF(0)
    const float* as = As + buf * BM * LDT + (wm * WM + frag_row) * LDT + frag_k;
    const float* bs = Bs + buf * BN * LDT + (wn * WN + frag_row) * LDT + frag_k;
    constexpr int STORE_KK = (VARIANT & 256) ? 2 : BK / 8 - 1;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDT + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(bs + j * 32 * LDT + kk * 8);
      if (kk == STORE_KK) {  // write the prefetched chunk to the other buffer UNDER an MFMA group
        __builtin_amdgcn_sched_barrier(0);
        CPROF(1)
        MP_CONV_STORE(buf ^ 1)
        __builtin_amdgcn_sched_barrier(0);
        CPROF(2)
      }
      // (experiment, MP_CONV_EXPERIMENTS builds only: | 32768 = raise the wave's issue priority for the duration of an MFMA group, so
      //  that the co-resident workgroup's loads / LDS traffic never delay the next MFMA of the wave that owns the matrix pipe)
      if constexpr ((VARIANT & 32768) != 0) __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
      if constexpr ((VARIANT & 32768) != 0) __builtin_amdgcn_s_setprio(0);
    }
    CPROF(3)
    __syncthreads();
    CPROF(4)
  }
#ifdef MP_RASTER_PROF
  if (lane == 0 && (blockIdx.x & 15) == 0)
    for (int k = 0; k < 5; ++k) atomicAdd(&g_conv_seg[k], cprof_acc[k]);
#endif
#undef MP_CONV_LOAD
#undef MP_BUF4
#undef MP_CONV_STORE
#undef MP_H4
#undef MP_CONV_ADVANCE
#undef MP_LD4
#undef MP_ST4

  if (clk_sample) {   // prologue + K loop of this workgroup
    atomicAdd(&g_conv_clk[0], __builtin_readcyclecounter() - clk_c0);
    atomicAdd(&g_conv_clk[1], __builtin_amdgcn_s_memrealtime() - clk_r0);
  }
  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  // (compile-time flags: one straight-line store loop per fused mode instead of four data-dependent branches per element)
  const int erow0 = wm * WM + (lane >> 5) * 4, en0 = n0 + wn * WN + (lane & 31);
  if constexpr (SPLITK) {
    float* part = p.partial + (size_t)blockIdx.y * (p.M - p.m_part_begin) * p.Cout;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      const int n = en0 + jn * 32;
      if (n >= p.Cout) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + erow0 + i * 32 + (r & 3) + 8 * (r >> 2);
          if (m < p.M) part[(size_t)(m - p.m_part_begin) * p.Cout + n] = acc[i][jn][r];
        }
    }
    return;
  }
  const int emode = (p.residual ? 1 : 0) | (p.relu ? 2 : 0) | (p.y_act ? 4 : 0);
  if (VEC_EPI && (p.Cout & 3) == 0) {   // 16-byte path (every backbone layer); Cout % 4 != 0 (detector predictor heads) takes the dword path
    float* stage = smem + wave * (WM * WN);   // the wave's slice of the idle A/B staging buffers (the K loop ended with a barrier)
    const int vr0 = wm * WM, vn0 = n0 + wn * WN;
    switch (emode) {
      case 0: conv_epilogue_vec<TM, TN, false, false, false>(p, acc, row_off, stage, vr0, vn0); break;
      case 1: conv_epilogue_vec<TM, TN, true, false, false>(p, acc, row_off, stage, vr0, vn0); break;
      case 2: conv_epilogue_vec<TM, TN, false, true, false>(p, acc, row_off, stage, vr0, vn0); break;
      case 3: conv_epilogue_vec<TM, TN, true, true, false>(p, acc, row_off, stage, vr0, vn0); break;
      case 4: conv_epilogue_vec<TM, TN, false, false, true>(p, acc, row_off, stage, vr0, vn0); break;
      case 5: conv_epilogue_vec<TM, TN, true, false, true>(p, acc, row_off, stage, vr0, vn0); break;
      case 6: conv_epilogue_vec<TM, TN, false, true, true>(p, acc, row_off, stage, vr0, vn0); break;
      default: conv_epilogue_vec<TM, TN, true, true, true>(p, acc, row_off, stage, vr0, vn0); break;
    }
  } else
  switch (emode) {
    case 0: conv_epilogue<TM, TN, false, false, false>(p, acc, row_off, erow0, en0); break;
    case 1: conv_epilogue<TM, TN, true, false, false>(p, acc, row_off, erow0, en0); break;
    case 2: conv_epilogue<TM, TN, false, true, false>(p, acc, row_off, erow0, en0); break;
    case 3: conv_epilogue<TM, TN, true, true, false>(p, acc, row_off, erow0, en0); break;
    case 4: conv_epilogue<TM, TN, false, false, true>(p, acc, row_off, erow0, en0); break;
    case 5: conv_epilogue<TM, TN, true, false, true>(p, acc, row_off, erow0, en0); break;
    case 6: conv_epilogue<TM, TN, false, true, true>(p, acc, row_off, erow0, en0); break;
    default: conv_epilogue<TM, TN, true, true, true>(p, acc, row_off, erow0, en0); break;
  }
  if constexpr (!PERSIST) break;
  __syncthreads();   // the next tile rewrites row_off (and the LDS stages) that slower waves may still be reading in their epilogue
  }
}

// sum of the k_split partial tiles in ascending split order + the fused epilogue (bias, residual, ReLU, pre-activation output)
__global__ __launch_bounds__(256) void conv_splitk_reduce(ConvParams p) {
  const int c4 = p.Cout >> 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int m_part = p.M - p.m_part_begin;
  if (idx >= (long)m_part * c4) return;
  const int mi = (int)(idx / c4), n = (int)(idx % c4) * 4;
  const int m = p.m_part_begin + mi;
  float4 v = *reinterpret_cast<const float4*>(p.partial + (size_t)mi * p.Cout + n);
  for (int z = 1; z < p.k_split; ++z) {
    const float4 q = *reinterpret_cast<const float4*>(p.partial + ((size_t)z * m_part + mi) * p.Cout + n);
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
  }
  const int wo = m % p.Wo;
  const int t = m / p.Wo;
  const int ho = t % p.Ho;
  const int nn = t / p.Ho;
  const size_t off = ((((size_t)nn * p.Hop) + ho + p.out_border) * p.Wop + wo + p.out_border) * p.Cout + n;
  if (p.bias) {
    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (p.residual) {
    const float4 r = *reinterpret_cast<const float4*>(p.residual + off);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  if (p.y) *reinterpret_cast<float4*>(p.y + off) = v;
  if (p.y_act) {
    const float4 sc = *reinterpret_cast<const float4*>(p.act_scale + n), sh = *reinterpret_cast<const float4*>(p.act_shift + n);
    float4 a;
    a.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); a.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
    a.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); a.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
    *reinterpret_cast<float4*>(p.y_act + off) = a;
  }
}

// resident workgroup slots of the device (2 per CU: LDS-limited); set by mp_conv2d_nhwc before the first launch
static int g_resident_workgroups = 0;

template <int BM, int BN, int WM, int WN, int VARIANT, bool RAGGED = false>
static int launch_splitk(const ConvParams& p, hipStream_t s, double alg_k) {
  ConvParams q = p;
  q.n_mblocks = ceil_div(p.M, BM);
  q.n_nblocks = ceil_div(p.Cout, BN);
  const size_t lds = (size_t)(2 * BM * LDS_LD + 2 * BN * LDS_LD) * sizeof(float) + BM * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    MP_CHECK_HIP(hipFuncSetAttribute((const void*)conv_nhwc_f32_mfma<BM, BN, WM, WN, VARIANT, RAGGED, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const double m_part = (double)(p.M - p.m_part_begin);
  {
    ProfScope prof(BN == 64 ? "conv_nhwc_f32_mfma<128,64,64,32>/splitk" : "conv_nhwc_f32_mfma<128,128,64,64>/splitk",
                   2.0 * m_part * p.Cout * alg_k,
                   4.0 * (m_part * p.stride * p.stride * p.C + (double)p.n_chunks * BK * p.Cout + m_part * p.Cout), s);
    hipLaunchKernelGGL((conv_nhwc_f32_mfma<BM, BN, WM, WN, VARIANT, RAGGED, true>),
                       dim3(q.n_mblocks * q.n_nblocks - q.tile_begin, q.k_split), dim3(256), lds, s, q);
  }
  ProfScope prof("conv_splitk_reduce", 0.0, 4.0 * m_part * p.Cout * (q.k_split + 1), s);
  hipLaunchKernelGGL(conv_splitk_reduce, dim3(ceil_div((long)m_part * (p.Cout / 4), 256L)), dim3(256), 0, s, q);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

template <int BM, int BN, int WM, int WN, int VARIANT, bool RAGGED = false, bool AHALF = false>
static int launch(const ConvParams& p, hipStream_t s, double alg_k, int n_tiles_main = 0) {
  ConvParams q = p;
  q.n_mblocks = ceil_div(p.M, BM);
  q.n_nblocks = ceil_div(p.Cout, BN);
  constexpr int NBUF = 2;
  constexpr int LDT = LDS_LD;
  // MP_CONV_LDS_PAD_KB (tuning experiment): extra, unused LDS per workgroup -- e.g. 30 forces ONE workgroup per CU, which separates the
  // main loop's own efficiency from the interplay of two co-resident workgroups (scripts/conv_slope.py)
  static const size_t lds_pad = getenv("MP_CONV_LDS_PAD_KB") ? (size_t)atoi(getenv("MP_CONV_LDS_PAD_KB")) * 1024 : 0;
  const size_t lds = (size_t)(NBUF * BM * LDT + NBUF * BN * LDT) * sizeof(float) + BM * sizeof(int) + lds_pad;
  static bool attr_set = false;
  if (!attr_set) {
    MP_CHECK_HIP(hipFuncSetAttribute((const void*)conv_nhwc_f32_mfma<BM, BN, WM, WN, VARIANT, RAGGED, false, AHALF>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  // n_tiles_main > 0: only the first n_tiles_main tiles (whole rounds of resident workgroups); a split-K launch covers the rest
  const int n_tiles = n_tiles_main > 0 ? n_tiles_main : q.n_mblocks * q.n_nblocks;
  const double m_here = n_tiles_main > 0 ? (double)(n_tiles_main / q.n_nblocks) * BM : (double)p.M;
  dim3 grid(n_tiles);
  if constexpr ((VARIANT & 16384) != 0) {   // persistent: one workgroup per resident slot, each walks its share of the tiles
    q.k_split = n_tiles;
    grid = dim3(std::min(n_tiles, g_resident_workgroups > 0 ? g_resident_workgroups : 512));
  }
  // algorithmic work of this launch: 2*MACs over the REAL (unpadded) reduction length; bytes = input + weights + output once
  static const bool detail = getenv("MP_PROF_DETAIL") != nullptr;  // tuning aid: one profiler row per layer shape
  const char* pname = BN == 64 ? "conv_nhwc_f32_mfma<128,64,64,32>" : "conv_nhwc_f32_mfma<128,128,64,64>";
  if (AHALF) pname = "conv_nhwc_f32_mfma<128,64,64,32>/x_f16";   // own profiler row: never mixed into the fp32 kernels' roofline figures
  if (detail) {
    static std::vector<std::string> names;  // stable storage for the profiler's name pointers
    char buf[96];
    snprintf(buf, sizeof(buf), "%s/k%d_c%d_s%d_n%d", pname, p.KH, p.C, p.stride, p.Cout);
    bool found = false;
    for (auto& n : names) if (n == buf) { pname = n.c_str(); found = true; break; }
    if (!found) { names.reserve(64); names.emplace_back(buf); pname = names.back().c_str(); }
  }
  ProfScope prof(pname, 2.0 * m_here * p.Cout * alg_k,
                 (AHALF ? 2.0 : 4.0) * m_here * p.stride * p.stride * p.C + 4.0 * ((double)p.n_chunks * BK * p.Cout + m_here * p.Cout), s);
  hipLaunchKernelGGL((conv_nhwc_f32_mfma<BM, BN, WM, WN, VARIANT, RAGGED, false, AHALF>), grid, dim3(256), lds, s, q);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

// The packed weight blob is tiled for a fixed BN per layer: BN = 64 when Cout == 64, else 128.
static inline int conv_bn_tile(int Cout) { return Cout <= 64 ? 64 : 128; }

}  // namespace mp

using namespace mp;

extern "C" int mp_conv_clock_read(double* shader_mhz, int reset) {
  MP_REQUIRE(shader_mhz != nullptr, "mp_conv_clock_read: null output");
  unsigned long long h[2] = {0, 0};
  MP_CHECK_HIP(hipDeviceSynchronize());
  MP_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_conv_clk), sizeof(h)));
  *shader_mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
  if (reset) {
    const unsigned long long z[2] = {0, 0};
    MP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_clk), z, sizeof(z)));
  }
  return MP_OK;
}

#ifdef MP_RASTER_PROF
extern "C" int mp_conv_prof_read(unsigned long long* out2, int reset) {
  MP_CHECK_HIP(hipDeviceSynchronize());
  MP_CHECK_HIP(hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_conv_clk), 2 * sizeof(unsigned long long)));
  MP_CHECK_HIP(hipMemcpyFromSymbol(out2 + 2, HIP_SYMBOL(g_conv_seg), 8 * sizeof(unsigned long long)));   // out2 holds 10 words
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    MP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_clk), z, 2 * sizeof(unsigned long long)));
    MP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_seg), z, sizeof(z)));
  }
  return MP_OK;
}
#endif

extern "C" size_t mp_conv_packed_floats(int Cin_p, int Cout, int KH, int KW) {
  const int BN = conv_bn_tile(Cout);
  const int nblk = ceil_div(Cout, BN);
  const int n_chunks = ceil_div((long)KH * KW * Cin_p, BK);
  return (size_t)nblk * n_chunks * BN * BK;
}

// packed[nb][chunk][n_local][k], K index = kh*(KW*Cin_p) + kw*Cin_p + c over the concatenated row runs (zero padded)
extern "C" int mp_conv_pack_weights(const float* w, int Cout, int Cin, int KH, int KW, int Cin_p,
                                    const float* scale, float* packed) {
  MP_REQUIRE(w && packed && Cin_p >= Cin && (Cin_p % 4) == 0, "mp_conv_pack_weights: bad arguments");
  const int BN = conv_bn_tile(Cout);
  const int nblk = ceil_div(Cout, BN);
  const int run = KW * Cin_p;
  const int k_total = KH * run;
  const int n_chunks = ceil_div(k_total, BK);
  const size_t total = mp_conv_packed_floats(Cin_p, Cout, KH, KW);
  memset(packed, 0, total * sizeof(float));
  for (int nb = 0; nb < nblk; ++nb)
    for (int ch = 0; ch < n_chunks; ++ch) {
      float* tile = packed + ((size_t)nb * n_chunks + ch) * BN * BK;
      for (int nl = 0; nl < BN; ++nl) {
        const int n = nb * BN + nl;
        if (n >= Cout) continue;
        const float s = scale ? scale[n] : 1.f;
        for (int k = 0; k < BK; ++k) {
          const int kidx = ch * BK + k;
          if (kidx >= k_total) continue;
          const int kh = kidx / run, jj = kidx % run;
          const int kw = jj / Cin_p, c = jj % Cin_p;
          if (c >= Cin) continue;
          tile[nl * BK + k] = w[(((size_t)n * Cin + c) * KH + kh) * KW + kw] * s;
        }
      }
    }
  return MP_OK;
}

// How a conv launch is laid out on the chip (host-side decision, exposed through mp_conv2d_plan for the CPU tests):
//   mode 0: one single-pass launch;
//   mode 1: small grid (< 192 tiles: the released K = 1 / K = 5 refiner passes) -- every tile's K loop is split over k_split
//           workgroups and reduced deterministically;
//   mode 2: full rounds + split-K tail -- with `resident` workgroups (2 per CU) a grid of n_tiles runs ceil(n_tiles / resident)
//           rounds; when the last round is less than half full, its tiles are split along K over the idle slots instead
//           (layer 3 at 576 rows: 2700 tiles = 5.27 rounds -> 5 rounds + a third of a round).
struct ConvPlan {
  int mode, k_split, chunks_per_split, n_main, m_begin;
};

static ConvPlan plan_conv(const ConvParams& p, bool small, long ws_floats, int resident, bool splitk_on, bool tail_on) {
  ConvPlan pl = {0, 1, p.n_chunks, 0, 0};
  const int n_nb = ceil_div(p.Cout, small ? 64 : 128);
  const int n_tiles = ceil_div(p.M, 128) * n_nb;
  if (!splitk_on || ws_floats <= 0 || (p.Cout % 4) != 0) return pl;
  if (n_tiles < 192 && p.n_chunks >= 8) {
    long S = std::min<long>(ceil_div(512, n_tiles), p.n_chunks / 4);
    S = std::min<long>(S, ws_floats / ((long)p.M * p.Cout));
    if (S >= 2) {
      pl.mode = 1;
      pl.chunks_per_split = ceil_div(p.n_chunks, (int)S);
      pl.k_split = ceil_div(p.n_chunks, pl.chunks_per_split);
      return pl;
    }
  }
  const int n_tail = n_tiles % resident;
  if (tail_on && n_tiles > resident && n_tail > 0 && 2 * n_tail <= resident && p.run % BK == 0 && resident % n_nb == 0) {
    const int n_main = n_tiles - n_tail;
    const int m_begin = (n_main / n_nb) * 128;
    long S = std::min<long>(resident / n_tail, p.n_chunks / 6);
    S = std::min<long>(S, ws_floats / ((long)(p.M - m_begin) * p.Cout));
    if (S >= 2) {
      pl.mode = 2;
      pl.chunks_per_split = ceil_div(p.n_chunks, (int)S);
      pl.k_split = ceil_div(p.n_chunks, pl.chunks_per_split);
      pl.n_main = n_main;
      pl.m_begin = m_begin;
    }
  }
  return pl;
}

static int make_params(const mp_conv_desc* d, ConvParams* p) {
  MP_REQUIRE(d && d->d_x && d->d_w && (d->d_y || d->d_y_act), "mp_conv2d_nhwc: null pointer");
  MP_REQUIRE(d->C % 4 == 0, "mp_conv2d_nhwc: C (%d) must be a multiple of 4", d->C);
  MP_REQUIRE(d->in_border >= d->pad, "mp_conv2d_nhwc: in_border %d < pad %d", d->in_border, d->pad);
  MP_REQUIRE(d->stride >= 1 && d->KH >= 1 && d->KW >= 1 && d->Cout >= 1, "mp_conv2d_nhwc: bad geometry");
  MP_REQUIRE(!d->d_y_act || (d->d_act_scale && d->d_act_shift), "mp_conv2d_nhwc: y_act needs scale/shift");
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  const long M = (long)d->N * Ho * Wo;
  MP_REQUIRE(M > 0 && M < (1L << 31), "mp_conv2d_nhwc: M out of range");
  const long out_elems = (long)d->N * (Ho + 2 * d->out_border) * (Wo + 2 * d->out_border) * d->Cout;
  MP_REQUIRE(out_elems < (1L << 31), "mp_conv2d_nhwc: output too large for 32-bit offsets (%ld)", out_elems);
  p->x = d->d_x;
  p->w = d->d_w;
  p->bias = d->d_bias;
  p->residual = d->d_residual;
  p->act_scale = d->d_act_scale;
  p->act_shift = d->d_act_shift;
  p->y = d->d_y;
  p->y_act = d->d_y_act;
  p->M = (int)M;
  p->Ho = Ho;
  p->Wo = Wo;
  p->Hp = d->H + 2 * d->in_border;
  p->Wp = d->W + 2 * d->in_border;
  p->C = d->C;
  p->in_off = d->in_border - d->pad;
  p->stride = d->stride;
  p->Cout = d->Cout;
  p->Hop = Ho + 2 * d->out_border;
  p->Wop = Wo + 2 * d->out_border;
  p->out_border = d->out_border;
  p->KH = d->KH;
  p->run = d->KW * d->C;
  p->n_chunks = ceil_div((long)d->KH * p->run, BK);
  p->relu = d->relu;
  p->partial = nullptr;
  p->k_split = 1;
  p->chunks_per_split = p->n_chunks;
  p->tile_begin = 0;
  p->m_part_begin = 0;
  return MP_OK;
}

extern "C" int mp_conv2d_nhwc(const mp_conv_desc* d, mp_stream stream) {
  ConvParams p;
  int rc = make_params(d, &p);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const double alg_k = (double)d->KH * d->KW * (d->c_real > 0 ? d->c_real : d->C);
  static const int variant = getenv("MP_CONV_VARIANT") ? atoi(getenv("MP_CONV_VARIANT")) : 8449;  // default: buffer loads, LDS store under the 3rd of 4 MFMA groups; others = A/B timing
  const bool small = conv_bn_tile(d->Cout) == 64;
  if (d->x_f16) {   // half-precision input (the stems of the "fp16 renders" mode): single-pass launches of the 128x64 tile only
    MP_REQUIRE(small, "mp_conv2d_nhwc: x_f16 is implemented for Cout <= 64 (the stem convolutions), got Cout = %d", d->Cout);
    return p.run % BK != 0 ? launch<128, 64, 64, 32, 8449, true, true>(p, s, alg_k) : launch<128, 64, 64, 32, 8449, false, true>(p, s, alg_k);
  }
  static const int splitk_on = getenv("MP_CONV_SPLITK") ? atoi(getenv("MP_CONV_SPLITK")) : 1;
  static const int tail_on = getenv("MP_CONV_TAIL") ? atoi(getenv("MP_CONV_TAIL")) : 1;
  static int resident = 0;
  if (!resident) {
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    resident = 2 * n_cu;
    g_resident_workgroups = resident;
  }
  const ConvPlan plan = plan_conv(p, small, d->d_splitk_ws ? d->splitk_ws_floats : 0, resident, splitk_on != 0, tail_on != 0);
  if (plan.mode == 1) {  // small grid: every tile split along K
    p.chunks_per_split = plan.chunks_per_split;
    p.k_split = plan.k_split;
    p.partial = d->d_splitk_ws;
    if (p.run % BK != 0) return small ? launch_splitk<128, 64, 64, 32, 8193, true>(p, s, alg_k) : launch_splitk<128, 128, 64, 64, 8193, true>(p, s, alg_k);
    return small ? launch_splitk<128, 64, 64, 32, 8193>(p, s, alg_k) : launch_splitk<128, 128, 64, 64, 8193>(p, s, alg_k);
  }
  // The product library holds ONE schedule (8449).  The alternatives measured against it on the MI355X -- persistent workgroups (24833),
  // the LDS-transposed 16-byte epilogue (73985), both (90369), 64-bit global_load addressing (257): all bit-identical, all within +-1 %
  // (profiles/r03_conv_ab_epilogue_persistent.txt) -- are compiled only into MP_CONV_EXPERIMENTS builds (scripts/microbench).
#ifdef MP_CONV_EXPERIMENTS
  const bool vec = (variant & 65536) != 0;   // LDS-transposed 16-byte epilogue (A/B against the dword epilogue)
#else
  (void)variant;
#endif
  if (plan.mode == 2) {  // whole rounds single-pass, the tiles of the half-empty last round split along K
    int rc2;
#ifdef MP_CONV_EXPERIMENTS
    if (vec) rc2 = small ? launch<128, 64, 64, 32, 73985>(p, s, alg_k, plan.n_main) : launch<128, 128, 64, 64, 73985>(p, s, alg_k, plan.n_main);
    else
#endif
    rc2 = small ? launch<128, 64, 64, 32, 8449>(p, s, alg_k, plan.n_main) : launch<128, 128, 64, 64, 8449>(p, s, alg_k, plan.n_main);
    if (rc2) return rc2;
    p.chunks_per_split = plan.chunks_per_split;
    p.k_split = plan.k_split;
    p.partial = d->d_splitk_ws;
    p.tile_begin = plan.n_main;
    p.m_part_begin = plan.m_begin;
    return small ? launch_splitk<128, 64, 64, 32, 8193>(p, s, alg_k) : launch_splitk<128, 128, 64, 64, 8193>(p, s, alg_k);
  }
  if (p.run % BK != 0) {  // ragged K (stems): per-lane K bookkeeping
#ifdef MP_CONV_EXPERIMENTS
    if (vec) return small ? launch<128, 64, 64, 32, 73985, true>(p, s, alg_k) : launch<128, 128, 64, 64, 73985, true>(p, s, alg_k);
#endif
    return small ? launch<128, 64, 64, 32, 8449, true>(p, s, alg_k) : launch<128, 128, 64, 64, 8449, true>(p, s, alg_k);
  }
#ifdef MP_CONV_EXPERIMENTS
  switch (variant) {  // every variant computes the same result; the others are kept for A/B timing
    case 257: return small ? launch<128, 64, 64, 32, 257>(p, s, alg_k) : launch<128, 128, 64, 64, 257>(p, s, alg_k);  // the default schedule with global_load (64-bit lane addresses)
    case 24833: return small ? launch<128, 64, 64, 32, 24833>(p, s, alg_k) : launch<128, 128, 64, 64, 24833>(p, s, alg_k);  // 8449 with persistent workgroups
    case 73985: return small ? launch<128, 64, 64, 32, 73985>(p, s, alg_k) : launch<128, 128, 64, 64, 73985>(p, s, alg_k);  // 8449 + 16-byte epilogue
    case 90369: return small ? launch<128, 64, 64, 32, 90369>(p, s, alg_k) : launch<128, 128, 64, 64, 90369>(p, s, alg_k);  // persistent + 16-byte epilogue
    default: break;
  }
#endif
  return small ? launch<128, 64, 64, 32, 8449>(p, s, alg_k) : launch<128, 128, 64, 64, 8449>(p, s, alg_k);
}

extern "C" int mp_conv2d_plan(const mp_conv_desc* d, int n_cu, int32_t* out5) {
  MP_REQUIRE(out5 && n_cu > 0, "mp_conv2d_plan: bad arguments");
  ConvParams p;
  int rc = make_params(d, &p);
  if (rc) return rc;
  ConvPlan pl = plan_conv(p, conv_bn_tile(d->Cout) == 64, d->d_splitk_ws ? d->splitk_ws_floats : 0, 2 * n_cu, true, true);
  if (d->x_f16) pl = ConvPlan{0, 1, p.n_chunks, 0, 0};   // half-precision inputs always run as one single-pass launch
  out5[0] = pl.mode; out5[1] = pl.k_split; out5[2] = pl.chunks_per_split; out5[3] = pl.n_main; out5[4] = pl.m_begin;
  return MP_OK;
}

extern "C" const char* mp_conv2d_kernel_name(const mp_conv_desc* d) {
  return conv_bn_tile(d->Cout) == 64 ? "conv_nhwc_f32_mfma<128,64,64,32>" : "conv_nhwc_f32_mfma<128,128,64,64>";
}
