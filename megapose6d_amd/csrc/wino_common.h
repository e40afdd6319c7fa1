// Shared declarations of the fused Winograd F(2x2, 3x3) kernels (conv_wino.hip: fp32 MFMA; conv_wino_bf16.hip: exact bf16 pieces).
#pragma once
#include "common.h"

namespace mp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WT = 64;        // tiles per workgroup
constexpr int WCK = 16;       // input channels per K step (two 8-channel MFMA half-steps)
constexpr int WCOUT = 64;     // output channels per workgroup

struct WinoParams {
  const float* __restrict__ x;
  const float* __restrict__ u;
  const float* __restrict__ bias;
  const float* __restrict__ residual;
  const float* __restrict__ act_scale;
  const float* __restrict__ act_shift;
  float* __restrict__ y;
  float* __restrict__ y_act;
  int N, Ho, Wo;
  int Hp, Wp, C;        // padded input geometry
  int in_off;           // in_border - 1
  int Cout;
  int Hop, Wop, out_border;
  int tiles_x, tiles_y, n_tiles;
  int n_chunks;         // C / 8: 8-channel half-steps (the unit of the packed weights)
  int n_steps;          // C / 16
  int relu;
  int n_cblocks;        // Cout / 64
  int out_bytes;        // size of the output tensor (range check of the epilogue's buffer accesses)
  unsigned mg_tx, sh_tx, mg_ty, sh_ty, mg_cb, sh_cb;   // bf16 kernel: magic numbers of the divisions by tiles_x / tiles_y / n_cblocks (wino_fastdiv)
  int n_units;          // bf16 kernel, persistent form: units (= workgroups of the plain launch) the resident workgroups walk
  int telemetry;        // != 0: every 64th workgroup adds its clock readings to the kernel's telemetry counters (mp_conv_wino_bf16_telemetry)
};

__device__ __forceinline__ float4 buf4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Division of a 32-bit unsigned by a launch constant d >= 1 (Granlund / Montgomery round-up method, exact for every 32-bit numerator):
// host: wino_fastdiv_make(d, &M, &s); device: q = wino_fastdiv(t, d, M, s).  Seven vector instructions with the remainder instead of the
// ~25 of a generic division: the tile -> (image, row, column) arithmetic sits in front of a workgroup's first memory request.
static inline void wino_fastdiv_make(unsigned d, unsigned* M, unsigned* s) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                       // l = ceil(log2 d)
  *M = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  *s = l;
}
__device__ __forceinline__ unsigned wino_fastdiv(unsigned t, unsigned d, unsigned M, unsigned s) {
  const unsigned t1 = __umulhi(t, M);
  const unsigned q = (t1 + ((t - t1) >> 1)) >> (s - 1);   // (s >= 1 whenever d >= 2)
  return d == 1 ? t : q;
}

// LDS: V[2][16][64][16] floats (128 KB) during the K loop; S[4][2][64][64] floats (128 KB) in the epilogue; + the tile table.
constexpr int WV_STAGE = 16 * WT * WCK;                          // floats per V stage (64 KB)
constexpr size_t WINO_LDS_BYTES = (size_t)4 * 2 * WT * WCOUT * sizeof(float) + (size_t)WT * 2 * sizeof(int);

}  // namespace mp
