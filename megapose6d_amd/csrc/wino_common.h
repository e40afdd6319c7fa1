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
  int telemetry;        // != 0: every 64th workgroup adds its clock readings to the kernel's telemetry counters (mp_conv_wino_bf16_telemetry)
};

__device__ __forceinline__ float4 buf4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// LDS: V[2][16][64][16] floats (128 KB) during the K loop; S[4][2][64][64] floats (128 KB) in the epilogue; + the tile table.
constexpr int WV_STAGE = 16 * WT * WCK;                          // floats per V stage (64 KB)
constexpr size_t WINO_LDS_BYTES = (size_t)4 * 2 * WT * WCOUT * sizeof(float) + (size_t)WT * 2 * sizeof(int);

}  // namespace mp
