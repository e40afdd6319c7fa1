// Which ingredient of the conv K loop costs the cycles a register-only MFMA loop does not?
// One "chunk" = 64 x v_mfma_f32_32x32x2_f32 per wave (4 k-groups x 16, four accumulators), exactly the conv's 128x128 tile loop, with
// ingredients switched on one at a time (PARTS bits):
//   1  fragments come from LDS (16 ds_read_b128 per chunk and wave, [row][36] float layout, read one k-group ahead)
//   2  8 ds_write_b128 per chunk and thread (register -> other LDS buffer)
//   4  one workgroup barrier per chunk
//   8  8 global_load_dwordx4 per chunk and thread (4 scattered 128-B row pieces + 4 contiguous), consumed by the LDS writes when bit 2 is set
//      (otherwise kept alive by an empty asm at the place of the writes)
//  16  with 8: buffer_load_dwordx4 (scalar resource + 32-bit lane offsets) instead of global_load_dwordx4 (64-bit lane addresses)
//  32  with 8: only the 4 contiguous loads
// Run with 1 or 2 workgroups per CU (LDS padding).  Prints shader cycles per chunk (ideal 4096 per resident workgroup).
// Build: hipcc -O3 --offload-arch=gfx950 conv_loop_parts.hip -o conv_loop_parts
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int LDT = 36, BM = 128, BN = 128;

#define SB __builtin_amdgcn_sched_barrier(0)

template <int PARTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void loop_kernel(const float* __restrict__ gsrc, size_t gfloats,
                                                                                              float* out, unsigned long long* clk, int chunks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][BM][LDT]
  float* Bs = smem + 2 * BM * LDT;  // [2][BN][LDT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < 2 * (BM + BN) * LDT; i += 256) smem[i] = (float)((i * 7) % 13 - 6) * 0.125f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
  const int a_r0 = tid >> 3, a_c4 = tid & 7;
  // conv-like global addressing: four "pixel rows" far apart + one contiguous weight stream
  const float* ap[4];
  for (int i = 0; i < 4; ++i) ap[i] = gsrc + ((size_t)(blockIdx.x * 128 + a_r0 + 32 * i) * 2304) % (gfloats / 2) + a_c4 * 4;
  const float* bp = gsrc + gfloats / 2 + (size_t)(blockIdx.x % 2) * 65536 + (tid >> 3) * 32 + a_c4 * 4;
  // the same addresses as byte offsets from the buffer base (the source is < 4 GB)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gsrc, 0, (int)(gfloats * 4 > 0x7fffffffull ? 0x7fffffff : gfloats * 4), 0x00020000);
  int a_boff[4];
  for (int i = 0; i < 4; ++i) a_boff[i] = (int)((ap[i] - gsrc) * 4);
  const int b_boff = (int)((bp - gsrc) * 4);
  float4 g[8];
  for (int i = 0; i < 8; ++i) g[i] = make_float4(0.25f * i, 0.5f, -0.25f, 0.125f);
  float4 afr[2][2], bfr[2][2];
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 2; ++i) {
      afr[s][i] = make_float4(0.5f, -0.25f, 0.125f, 0.75f);
      bfr[s][i] = make_float4(-0.5f, 0.25f, 0.375f, -0.125f);
    }
#define FRAG_READ(SLOT, AS, BS, KOFF)                                                                      \
  if constexpr ((PARTS & 1) != 0) {                                                                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) afr[SLOT][i] = *reinterpret_cast<const float4*>((AS) + i * 32 * LDT + (KOFF));  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) bfr[SLOT][j] = *reinterpret_cast<const float4*>((BS) + j * 32 * LDT + (KOFF));  \
  }
#define MFMA_ROW(SLOT, I)                                                                                    \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                            \
    acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[SLOT][I].x, bfr[SLOT][j].x, acc[I][j], 0, 0, 0);     \
    acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[SLOT][I].y, bfr[SLOT][j].y, acc[I][j], 0, 0, 0);     \
    acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[SLOT][I].z, bfr[SLOT][j].z, acc[I][j], 0, 0, 0);     \
    acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[SLOT][I].w, bfr[SLOT][j].w, acc[I][j], 0, 0, 0);     \
  }
  const unsigned long long c0 = __builtin_readcyclecounter();
  int aoff = 0;
  for (int c = 0; c < chunks; ++c) {
    const int buf = c & 1;
    const float* as = As + buf * BM * LDT + (wm * 64 + frag_row) * LDT + frag_k;
    const float* bs = Bs + buf * BN * LDT + (wn * 64 + frag_row) * LDT + frag_k;
    const float* as_n = As + (buf ^ 1) * BM * LDT + (wm * 64 + frag_row) * LDT + frag_k;
    const float* bs_n = Bs + (buf ^ 1) * BN * LDT + (wn * 64 + frag_row) * LDT + frag_k;
    FRAG_READ(1, as, bs, 8)
    SB;
    MFMA_ROW(0, 0)
    SB;
    if constexpr ((PARTS & 2) != 0) {
      float* as_w = As + (buf ^ 1) * BM * LDT + a_r0 * LDT + a_c4 * 4;
      float* bs_w = Bs + (buf ^ 1) * BN * LDT + a_r0 * LDT + a_c4 * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<float4*>(as_w + i * 32 * LDT) = g[i];
        *reinterpret_cast<float4*>(bs_w + i * 32 * LDT) = g[4 + i];
      }
    }
    if constexpr ((PARTS & 2) == 0 && (PARTS & 8) != 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(g[i].x), "v"(g[i].y), "v"(g[i].z), "v"(g[i].w));
    }
    SB;
    MFMA_ROW(0, 1)
    SB;
    if constexpr ((PARTS & 8) != 0) {
      aoff = (aoff + 32) & 2047;
      if constexpr ((PARTS & 16) != 0) {
        if constexpr ((PARTS & 32) == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, a_boff[i] + aoff * 4, 0, 0);
            g[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, b_boff + ((c & 15) * 4096 + i * 1024) * 4, 0, 0);
          g[4 + i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
      } else {
        if constexpr ((PARTS & 32) == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) g[i] = *reinterpret_cast<const float4*>(ap[i] + aoff);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) g[4 + i] = *reinterpret_cast<const float4*>(bp + (size_t)(c & 15) * 4096 + i * 1024);
      }
    }
    SB;
    FRAG_READ(0, as, bs, 16)
    SB;
    MFMA_ROW(1, 0)
    MFMA_ROW(1, 1)
    SB;
    FRAG_READ(1, as, bs, 24)
    SB;
    MFMA_ROW(0, 0)
    MFMA_ROW(0, 1)
    SB;
    if constexpr ((PARTS & 4) != 0) __syncthreads();
    FRAG_READ(0, as_n, bs_n, 0)
    SB;
    MFMA_ROW(1, 0)
    MFMA_ROW(1, 1)
    SB;
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  for (int i = 0; i < 8; ++i) s += g[i].x;
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) atomicAdd(clk, c1 - c0);
}

template <int PARTS>
static void run(const char* what, int wgs_per_cu, const float* gsrc, size_t gfloats, float* out, unsigned long long* clk, int n_cu) {
  const int chunks = 2000;
  const size_t lds_min = (size_t)2 * (BM + BN) * LDT * sizeof(float);
  const size_t lds = wgs_per_cu == 1 ? lds_min + 20 * 1024 : lds_min;
  hipFuncSetAttribute((const void*)loop_kernel<PARTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = n_cu * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  unsigned long long cyc = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(clk, 0, sizeof(unsigned long long));
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(loop_kernel<PARTS>, dim3(grid), dim3(256), lds, 0, gsrc, gfloats, out, clk, chunks);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) {
      best = ms;
      hipMemcpy(&cyc, clk, sizeof(cyc), hipMemcpyDeviceToHost);
    }
  }
  const double tf = (double)grid * 4 * chunks * 64.0 * 4096.0 / (best * 1e-3) * 1e-12;
  printf("%d WG/CU  parts=%2d  %-58s %8.3f ms  %6.1f TFLOP/s  %7.0f cycles per chunk-round (ideal %d)\n", wgs_per_cu, PARTS, what, best, tf,
         (double)cyc / grid / chunks * wgs_per_cu, 4096 * wgs_per_cu);
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  // argv[1]: size of the global source in MiB of floats (default 64 Mi floats = 256 MB: the A stream misses L2; 1 = 4 MB: L2-resident)
  const size_t gfloats = (size_t)(argc > 1 ? atoi(argv[1]) : 64) << 20;
  float *gsrc, *out;
  unsigned long long* clk;
  float* gzero;
  hipMalloc(&gsrc, gfloats * sizeof(float));
  hipMalloc(&gzero, gfloats * sizeof(float));
  hipMemset(gzero, 0, gfloats * sizeof(float));
  {
    std::vector<float> h(gfloats);
    unsigned x = 12345u;
    for (size_t i = 0; i < gfloats; ++i) {
      x = x * 1664525u + 1013904223u;
      h[i] = (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;   // uniform [-1, 1)
    }
    hipMemcpy(gsrc, h.data(), gfloats * sizeof(float), hipMemcpyHostToDevice);
  }
  hipMalloc(&out, (size_t)n_cu * 2 * 256 * sizeof(float));
  hipMalloc(&clk, sizeof(unsigned long long));
  for (int w = 1; w <= 2; ++w) {
    run<0>("MFMA only (register operands)", w, gsrc, gfloats, out, clk, n_cu);
    run<4>("+ barrier per chunk", w, gsrc, gfloats, out, clk, n_cu);
    run<1>("+ LDS fragment reads", w, gsrc, gfloats, out, clk, n_cu);
    run<5>("+ LDS fragment reads + barrier", w, gsrc, gfloats, out, clk, n_cu);
    run<3>("+ LDS reads + LDS writes", w, gsrc, gfloats, out, clk, n_cu);
    run<7>("+ LDS reads + LDS writes + barrier", w, gsrc, gfloats, out, clk, n_cu);
    run<8>("+ global loads only (kept alive, not stored)", w, gsrc, gfloats, out, clk, n_cu);
    run<12>("+ global loads (not stored) + barrier", w, gsrc, gfloats, out, clk, n_cu);
    run<13>("+ global loads (not stored) + LDS reads + barrier", w, gsrc, gfloats, out, clk, n_cu);
    run<15 + 32>("everything, only the 4 contiguous loads", w, gsrc, gfloats, out, clk, n_cu);
    run<15 + 16>("everything, buffer_load instead of global_load", w, gsrc, gfloats, out, clk, n_cu);
    run<15>("everything (the conv loop without its address arithmetic)", w, gsrc, gfloats, out, clk, n_cu);
    run<15>("everything, all-zero global data", w, gzero, gfloats, out, clk, n_cu);
  }
  return 0;
}
