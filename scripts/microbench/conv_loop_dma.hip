// Prototype of the conv K loop with LDS-direct loads (buffer_load_dwordx4 ... lds) into an UNPADDED, XOR-swizzled tile, against the
// register-staged loop with the padded [row][36] tile -- same MFMAs, same barriers, synthetic conv-like addresses.
//   tile 128 x BN (BN = 128: wave tile 64x64, 64 MFMAs per chunk and wave; BN = 64: wave tile 64x32, 32 MFMAs), BK = 32
//   DMA = false: 8 (6) buffer_load_dwordx4 -> VGPRs -> ds_write_b128 under the 3rd MFMA group, tiles [row][36]
//   DMA = true : 8 (6) buffer_load_dwordx4 ... lds per lane and chunk, no VGPR staging, tiles [row][32] with the 16-byte slot of
//                (row r, k-quad q) at r * 8 + (q ^ ((r >> 1) & 7)): conflict-free ds_read_b128 fragments, contiguous 1-KB DMA writes
// WGS workgroups per CU are forced through LDS padding.  Prints TFLOP/s.
// Build: hipcc -O3 --offload-arch=gfx950 conv_loop_dma.hip -o conv_loop_dma
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define SB __builtin_amdgcn_sched_barrier(0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
// LDS-direct load as inline asm: through the builtin the compiler treats the LDS write as a may-alias of every later ds_read and
// waits vmcnt(0) right after issuing the loads (the whole latency exposed every chunk); here the kernel waits itself, before the
// barrier that publishes the buffer.  m0 = LDS byte address of the wave's 1-KB destination (wave-uniform).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds_byte_addr, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#endif
}
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p; }

template <int BN, bool DMA>
__global__ __launch_bounds__(256) void loop_kernel(const float* __restrict__ gsrc, size_t gfloats, float* out, int chunks) {
  constexpr int BM = 128, TN = BN / 64, LDT = DMA ? 32 : 36, B_LD = BN / 32;
  extern __shared__ __attribute__((aligned(1024))) float smem[];
  float* As = smem;                 // [2][BM][LDT]
  float* Bs = smem + 2 * BM * LDT;  // [2][BN][LDT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < 2 * (BM + BN) * LDT; i += 256) smem[i] = (float)((i * 7) % 13 - 6) * 0.125f;
  __syncthreads();
  f32x16 acc[2][TN];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frag_row = lane & 31, hi = lane >> 5;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gsrc, 0, -1, 0x00020000);
  i32x4 rsrc4;   // the same descriptor as four scalars for the inline asm
  rsrc4.x = (int)(unsigned)(size_t)gsrc; rsrc4.y = (int)(((size_t)gsrc >> 32) & 0xffffu); rsrc4.z = -1; rsrc4.w = 0x00020000;
  // conv-like global addressing: pixel rows 9 KB apart (a 3x3 layer with 256 channels... any stride works), weights contiguous
  const size_t half = gfloats / 2;
  int a_voff[4], b_voff[B_LD];
  if constexpr (DMA) {
    // wave w, load k covers slots [(w * 4 + k) * 64, +64): lane -> (row, k-quad) through the swizzle
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int S = (wave * 4 + k) * 64 + lane, r = S >> 3, q = (S & 7) ^ ((r >> 1) & 7);
      a_voff[k] = (int)((((size_t)(blockIdx.x * 128 + r) * 2304) % half) * 4) + q * 16;
    }
#pragma unroll
    for (int k = 0; k < B_LD; ++k) {
      const int S = (wave * B_LD + k) * 64 + lane, r = S >> 3, q = (S & 7) ^ ((r >> 1) & 7);
      b_voff[k] = (int)(half * 4) + (blockIdx.x % 2) * 262144 + r * 128 + q * 16;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) a_voff[k] = (int)((((size_t)(blockIdx.x * 128 + (tid >> 3) + 32 * k) * 2304) % half) * 4) + (tid & 7) * 16;
#pragma unroll
    for (int k = 0; k < B_LD; ++k) b_voff[k] = (int)(half * 4) + (blockIdx.x % 2) * 262144 + ((tid >> 3) + 32 * k) * 128 + (tid & 7) * 16;
  }
  float4 g[8];
  for (int i = 0; i < 8; ++i) g[i] = make_float4(0.25f * i, 0.5f, -0.25f, 0.125f);
  // fragment addresses (floats) inside one buffer for k-group 0
  int a_fr, b_fr;
  if constexpr (DMA) {
    const int Ra = wm * 64 + frag_row, Rb = wn * (BN / 2) + frag_row;
    a_fr = Ra * 32 + ((hi ^ ((Ra >> 1) & 7)) << 2);
    b_fr = Rb * 32 + ((hi ^ ((Rb >> 1) & 7)) << 2);
  } else {
    a_fr = (wm * 64 + frag_row) * LDT + hi * 4;
    b_fr = (wn * (BN / 2) + frag_row) * LDT + hi * 4;
  }
  int soff = 0;
  for (int c = 0; c < chunks; ++c) {
    const int buf = c & 1;
    soff = (soff + 128) & 8191;
    // ---- next chunk: global -> (registers | LDS) ----
    if constexpr (DMA) {
      const unsigned as_w = __builtin_amdgcn_readfirstlane(lds_addr(As + (buf ^ 1) * BM * LDT + wave * 4 * 256));   // 1 KB per wave-load
      const unsigned bs_w = __builtin_amdgcn_readfirstlane(lds_addr(Bs + (buf ^ 1) * BN * LDT + wave * B_LD * 256));
#pragma unroll
      for (int k = 0; k < 4; ++k) dma16(rsrc4, as_w + k * 1024, a_voff[k], soff);
#pragma unroll
      for (int k = 0; k < B_LD; ++k) dma16(rsrc4, bs_w + k * 1024, b_voff[k], (c & 15) * 16384);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, a_voff[k], soff, 0);
        g[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
#pragma unroll
      for (int k = 0; k < B_LD; ++k) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, b_voff[k], (c & 15) * 16384, 0);
        g[4 + k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
    }
    SB;
    const float* as = As + buf * BM * LDT;
    const float* bs = Bs + buf * BN * LDT;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float4 af[2], bf[TN];
      const int ao = DMA ? (a_fr ^ (kk * 8)) : a_fr + kk * 8, bo = DMA ? (b_fr ^ (kk * 8)) : b_fr + kk * 8;
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4*>(as + ao + i * 32 * LDT);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(bs + bo + j * 32 * LDT);
      if constexpr (!DMA) {
        if (kk == 2) {
          SB;
          float* as_w = As + (buf ^ 1) * BM * LDT + (tid >> 3) * LDT + (tid & 7) * 4;
          float* bs_w = Bs + (buf ^ 1) * BN * LDT + (tid >> 3) * LDT + (tid & 7) * 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(as_w + k * 32 * LDT) = g[k];
#pragma unroll
          for (int k = 0; k < B_LD; ++k) *reinterpret_cast<float4*>(bs_w + k * 32 * LDT) = g[4 + k];
          SB;
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-direct loads have landed
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int BN, bool DMA>
static void run(int wgs_per_cu, const float* gsrc, size_t gfloats, float* out, int n_cu) {
  const int chunks = 2000;
  constexpr int LDT = DMA ? 32 : 36;
  const size_t lds_min = (size_t)2 * (128 + BN) * LDT * sizeof(float);
  // force the residency: the allocation is sized so that exactly wgs_per_cu workgroups fit into 160 KB
  size_t lds = (size_t)160 * 1024 / wgs_per_cu;
  lds -= lds % 1024;
  if (lds < lds_min) {
    printf("tile 128x%-3d %s: %d WG/CU does not fit (%zu B needed)\n", BN, DMA ? "LDS-direct" : "staged    ", wgs_per_cu, lds_min);
    return;
  }
  hipFuncSetAttribute((const void*)loop_kernel<BN, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = n_cu * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((loop_kernel<BN, DMA>), dim3(grid), dim3(256), lds, 0, gsrc, gfloats, out, chunks);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double tf = (double)grid * 4 * chunks * (16.0 * BN / 32) * 4096.0 / (best * 1e-3) * 1e-12;
  printf("tile 128x%-3d %s  %d WG/CU  %8.3f ms  %6.1f TFLOP/s\n", BN, DMA ? "LDS-direct" : "staged    ", wgs_per_cu, best, tf);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  const size_t gfloats = (size_t)8 << 20;
  float *gsrc, *out;
  hipMalloc(&gsrc, gfloats * sizeof(float) + 65536);
  hipMalloc(&out, (size_t)n_cu * 4 * 256 * sizeof(float));
  {
    std::vector<float> h(gfloats + 16384);
    unsigned x = 12345u;
    for (size_t i = 0; i < h.size(); ++i) {
      x = x * 1664525u + 1013904223u;
      h[i] = (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    hipMemcpy(gsrc, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  for (int rep = 0; rep < 2; ++rep) {
    run<128, false>(2, gsrc, gfloats, out, n_cu);
    run<128, true>(2, gsrc, gfloats, out, n_cu);
    run<64, false>(2, gsrc, gfloats, out, n_cu);
    run<64, true>(2, gsrc, gfloats, out, n_cu);
    run<64, false>(3, gsrc, gfloats, out, n_cu);
    run<64, true>(3, gsrc, gfloats, out, n_cu);
    run<64, true>(4, gsrc, gfloats, out, n_cu);
  }
  return 0;
}
