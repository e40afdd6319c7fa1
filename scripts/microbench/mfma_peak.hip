// Register-only fp32 MFMA throughput probe for gfx950: what fraction of the 157.3 TFLOP/s datasheet peak does a loop of
// v_mfma_f32_32x32x2_f32 with NO memory traffic reach?  Same accumulator pattern as the conv kernel (4 accumulators per wave,
// consecutive MFMAs alternate between two of them), 4 waves per workgroup, `blocks_per_cu` workgroups per CU.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(int blocks_per_cu, int n_cu) {
  const int iters = 4000;
  const int grid = n_cu * blocks_per_cu;
  float* out;
  hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f, 2.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 /*waves*/ * iters * 16.0 * NACC * (32.0 * 32 * 2 * 2);
  printf("accumulators/wave=%d workgroups/CU=%d (waves/SIMD=%d): %.3f ms, %.1f TFLOP/s = %.1f%% of 157.3\n", NACC, blocks_per_cu, blocks_per_cu, ms,
         flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
  hipFree(out);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  for (int bpc : {1, 2, 3}) {
    run<1>(bpc, p.multiProcessorCount);
    run<2>(bpc, p.multiProcessorCount);
    run<4>(bpc, p.multiProcessorCount);
  }
  return 0;
}
