// Is the fp32 MFMA rate on MI355X limited by POWER (clock give-back) once the operands toggle like real data?
// Register-only loops of v_mfma_f32_32x32x2_f32 (2 workgroups of 4 waves per CU, 4 accumulators per wave) with
//   mode 0: operands = small constants (the round-1 peak probe), mode 1: all-zero operands, mode 2: random operands in [-1, 1)
//   that change every instruction (8 + 8 registers rotated), mode 3: as 2 but N(0,1)-like magnitudes.
// Reports TFLOP/s and the effective shader clock = delta(s_memtime) / delta(s_memrealtime, 100 MHz).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ ops, float* out, unsigned long long* clk, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int k = 0; k < 8; ++k) {
    a[k] = ops[(k * 256 + threadIdx.x) * 2 + 0];
    b[k] = ops[(k * 256 + threadIdx.x) * 2 + 1];
  }
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 7], b[(u + 3 * i) & 7], acc[i], 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = c1 - c0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount * 2, iters = 6000;
  float *ops, *out;
  unsigned long long* clk;
  hipMalloc(&ops, 8 * 256 * 2 * sizeof(float));
  hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
  hipMalloc(&clk, (size_t)grid * 2 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[4] = {"small integer constants", "all zero", "random uniform [-1,1)", "random, sum of 3 uniforms (bell shaped)"};
  const int order[] = {0, 1, 2, 3, 3, 2, 1, 0, 0, 2, 0, 2};
  for (int oi = 0; oi < 12; ++oi) {
    const int mode = order[oi];
    std::vector<float> h(8 * 256 * 2);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
      const float u = 2.f * rand() / (float)RAND_MAX - 1.f;
      const float g = (2.f * rand() / (float)RAND_MAX - 1.f) + (2.f * rand() / (float)RAND_MAX - 1.f) + u;
      h[i] = mode == 0 ? (float)(1 + (int)(i % 5)) : mode == 1 ? 0.f : mode == 2 ? u : g;
    }
    hipMemcpy(ops, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, ops, out, clk, rep == 0 ? 200 : iters);
      hipDeviceSynchronize();
    }
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, ops, out, clk, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(grid * 2);
    hipMemcpy(hc.data(), clk, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    for (int g = 0; g < grid; ++g) { cs += (double)hc[2 * g]; rs += (double)hc[2 * g + 1]; }
    const double flops = (double)grid * 4 * iters * 8.0 * 4 * (32.0 * 32 * 2 * 2);
    printf("mode %d (%s): %.3f ms, %.1f TFLOP/s = %.1f%% of 157.3; shader clock / 100 MHz realtime clock = %.3f -> %.0f MHz effective\n", mode, names[mode],
           ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100, cs / rs, cs / rs * 100.0);
  }
  return 0;
}
