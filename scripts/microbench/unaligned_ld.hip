// Probe: do 16-byte global loads from 4-byte-aligned (not 16-byte-aligned) addresses work on gfx950, and what do they cost?
// (Needed to decide whether the 9-channel coarse stem can read its input with pixel stride 9 floats instead of 12.)
// Build: hipcc --offload-arch=gfx950 -O3 -o unaligned_ld unaligned_ld.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void ld_kernel(const float* __restrict__ src, float* __restrict__ dst, int off, long n4, int stride_floats) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float* p = src + off + i * stride_floats;
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    acc += v.x + 2.f * v.y + 3.f * v.z + 4.f * v.w;
  }
  dst[(long)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  const long n = 1L << 28;  // 1 GiB of floats / 4
  float *src, *dst;
  hipMalloc(&src, (n + 64) * sizeof(float));
  const int grid = 256 * 8, block = 256;
  hipMalloc(&dst, (long)grid * block * sizeof(float));
  std::vector<float> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977);
  for (long o = 0; o < n + 64; o += (long)h.size()) {
    const long m = std::min<long>((long)h.size(), n + 64 - o);
    hipMemcpy(src + o, h.data(), m * sizeof(float), hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int stride : {4, 9}) {      // 4: dense float4 stream; 9: one float4 per 9-float "pixel" (the stem's access pattern)
    for (int off : {0, 1, 2, 3}) {
      const long n4 = (n - 16) / stride;
      // correctness on a small prefix
      hipLaunchKernelGGL(ld_kernel, dim3(1), dim3(64), 0, 0, src, dst, off, 64L, stride);
      std::vector<float> got(64);
      hipMemcpy(got.data(), dst, 64 * sizeof(float), hipMemcpyDeviceToHost);
      int bad = 0;
      for (int l = 0; l < 64; ++l) {
        const long b = off + (long)l * stride;
        const float want = h[b % h.size()] + 2.f * h[(b + 1) % h.size()] + 3.f * h[(b + 2) % h.size()] + 4.f * h[(b + 3) % h.size()];
        if (got[l] != want) ++bad;
      }
      hipLaunchKernelGGL(ld_kernel, dim3(grid), dim3(block), 0, 0, src, dst, off, n4, stride);
      hipEventRecord(e0);
      hipLaunchKernelGGL(ld_kernel, dim3(grid), dim3(block), 0, 0, src, dst, off, n4, stride);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("stride %d floats, offset %d floats: %s, %.3f ms, %.1f GB/s useful\n", stride, off, bad ? "WRONG VALUES" : "values ok", ms,
             (double)n4 * 16 / ms / 1e6);
    }
  }
  return 0;
}
