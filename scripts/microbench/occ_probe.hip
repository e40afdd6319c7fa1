// How many 256-thread workgroups does the runtime consider resident per CU for a given dynamic LDS size?  (gfx950: 160 KB LDS per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float* o) {
  extern __shared__ float s[];
  s[threadIdx.x] = 1.f;
  __syncthreads();
  o[threadIdx.x] = s[255 - threadIdx.x];
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  int v = 0;
  hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0);
  printf("sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu (attr %d), regsPerBlock %d\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, v, p.regsPerBlock);
  for (int kb = 32; kb <= 82; kb += 2) {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, (size_t)kb * 1024);
    printf("%d KB -> %d workgroups per CU\n", kb, n);
  }
  return 0;
}
