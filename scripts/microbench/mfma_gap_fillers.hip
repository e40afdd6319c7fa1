// What does an instruction cost when it sits in the gap between two v_mfma_f32_32x32x16_bf16 of ONE wave per SIMD (the situation of the
// bf16x9 Winograd K loop: 512-register kernel, in-order issue, nothing else resident)?
// Loop body = 16 x { MFMA (rotating over 16 accumulators, as the Winograd loop does); K filler instructions of one kind }, all inline asm
// between sched_barriers.  Prints shader cycles per MFMA (s_memtime of wave 0) for K = 0..8 and every kind: the floor is 32 (8 passes x 4).
// Also checks the candidate "round-to-nearest" three-piece split  p1 = cvt_pk_bf16(v), r = dot2_bf16(p1, -1, v), p2 = cvt_pk_bf16(r), ...
// for exactness (p1 + p2 + p3 == v in double) on random and extreme fp32 inputs.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_gap_fillers.hip -o _build/mfma_gap_fillers
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SB __builtin_amdgcn_sched_barrier(0);

enum Kind { AND = 0, SUB, PKADD, PERM, CVTPK, DOT2, DSW128_AND, DSR128_AND, DEP_AND_SUB, BUFLD_AND, SALU, N_KINDS };
static const char* KIND_NAME[N_KINDS] = {"v_and_b32 (independent)", "v_sub_f32 (independent)", "v_pk_add_f32", "v_perm_b32", "v_cvt_pk_bf16_f32",
                                         "v_dot2_f32_bf16", "1 ds_write_b128 + (K-1) v_and", "1 ds_read_b128 + (K-1) v_and",
                                         "dependent pairs: v_and -> v_sub", "1 buffer_load_dwordx4 + (K-1) v_and", "s_add_i32"};

template <int KIND, int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gap_kernel(const float* __restrict__ src, float* out,
                                                                                             unsigned long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc[16];
  for (int a = 0; a < 16; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  u32x4 A = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, B = {0x3f003f00u, 0x3f003f00u, 0x3f003f00u, 0x3f003f00u};
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = src[(tid * 16 + i) & 4095];
  f32x2 p[4];
  for (int i = 0; i < 4; ++i) p[i] = f32x2{f[2 * i], f[2 * i + 1]};
  u32x4 wdata = {1u, 2u, 3u, 4u}, rdata = {0u, 0u, 0u, 0u}, gdata = {0u, 0u, 0u, 0u};
  float* lds_w = smem + tid * 4;
  const float* lds_r = smem + 4096 + lane * 4;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4096 * 4, 0x00020000);
  int sacc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[s]) : "v"(A), "v"(B));
      SB
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int q = (s * K + k) & 15;
        if constexpr (KIND == AND) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(f[q]));
        if constexpr (KIND == SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[q]) : "v"(f[(q + 5) & 15]));
        if constexpr (KIND == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[q & 3]) : "v"(p[(q + 1) & 3]));
        if constexpr (KIND == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(f[q]) : "v"(f[(q + 5) & 15]), "s"(0x07060302u));
        if constexpr (KIND == CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[q]) : "v"(f[(q + 5) & 15]));
        if constexpr (KIND == DOT2) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(f[q]) : "v"(f[(q + 5) & 15]), "s"(0x0000bf80u));
        if constexpr (KIND == DSW128_AND) {
          if (k == 0) { *reinterpret_cast<u32x4*>(lds_w + ((s & 3) * 1024)) = wdata; }
          else asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(f[q]));
        }
        if constexpr (KIND == DSR128_AND) {
          if (k == 0) { rdata ^= *reinterpret_cast<const u32x4*>(lds_r + ((s & 3) * 1024)); }
          else asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(f[q]));
        }
        if constexpr (KIND == DEP_AND_SUB) {
          if ((k & 1) == 0) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(f[(q + 8) & 15]) : "v"(f[q]));
          else asm volatile("v_sub_f32 %0, %1, %0" : "+v"(f[(q + 7) & 15]) : "v"(f[(q + 15) & 15]));
        }
        if constexpr (KIND == BUFLD_AND) {
          if (k == 0) { if ((s & 3) == 0) gdata ^= __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, (s & 12) * 256, 0); }
          else asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(f[q]));
        }
        if constexpr (KIND == SALU) asm volatile("s_add_i32 %0, %0, 3" : "+s"(sacc));
      }
      SB
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) atomicAdd(clk, t1 - t0);
  float sum = 0.f;
  for (int a = 0; a < 16; ++a) sum += acc[a][lane & 15];
  for (int i = 0; i < 16; ++i) sum += f[i];
  for (int i = 0; i < 4; ++i) sum += p[i].x + p[i].y;
  sum += (float)(rdata.x ^ rdata.y ^ rdata.z ^ rdata.w ^ gdata.x ^ gdata.w) * 1e-30f + (float)sacc * 1e-30f;
  if (sum == 123.456f) out[tid] = sum;
}

// the candidate split, 2 elements per thread: pieces as bf16 bit patterns
__global__ void split_rne_kernel(const float* v, unsigned* pieces, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float a = v[2 * i], b = v[2 * i + 1];
  unsigned p1, p2, p3;
  float ra, rb, qa, qb;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(a), "v"(b));
  asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(ra) : "v"(p1), "s"(0x0000bf80u), "v"(a));
  asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(rb) : "v"(p1), "s"(0xbf800000u), "v"(b));
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p2) : "v"(ra), "v"(rb));
  asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(qa) : "v"(p2), "s"(0x0000bf80u), "v"(ra));
  asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(qb) : "v"(p2), "s"(0xbf800000u), "v"(rb));
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p3) : "v"(qa), "v"(qb));
  pieces[3 * i] = p1; pieces[3 * i + 1] = p2; pieces[3 * i + 2] = p3;
}

template <int KIND, int K>
static double run_one(const float* d_src, float* d_out, unsigned long long* d_clk, int n_cu) {
  const int iters = 200;
  hipFuncSetAttribute((const void*)gap_kernel<KIND, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipMemset(d_clk, 0, 8);
  hipLaunchKernelGGL((gap_kernel<KIND, K>), dim3(n_cu), dim3(256), 100 * 1024, 0, d_src, d_out, d_clk, 20);
  hipMemset(d_clk, 0, 8);
  hipLaunchKernelGGL((gap_kernel<KIND, K>), dim3(n_cu), dim3(256), 100 * 1024, 0, d_src, d_out, d_clk, iters);
  hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, d_clk, 8, hipMemcpyDeviceToHost);
  return (double)c / n_cu / (iters * 16.0);
}

template <int KIND>
static void run_kind(const float* d_src, float* d_out, unsigned long long* d_clk, int n_cu) {
  double r[9];
  r[0] = run_one<KIND, 0>(d_src, d_out, d_clk, n_cu); r[1] = run_one<KIND, 1>(d_src, d_out, d_clk, n_cu);
  r[2] = run_one<KIND, 2>(d_src, d_out, d_clk, n_cu); r[3] = run_one<KIND, 3>(d_src, d_out, d_clk, n_cu);
  r[4] = run_one<KIND, 4>(d_src, d_out, d_clk, n_cu); r[5] = run_one<KIND, 5>(d_src, d_out, d_clk, n_cu);
  r[6] = run_one<KIND, 6>(d_src, d_out, d_clk, n_cu); r[7] = run_one<KIND, 7>(d_src, d_out, d_clk, n_cu);
  r[8] = run_one<KIND, 8>(d_src, d_out, d_clk, n_cu);
  printf("GAP %-38s | cycles per MFMA at K = 0..8 fillers:", KIND_NAME[KIND]);
  for (int k = 0; k < 9; ++k) printf(" %5.1f", r[k]);
  printf("\n");
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  std::vector<float> h(4096);
  std::mt19937 rng(1);
  std::normal_distribution<float> G(0.f, 1.f);
  for (auto& x : h) x = G(rng);
  float *d_src, *d_out;
  unsigned long long* d_clk;
  hipMalloc(&d_src, 4096 * 4); hipMalloc(&d_out, 4096 * 4); hipMalloc(&d_clk, 8);
  hipMemcpy(d_src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  printf("%s, %d CUs: one wave per SIMD, 16 rotating accumulators, K filler instructions after every v_mfma_f32_32x32x16_bf16\n", prop.gcnArchName, n_cu);
  run_kind<AND>(d_src, d_out, d_clk, n_cu);
  run_kind<SUB>(d_src, d_out, d_clk, n_cu);
  run_kind<PKADD>(d_src, d_out, d_clk, n_cu);
  run_kind<PERM>(d_src, d_out, d_clk, n_cu);
  run_kind<CVTPK>(d_src, d_out, d_clk, n_cu);
  run_kind<DOT2>(d_src, d_out, d_clk, n_cu);
  run_kind<DSW128_AND>(d_src, d_out, d_clk, n_cu);
  run_kind<DSR128_AND>(d_src, d_out, d_clk, n_cu);
  run_kind<DEP_AND_SUB>(d_src, d_out, d_clk, n_cu);
  run_kind<BUFLD_AND>(d_src, d_out, d_clk, n_cu);
  run_kind<SALU>(d_src, d_out, d_clk, n_cu);

  // exactness of the round-to-nearest split
  const int n = 1 << 20;
  std::vector<float> v(n);
  std::uniform_int_distribution<uint32_t> U(0, 0xffffffffu);
  for (int i = 0; i < n; ++i) {
    uint32_t b = U(rng);
    if (i % 4 == 0) { float x = G(rng) * std::pow(10.f, (float)((i / 4) % 60 - 30)); memcpy(&b, &x, 4); }
    if (((b >> 23) & 0xff) == 0xff) b &= 0x7f7fffffu | 0x80000000u;   // no inf / NaN
    memcpy(&v[i], &b, 4);
  }
  v[0] = 1e-30f; v[1] = 1e30f; v[2] = 1e-38f; v[3] = 3e-39f; v[4] = 1.17549435e-38f; v[5] = 3.3e38f; v[6] = -1e-36f; v[7] = 0.f;
  float* d_v; unsigned* d_p;
  hipMalloc(&d_v, n * 4); hipMalloc(&d_p, (size_t)n / 2 * 3 * 4);
  hipMemcpy(d_v, v.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_rne_kernel, dim3(n / 2 / 256), dim3(256), 0, 0, d_v, d_p, n);
  std::vector<unsigned> pc((size_t)n / 2 * 3);
  hipMemcpy(pc.data(), d_p, pc.size() * 4, hipMemcpyDeviceToHost);
  auto bf = [](unsigned short h16) { uint32_t b = (uint32_t)h16 << 16; float f; memcpy(&f, &b, 4); return (double)f; };
  long bad = 0, bad_normal = 0, bad_big = 0;
  int shown = 0;
  for (int i = 0; i < n / 2; ++i)
    for (int e = 0; e < 2; ++e) {
      const float x = v[2 * i + e];
      const double s = bf((unsigned short)(pc[3 * i] >> (16 * e))) + bf((unsigned short)(pc[3 * i + 1] >> (16 * e))) + bf((unsigned short)(pc[3 * i + 2] >> (16 * e)));
      if (s != (double)x) {
        ++bad;
        const float ax = std::fabs(x);
        if (ax >= 1e-30f && ax < 1e38f) ++bad_normal;
        if (ax >= 1e38f) ++bad_big;
        if (shown < 8) { printf("  not exact: v = %.9g (0x%08x), pieces %04x %04x %04x sum %.9g\n", x, *(const unsigned*)&x, (pc[3 * i] >> (16 * e)) & 0xffff, (pc[3 * i + 1] >> (16 * e)) & 0xffff, (pc[3 * i + 2] >> (16 * e)) & 0xffff, s); ++shown; }
      }
    }
  printf("SPLIT round-to-nearest (cvt_pk_bf16 + dot2_bf16): %d values, %ld not exact (%ld of them with 1e-30 <= |v| < 1e38, %ld with |v| >= 1e38)\n", n, bad, bad_normal, bad_big);
  return 0;
}
