// Shared helpers for the HIP translation units of libmp_engine.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "mp_engine.h"
#include "mp_engine_debug.h"

namespace mp {

void set_error(const char* fmt, ...);

#define MP_CHECK_HIP(expr)                                                               \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::mp::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return MP_ERR_HIP;                                                                 \
    }                                                                                    \
  } while (0)

#define MP_REQUIRE(cond, ...)          \
  do {                                 \
    if (!(cond)) {                     \
      ::mp::set_error(__VA_ARGS__);    \
      return MP_ERR_INVALID;           \
    }                                  \
  } while (0)

// Optional per-launch HIP-event profiler (mp_profile_begin/end/query): events are recorded on the launch stream around
// each instrumented kernel; aggregation happens at query time.  Disabled = zero overhead.
struct ProfScope {
  // flops = ALGORITHMIC work (2 * MACs of the convolution); executed = the FLOPs the kernel really issues on the matrix
  // pipe whose dense peak is peak_tflops (Winograd executes fewer fp32 FLOPs, the exact-piece kernels more bf16 ones); < 0 = flops
  ProfScope(const char* name, double flops, double bytes, hipStream_t s, double executed = -1.0, double peak_tflops = 157.3);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware bijective remap of a linear workgroup id (8 XCDs, round-robin dispatch):
// physical id p runs on XCD p % 8; give each XCD a contiguous chunk of logical ids so that
// neighbouring tiles (shared halos / shared weight panels) hit the same private L2.
__device__ __forceinline__ int xcd_remap(int p, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = p & 7, idx = p >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace mp
