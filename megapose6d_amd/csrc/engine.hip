// engine.hip -- library-wide state of libmp_engine.so: error string, version, device probe.
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace mp {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mp

namespace mp {
struct ProfRec {
  const char* name;
  double flops, bytes, executed, peak;
  hipEvent_t e0, e1;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;   // events are recycled: creating two per launch inside a timed region costs host time
static hipEvent_t pool_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}

ProfScope::ProfScope(const char* name, double flops, double bytes, hipStream_t s, double executed, double peak_tflops) : slot(-1), stream(s) {
  if (!g_prof_on) return;
  ProfRec r;
  r.name = name; r.flops = flops; r.bytes = bytes; r.executed = executed < 0.0 ? flops : executed; r.peak = peak_tflops;
  r.e0 = pool_event();
  r.e1 = pool_event();
  if (!r.e0 || !r.e1) return;
  (void)hipEventRecord(r.e0, s);
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof[slot].e1, stream);
}
}  // namespace mp

extern "C" int mp_profile_begin(void) {
  for (auto& r : mp::g_prof) { mp::g_event_pool.push_back(r.e0); mp::g_event_pool.push_back(r.e1); }
  mp::g_prof.clear();
  mp::g_prof_on = true;
  return MP_OK;
}
extern "C" int mp_profile_end(void) {
  mp::g_prof_on = false;
  return MP_OK;
}
extern "C" int mp_profile_active(void) { return mp::g_prof_on ? 1 : 0; }
// Aggregated by kernel name.  idx enumerates distinct names; returns 1 when idx is past the end.
extern "C" int mp_profile_query(int idx, char* name, int name_len, int64_t* launches, double* total_ms, double* total_flops,
                                double* total_bytes) {
  return mp_profile_query_ex(idx, name, name_len, launches, total_ms, total_flops, total_bytes, nullptr, nullptr);
}

extern "C" int mp_profile_query_ex(int idx, char* name, int name_len, int64_t* launches, double* total_ms, double* total_flops,
                                   double* total_bytes, double* total_executed_flops, double* peak_tflops) {
  std::map<std::string, int> order;
  std::vector<std::string> names;
  for (auto& r : mp::g_prof)
    if (!order.count(r.name)) { order[r.name] = (int)names.size(); names.push_back(r.name); }
  if (idx < 0 || idx >= (int)names.size()) return 1;
  int64_t n = 0;
  double ms = 0, fl = 0, by = 0, ex = 0, pk = 0;
  for (auto& r : mp::g_prof) {
    if (names[idx] != r.name) continue;
    MP_CHECK_HIP(hipEventSynchronize(r.e1));
    float t = 0.f;
    MP_CHECK_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
    ms += t; fl += r.flops; by += r.bytes; ex += r.executed; pk = r.peak; ++n;
  }
  if (name && name_len > 0) { strncpy(name, names[idx].c_str(), name_len - 1); name[name_len - 1] = 0; }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (total_bytes) *total_bytes = by;
  if (total_executed_flops) *total_executed_flops = ex;
  if (peak_tflops) *peak_tflops = pk;
  return MP_OK;
}

namespace mp {
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void clock_probe_kernel(float* sink, unsigned long long* clk, int iters) {
  probe_f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int k = 0; k < 8; ++k) {   // operands that differ per lane and per instruction (bounded: the sums stay finite)
    a[k] = (float)((threadIdx.x * 7 + k * 13) % 31 - 15) * 0.03125f;
    b[k] = (float)((threadIdx.x * 5 + k * 11) % 29 - 14) * 0.03125f;
  }
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 7], b[(u + 3 * i) & 7], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sink[0] = s;   // keeps the loop alive
  if (threadIdx.x == 0) {
    atomicAdd(&clk[0], c1 - c0);
    atomicAdd(&clk[1], r1 - r0);
  }
}
}  // namespace mp

extern "C" int mp_clock_probe(double ms_target, double* shader_mhz, double* mfma_tflops, mp_stream stream) {
  MP_REQUIRE(ms_target > 0.0 && ms_target <= 2000.0, "mp_clock_probe: ms_target out of range");
  hipStream_t s = (hipStream_t)stream;
  int dev = 0, n_cu = 0;
  MP_CHECK_HIP(hipGetDevice(&dev));
  MP_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  float* sink = nullptr;
  unsigned long long* clk = nullptr;
  MP_CHECK_HIP(hipMalloc(&sink, sizeof(float)));
  MP_CHECK_HIP(hipMalloc(&clk, 2 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  MP_CHECK_HIP(hipEventCreate(&e0));
  MP_CHECK_HIP(hipEventCreate(&e1));
  // 32 MFMAs of 64 cycles per iteration and wave, two waves per SIMD: 4096 cycles per iteration at full rate
  const int iters = (int)(ms_target * 1e-3 * 2.4e9 / 4096.0) + 1;
  const int grid = 2 * n_cu;
  int rc = MP_OK;
  float ms = 0.f;
  unsigned long long h[2] = {0, 0};
  for (int rep = 0; rep < 2 && rc == MP_OK; ++rep) {   // the first launch warms the clock up
    if (hipMemsetAsync(clk, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) { rc = MP_ERR_HIP; break; }
    (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(mp::clock_probe_kernel, dim3(grid), dim3(256), 0, s, sink, clk, iters);
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = MP_ERR_HIP;
  }
  if (rc == MP_OK && hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) rc = MP_ERR_HIP;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  (void)hipFree(clk);
  if (rc != MP_OK) { mp::set_error("mp_clock_probe: HIP error"); return rc; }
  if (shader_mhz) *shader_mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
  if (mfma_tflops) *mfma_tflops = ms > 0.f ? (double)grid * 4.0 * iters * 32.0 * 4096.0 / (ms * 1e-3) * 1e-12 : 0.0;
  return MP_OK;
}

extern "C" int mp_version(void) { return 100; }
extern "C" const char* mp_last_error(void) { return mp::g_err; }

extern "C" int mp_device_info(int* n_cus, int* lds_bytes, char* arch_name, int arch_name_len) {
  int dev = 0;
  MP_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  MP_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  if (n_cus) *n_cus = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  MP_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "libmp_engine.so is built for gfx950 only, device is %s", prop.gcnArchName);
  return MP_OK;
}
