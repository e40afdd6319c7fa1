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
  double flops, bytes;
  hipEvent_t e0, e1;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

ProfScope::ProfScope(const char* name, double flops, double bytes, hipStream_t s) : slot(-1), stream(s) {
  if (!g_prof_on) return;
  ProfRec r;
  r.name = name; r.flops = flops; r.bytes = bytes;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  (void)hipEventRecord(r.e0, s);
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof[slot].e1, stream);
}
}  // namespace mp

extern "C" int mp_profile_begin(void) {
  for (auto& r : mp::g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  mp::g_prof.clear();
  mp::g_prof_on = true;
  return MP_OK;
}
extern "C" int mp_profile_end(void) {
  mp::g_prof_on = false;
  return MP_OK;
}
// Aggregated by kernel name.  idx enumerates distinct names; returns 1 when idx is past the end.
extern "C" int mp_profile_query(int idx, char* name, int name_len, int64_t* launches, double* total_ms, double* total_flops,
                                double* total_bytes) {
  std::map<std::string, int> order;
  std::vector<std::string> names;
  for (auto& r : mp::g_prof)
    if (!order.count(r.name)) { order[r.name] = (int)names.size(); names.push_back(r.name); }
  if (idx < 0 || idx >= (int)names.size()) return 1;
  int64_t n = 0;
  double ms = 0, fl = 0, by = 0;
  for (auto& r : mp::g_prof) {
    if (names[idx] != r.name) continue;
    MP_CHECK_HIP(hipEventSynchronize(r.e1));
    float t = 0.f;
    MP_CHECK_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
    ms += t; fl += r.flops; by += r.bytes; ++n;
  }
  if (name && name_len > 0) { strncpy(name, names[idx].c_str(), name_len - 1); name[name_len - 1] = 0; }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (total_bytes) *total_bytes = by;
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
