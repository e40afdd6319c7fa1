// engine.hip -- library-wide state of libmp_engine.so: error string, version, device probe.
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
