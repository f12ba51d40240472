// Library-level entry points: error text, ABI version, device check.
#include <stdarg.h>
#include <string.h>
#include "common.h"

namespace mslam {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mslam

extern "C" const char* mslam_last_error(void) { return mslam::g_err; }

extern "C" int mslam_abi_version(void) { return 1; }

extern "C" int mslam_device_check(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    mslam::set_error("no HIP device visible");
    return MSLAM_ENODEV;
  }
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    mslam::set_error("hipGetDeviceProperties failed");
    return MSLAM_ENODEV;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    mslam::set_error("device %d is %s; libmslam_hip.so is built for gfx950 only", dev, prop.gcnArchName);
    return MSLAM_ENODEV;
  }
  return MSLAM_OK;
}
