// C-ABI plumbing shared by every entry point: status codes, thread-local error text,
// launch checking.  No exception ever crosses the boundary; the library owns no global
// device state (callers own every buffer and pass the stream explicitly).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void svlm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int svlm_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    svlm_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SVLM_ELAUNCH;
  }
  return SVLM_OK;
}

extern "C" const char* svlm_last_error(void) { return g_err; }
extern "C" int svlm_abi_version(void) { return 1; }

// Number of CUs of the current device (used by host code to size split-KV grids).
extern "C" int svlm_device_cus(void) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    svlm_set_error("svlm_device_cus: no HIP device");
    return SVLM_ELAUNCH;
  }
  return p.multiProcessorCount;
}
