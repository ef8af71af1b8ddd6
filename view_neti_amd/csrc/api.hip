// Library-global pieces of libvneti_hip.so: version + thread-local error string.
#include "common.h"
#include "../../include/vneti.h"
#include <stdarg.h>

static thread_local char g_err[512] = {0};

void vneti_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int vneti_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vneti_set_error("%s: HIP launch error: %s", what, hipGetErrorString(e));
    return VNETI_EHIP;
  }
  return VNETI_OK;
}

extern "C" int vneti_version(void) { return VNETI_ABI_VERSION; }
extern "C" int vneti_precision(void) { return VN_PRECISION; }

extern "C" int vneti_last_error(char* buf, size_t n) {
  size_t len = strlen(g_err);
  if (buf && n > 0) {
    size_t c = len < n - 1 ? len : n - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return (int)len;
}
