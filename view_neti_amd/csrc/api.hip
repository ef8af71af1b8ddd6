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

// ---- CU-partitioned streams ("the pipelined VAE encode"): a side stream whose kernels may only
// occupy the compute units of `mask` (bit i = logical CU i; on gfx950 the driver deals the bits round-robin over the 8
// XCDs, so the first 8*k bits are k CUs on every XCD).  A LINEAR hipGraph launched on such a stream runs on that
// stream's queue and inherits the mask.
extern "C" int vneti_stream_create_cu_mask(const unsigned* mask, int nwords, void** stream) {
  if (!mask || nwords <= 0 || !stream) {
    vneti_set_error("vneti_stream_create_cu_mask: bad arguments");
    return VNETI_EARG;
  }
  hipStream_t s = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask);
  if (e != hipSuccess) {
    vneti_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    return VNETI_EHIP;
  }
  *stream = (void*)s;
  return VNETI_OK;
}

extern "C" int vneti_stream_get_cu_mask(void* stream, unsigned* mask, int nwords) {
  hipError_t e = hipExtStreamGetCUMask((hipStream_t)stream, (uint32_t)nwords, mask);
  if (e != hipSuccess) {
    vneti_set_error("hipExtStreamGetCUMask: %s", hipGetErrorString(e));
    return VNETI_EHIP;
  }
  return VNETI_OK;
}

extern "C" int vneti_stream_destroy(void* stream) {
  if (!stream) return VNETI_OK;
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) {
    vneti_set_error("hipStreamDestroy: %s", hipGetErrorString(e));
    return VNETI_EHIP;
  }
  return VNETI_OK;
}
