// Error reporting and ABI version of libyolo_hip.so.
#include "common.h"

namespace yolo {

char *err_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace yolo

YOLO_API int yolo_hip_abi_version(void) { return YOLO_HIP_ABI_VERSION; }
YOLO_API const char *yolo_hip_last_error(void) { return yolo::err_buf(); }

// A HIP stream of the lowest (low != 0) or the default scheduling priority on the current device.  The engine runs work that is off
// the critical path (weight gradients, the deferred optimizer pass) on a low-priority stream so that the dispatcher prefers
// the workgroups of the main chain whenever both have some ready.
YOLO_API int yolo_stream_create(int low, yolo_stream_t *out)
{
    if (!out) return yolo::fail(YOLO_E_ARG, "yolo_stream_create: out is NULL");
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStream_t s = nullptr;
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, low ? least : 0);
    if (e != hipSuccess) return yolo::fail((int)e, "yolo_stream_create: %s", hipGetErrorString(e));
    *out = (yolo_stream_t)s;
    return 0;
}
