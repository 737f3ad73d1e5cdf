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
