// Shared host-side helpers for libyolo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/yolo_hip.h"

#define YOLO_API extern "C" __attribute__((visibility("default")))

namespace yolo {

// per-thread last error text (returned by yolo_hip_last_error)
char *err_buf();
int fail(int code, const char *fmt, ...);

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

inline hipStream_t STRM(yolo_stream_t s) { return (hipStream_t)s; }

typedef unsigned short bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even; a plain cast compiles to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f32_to_bf16(float f)
{
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

}  // namespace yolo
