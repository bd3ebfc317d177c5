// Shared host-side helpers for libnsgp_repre_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/nsgp_repre.h"

namespace nsgp {

// thread-local last-error text surfaced through nsgp_last_error()
inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define NSGP_HIP(call)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return ::nsgp::fail(NSGP_ERR_HIP, "%s failed: %s (%s:%d)", #call,              \
                                hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)

#define NSGP_LAUNCH_CHECK()                                                                \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess)                                                              \
            return ::nsgp::fail(NSGP_ERR_HIP, "kernel launch failed: %s (%s:%d)",          \
                                hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace nsgp
