// Shared helpers for libskoots_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/skoots_hip.h"

namespace sk {

void set_error(const char* fmt, ...);

#define SK_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            ::sk::set_error(__VA_ARGS__);  \
            return SK_ERR_ARG;             \
        }                                  \
    } while (0)

#define SK_CHECK_HIP(expr)                                                            \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            ::sk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                            __FILE__, __LINE__);                                      \
            return SK_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

#define SK_CHECK_LAUNCH()                                                             \
    do {                                                                              \
        hipError_t _e = hipGetLastError();                                            \
        if (_e != hipSuccess) {                                                       \
            ::sk::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),\
                            __FILE__, __LINE__);                                      \
            return SK_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// Memory-bound grids: cap and grid-stride (256 CUs x 8 blocks).
static inline unsigned stream_grid(int64_t n, int block, int per_thread = 1) {
    int64_t g = (n + (int64_t)block * per_thread - 1) / ((int64_t)block * per_thread);
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (unsigned)g;
}

}  // namespace sk
