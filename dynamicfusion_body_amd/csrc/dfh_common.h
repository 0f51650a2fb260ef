// Shared host/device helpers for libdfusion_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/dfusion_hip.h"

namespace dfh {

// ---- error plumbing: nothing throws across the C ABI --------------------------------
char *last_error_buf();               // thread-local, 512 bytes (dfh_core.hip)
int fail(int code, const char *fmt, ...);

#define DFH_HIP_CHECK(expr)                                                               \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return ::dfh::fail(DFH_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define DFH_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return ::dfh::fail(DFH_E_BADARG, __VA_ARGS__); \
    } while (0)

// ---- small fixed-size parameter blocks passed by value to kernels ---------------------
struct Mat3 { double m[9]; };
struct Mat34 { double m[12]; };
struct Vec3 { double v[3]; };
struct DQ { double q[8]; };

constexpr int kWave = 64;             // CDNA wavefront

}  // namespace dfh
