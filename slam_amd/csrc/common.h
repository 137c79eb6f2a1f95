// Shared host-side helpers of libslam_amd.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <new>

#include "../../include/slam_amd.h"

void slam_set_error(const char *fmt, ...);

#define SLAM_HIP_CHECK(expr)                                                                          \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            slam_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SLAM_E_HIP;                                                                        \
        }                                                                                             \
    } while (0)

#define SLAM_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            slam_set_error(__VA_ARGS__);   \
            return SLAM_E_INVALID;         \
        }                                  \
    } while (0)

// No C++ exception crosses the C ABI: entry points that allocate host memory or use the worker pool are written as
// function-try-blocks closed by this macro (`extern "C" int f(...) try { ... } SLAM_CATCH_ALL("f")`).  `cleanup` runs
// first (e.g. draining streams so that no device work of the failed call outlives it).
#define SLAM_CATCH_ALL_WITH(name, cleanup)                            \
    catch (const std::bad_alloc &) {                                  \
        cleanup;                                                      \
        slam_set_error(name ": out of host memory");                  \
        return SLAM_E_HOST;                                           \
    }                                                                 \
    catch (const std::exception &e_) {                                \
        cleanup;                                                      \
        slam_set_error(name ": host failure: %s", e_.what());         \
        return SLAM_E_HOST;                                           \
    }                                                                 \
    catch (...) {                                                     \
        cleanup;                                                      \
        slam_set_error(name ": host failure");                        \
        return SLAM_E_HOST;                                           \
    }
#define SLAM_CATCH_ALL(name) SLAM_CATCH_ALL_WITH(name, (void)0)

static inline int slam_div_up(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t slam_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// 3x4 / 3x3 matrices handed to kernels by value
struct Mat34 { double m[12]; };
struct Mat33 { double m[9]; };

// A wave-uniform double moved to scalar registers (the FP64 VALU reads one scalar operand per instruction).  Values
// that are the same in every lane -- a hypothesis' projection matrix, a reduced sum -- then cost no vector registers.
__device__ __forceinline__ double to_scalar(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

