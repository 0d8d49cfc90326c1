// Shared helpers of libipdm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/ipdm_hip.h"

namespace ipdm {

void set_error(const char *fmt, ...);

#define IPDM_HIP_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            ipdm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                            __LINE__);                                                         \
            return IPDM_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

#define IPDM_REQUIRE(cond, ...)                   \
    do {                                          \
        if (!(cond)) {                            \
            ipdm::set_error(__VA_ARGS__);         \
            return IPDM_ERR_INVALID;              \
        }                                         \
    } while (0)

#define IPDM_LAUNCH_CHECK() IPDM_HIP_CHECK(hipGetLastError())

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// hipFuncAttributeMaxDynamicSharedMemorySize for kernels that use more than 64 KiB of LDS: set once per
// (kernel, device), under a lock -- handles may be driven from different threads and devices of one process.
int ensure_dynamic_lds(const void *kernel, size_t bytes);
// multiprocessor count of the CURRENT device (cached per device)
int device_cu_count();

// wave64 reductions (CDNA wavefront = 64 lanes)
__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ inline float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}

}  // namespace ipdm
