// Shared helpers of libipdm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
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

// ---- process-wide switches (ipdm_set_option / ipdm_get_option of the ABI).  One table instead of getenv() calls scattered
// over the launch paths: every entry is read from the environment ONCE (IPDM_<NAME>, kept as a debug alias), after that
// only ipdm_set_option changes it.  A UNet handle records the table at ipdm_unet_create; a forward refuses to run when
// an entry that shapes packed weights, workspace layout or kernel choice differs from that record (OPT_PER_CALL entries
// excepted), so a switch flipped between create and forward is an error, not a silently mismatched layout.
enum Opt {
    OPT_CONV_NO_UP2,         // Upsample layers in the reference's 3x3 form (per call: both weight sets are packed)
    OPT_CONV_NO_WUP2,        // wide Upsample layers on the 2x2-tap parity kernel instead of its F(2x2,2x2) form (per call; conv_wup2.hip)
    OPT_CONV_LEGACY, OPT_CONV1X1_LEGACY, OPT_CONVS2_LEGACY,      // route kernel families to the round-1 4-wave kernels
    OPT_CONV_NO_DIRECT, OPT_DIRECT_NO_PLANAR, OPT_DIRECT_MAX_CIN, OPT_DIRECT_NO_S2, OPT_DIRECT_NO_SKIP_FUSE,
    OPT_CONV_DBG, OPT_CONV_VEC4_STRICT, OPT_CONV_NO_SPLITK, OPT_CONV_NO_WINO, OPT_WINO_V1, OPT_WINO2_MIN_TILES, OPT_CONV1X1_NO_QUARTER, OPT_CONV_NO_PW, OPT_PW_ITEM, OPT_PW_FORCE,
    OPT_CONV_NM,             // opt-in: narrow stride-1 layers on the 16-cout MFMA (conv_nm.hip): 1 = 16-cout layers, 2 = 8-cout too
    OPT_GN_TWO_STAGE, OPT_GN_UNFUSED,
    OPT_UNET_TRANSPOSE,      // -1 automatic | 0 never | 1 always
    OPT_ATTN_NO_KVSPLIT, OPT_ATTN_LEGACY, OPT_ATTN_NO_ZSEQ,
    OPT_ART_PER_VIEW,
    OPT_CONV_BF16X3,         // opt-in (round 6): conv_wino2's layers with the channel contraction on the bf16 matrix pipe, error-free 3-way split (conv_wino3.hip; per call)
    OPT_COUNT
};
int opt(Opt o);
bool opt_per_call(int o);                 // entries a live handle tolerates being changed
const char *opt_name(int o);
void opt_snapshot(int (&dst)[OPT_COUNT]);
// first entry (not per-call) whose current value differs from `rec`, or -1
int opt_changed_since(const int (&rec)[OPT_COUNT]);
// the current values of the per-call entries, in table order (graph keys)
std::vector<int> opt_per_call_values();

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
