// Per-launch HIP-event timing of the hot kernels, on the stream they are launched on (bench.py's
// roofline leg: achieved = algorithmic FLOPs of a launch / its measured duration).  Off by default:
// zero cost on the product path.
#include <vector>
#include "unet_kernels.h"

namespace ipdm {
namespace {
struct Rec { hipEvent_t a, b; int cls; double flops; };
struct Prof {
    bool on = false;
    std::vector<Rec> pool;
    size_t used = 0;
    unsigned mask = ~0u;          // classes being recorded (ipdm_profile_begin_classes)
    bool skip = false;            // the launch between prof_before / prof_after belongs to a class that is not
} g_prof;
}  // namespace

bool prof_enabled() { return g_prof.on && g_prof.used < g_prof.pool.size(); }
void prof_before(int cls, hipStream_t st)
{
    g_prof.skip = !(g_prof.mask >> cls & 1u);
    if (g_prof.skip) return;
    Rec &r = g_prof.pool[g_prof.used];
    r.cls = cls;
    (void)hipEventRecord(r.a, st);
}
void prof_after(int cls, double flops, hipStream_t st)
{
    if (g_prof.skip) { g_prof.skip = false; return; }
    Rec &r = g_prof.pool[g_prof.used++];
    r.flops = flops;
    (void)hipEventRecord(r.b, st);
}
}  // namespace ipdm

using namespace ipdm;

// Starts recording up to max_launches kernel launches (events are created once and reused) of the classes in class_mask
// (bit c = class c): an event pair costs the stream about a microsecond, so a timed region records only the classes it reports.
extern "C" int ipdm_profile_begin(int32_t max_launches) { return ipdm_profile_begin_classes(max_launches, ~0u); }
extern "C" int ipdm_profile_begin_classes(int32_t max_launches, uint32_t class_mask)
{
    IPDM_REQUIRE(max_launches > 0, "profile_begin: bad capacity");
    g_prof.mask = class_mask;
    g_prof.skip = false;
    while ((int)g_prof.pool.size() < max_launches) {
        Rec r;
        IPDM_HIP_CHECK(hipEventCreate(&r.a));
        IPDM_HIP_CHECK(hipEventCreate(&r.b));
        r.cls = 0; r.flops = 0;
        g_prof.pool.push_back(r);
    }
    g_prof.used = 0;
    g_prof.on = true;
    return IPDM_OK;
}

// Stops recording; the caller must have synchronised the stream.  out_* are arrays of n_classes entries (the caller
// states its array length; the library has PROF_CLASSES classes and fails on a shorter array instead of overflowing it):
// total algorithmic FLOPs, total kernel milliseconds, launch count per class.
extern "C" int ipdm_profile_end(double *out_flops, double *out_ms, int64_t *out_launches, int32_t n_classes)
{
    IPDM_REQUIRE(out_flops && out_ms && out_launches, "profile_end: null argument");
    IPDM_REQUIRE(n_classes >= PROF_CLASSES, "profile_end: arrays of %d entries, the library records %d classes", n_classes, PROF_CLASSES);
    g_prof.on = false;
    for (int c = 0; c < n_classes; ++c) { out_flops[c] = 0; out_ms[c] = 0; out_launches[c] = 0; }
    for (size_t i = 0; i < g_prof.used; ++i) {
        const Rec &r = g_prof.pool[i];
        float ms = 0.f;
        IPDM_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        out_flops[r.cls] += r.flops;
        out_ms[r.cls] += ms;
        out_launches[r.cls] += 1;
    }
    g_prof.used = 0;
    return IPDM_OK;
}

// ---------------------------------------------------------------------------------------------- clock probe (diagnostic)
// One wave that does nothing but read the two clocks: s_memtime (shader clock) and s_memrealtime (100 MHz reference) every
// `period_us`, `samples` times.  Launched on a stream of its own it co-resides with whatever kernel fills the chip (one wave
// slot, no LDS), so the quotient of the differences is the clock the chip HOLDS under that kernel -- with no profiler
// attached and no stamp in the kernel being measured (MI355X_MICROARCH.md, DVFS give-back item 6).
namespace {
__global__ void clock_probe_kernel(unsigned long long *out, int samples, unsigned long long period_ticks)
{
    if (threadIdx.x != 0) return;
    unsigned long long r_next = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < samples; ++i) {
        unsigned long long r;
        do { __builtin_amdgcn_s_sleep(32); r = __builtin_amdgcn_s_memrealtime(); } while (r < r_next);
        out[2 * i] = __builtin_amdgcn_s_memtime();
        out[2 * i + 1] = r;
        r_next = r + period_ticks;
    }
}
}  // namespace

extern "C" int ipdm_clock_probe(uint64_t *d_out, int32_t samples, int32_t period_us, void *stream)
{
    IPDM_REQUIRE(d_out && samples > 1 && period_us > 0, "clock_probe: bad argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long *)d_out, samples,
                       (unsigned long long)period_us * 100ull);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}
