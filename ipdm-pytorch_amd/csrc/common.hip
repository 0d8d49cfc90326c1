#include <mutex>
#include <set>
#include <utility>
#include "common.h"

namespace ipdm {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
std::mutex g_attr_mu;
std::set<std::pair<const void *, int>> g_attr_done;    // (kernel, device) pairs already configured
int g_cus[64];                                         // per device ordinal, 0 = not read yet
}  // namespace

int ensure_dynamic_lds(const void *kernel, size_t bytes)
{
    int dev = 0;
    IPDM_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if (g_attr_done.count({kernel, dev})) return IPDM_OK;
    IPDM_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    g_attr_done.insert({kernel, dev});
    return IPDM_OK;
}

int device_cu_count()
{
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if (!g_cus[dev])
        g_cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    return g_cus[dev];
}
}  // namespace ipdm

extern "C" const char *ipdm_last_error(void) { return ipdm::g_err; }
extern "C" int ipdm_abi_version(void) { return 1; }
