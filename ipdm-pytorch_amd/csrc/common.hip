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
}  // namespace ipdm

extern "C" const char *ipdm_last_error(void) { return ipdm::g_err; }
extern "C" int ipdm_abi_version(void) { return 1; }
