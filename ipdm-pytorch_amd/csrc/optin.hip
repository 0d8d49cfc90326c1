// The opt-in kernel -- the 16-cout MFMA form of the narrow layers (conv_nm.hip: option conv_nm) -- lives in a SECOND shared
// object, libipdm_hip_optin.so, so that the product library carries only what the default path dispatches.  This file is the
// product side: the functions the dispatchers call, which load the second library (from the directory this one was loaded
// from) the first time the option actually asks for it, and fail with IPDM_ERR_UNSUPPORTED when it is not there.
//
// The second library resolves the product's internals (options table, profiling hooks, error text, statistics geometry)
// against `libipdm_hip.so` by name.  When THIS code was loaded from a file of another name (a variant build,
// IPDM_LIB_PATH=libipdm_hip_<tag>.so) that would map a second copy of the product with its own option values and error
// buffer: ipdm_optin_bound_to() reports which copy the second library bound to, and a mismatch refuses the load.
#include <dlfcn.h>
#include <mutex>
#include <string>
#include "common.h"
#include "unet_kernels.h"

namespace ipdm {
namespace {
std::mutex g_mu;
void *g_handle = nullptr;
bool g_tried = false;
std::string g_why;                      // why the load failed (repeated into the error text on every call)

void *optin_sym(const char *name)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_tried) {
        g_tried = true;
        Dl_info info;
        std::string path = "libipdm_hip_optin.so";
        if (dladdr((const void *)&optin_sym, &info) && info.dli_fname) {
            const std::string self = info.dli_fname;
            const size_t slash = self.rfind('/');
            if (slash != std::string::npos) path = self.substr(0, slash + 1) + path;
        }
        g_handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!g_handle) {
            const char *e = dlerror();
            g_why = path + " cannot be loaded: " + (e ? e : "?");
        } else {
            using Fn = const void *(*)();
            Fn bound = (Fn)dlsym(g_handle, "ipdm_optin_bound_to");
            if (!bound || bound() != (const void *)&ipdm_last_error) {
                g_why = path + " is bound to another copy of the product library than the one in use (load the product as "
                               "libipdm_hip.so to use the opt-in kernels)";
                dlclose(g_handle);
                g_handle = nullptr;
            } else {
                using Lay = void (*)(int *);
                Lay lay = (Lay)dlsym(g_handle, "ipdm_optin_layout");
                int got[3] = {-1, -1, -1};
                if (lay) lay(got);
                if (got[0] != IPDM_ABI_VERSION || got[1] != (int)sizeof(ConvArgs) || got[2] != (int)OPT_COUNT) {
                    g_why = path + " was built against another layout (ABI " + std::to_string(got[0]) + ", ConvArgs " + std::to_string(got[1]) +
                            " bytes, " + std::to_string(got[2]) + " options; this library: ABI " + std::to_string(IPDM_ABI_VERSION) + ", " +
                            std::to_string(sizeof(ConvArgs)) + " bytes, " + std::to_string((int)OPT_COUNT) + "): rebuild both (make -C csrc)";
                    dlclose(g_handle);
                    g_handle = nullptr;
                }
            }
        }
    }
    if (!g_handle) { set_error("opt-in kernels requested but %s", g_why.c_str()); return nullptr; }
    void *f = dlsym(g_handle, name);
    if (!f) set_error("libipdm_hip_optin.so has no symbol %s", name);
    return f;
}
}  // namespace

bool conv_nm_eligible(const ConvArgs &a)
{
    if (opt(OPT_CONV_NM) <= 0) return false;              // (the default path never touches the second library)
    using Fn = int (*)(const ConvArgs *);
    Fn f = (Fn)optin_sym("ipdm_optin_conv_nm_eligible");
    return f && f(&a) != 0;
}

int conv2d_nm_launch(const ConvArgs &a, hipStream_t st)
{
    using Fn = int (*)(const ConvArgs *, hipStream_t);
    Fn f = (Fn)optin_sym("ipdm_optin_conv2d_nm_launch");
    return f ? f(&a, st) : IPDM_ERR_UNSUPPORTED;
}

}  // namespace ipdm
