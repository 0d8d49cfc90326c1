// The opt-in kernels -- split-bf16 convolution / attention (conv_sx.hip, attn_sx.hip: options conv_split, attn_split) and the
// 16-cout MFMA form of the narrow layers (conv_nm.hip: option conv_nm) -- live in a SECOND shared object,
// libipdm_hip_optin.so, so that the product library carries only what the default path dispatches.  This file is the
// product side: the functions the dispatchers call, which load the second library (from the directory this one was loaded
// from) the first time an opt-in mode actually asks for it, and fail with IPDM_ERR_UNSUPPORTED when it is not there.
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
        if (!g_handle) set_error("opt-in kernels requested but %s cannot be loaded: %s", path.c_str(), dlerror());
    }
    if (!g_handle) return nullptr;
    void *f = dlsym(g_handle, name);
    if (!f) set_error("libipdm_hip_optin.so has no symbol %s", name);
    return f;
}
}  // namespace

// interleave code 100 + pieces marks weights packed for the split-bf16 kernel
int conv_sx_pieces(int interleave) { return interleave >= 100 ? interleave - 100 : 0; }

void conv_sx_pack_weights(const float *w, int Cout, int Cin, int ns, std::vector<float> &packed, int &cin_pad, int &cout_pad)
{
    using Fn = void (*)(const float *, int, int, int, std::vector<float> *, int *, int *);
    if (Fn f = (Fn)optin_sym("ipdm_optin_conv_sx_pack_weights")) f(w, Cout, Cin, ns, &packed, &cin_pad, &cout_pad);
    else { packed.clear(); cin_pad = cout_pad = 0; }      // (the launch that follows reports the missing library)
}

int conv2d_sx_launch(const ConvArgs &a, hipStream_t st)
{
    using Fn = int (*)(const ConvArgs *, hipStream_t);
    Fn f = (Fn)optin_sym("ipdm_optin_conv2d_sx_launch");
    return f ? f(&a, st) : IPDM_ERR_UNSUPPORTED;
}

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

size_t attention_sx_scratch_floats(int B, int heads, int T)
{
    if (opt(OPT_ATTN_SPLIT) != 3) return 0;               // (attn_split is recorded at ipdm_unet_create: a handle never changes mode)
    using Fn = size_t (*)(int, int, int);
    Fn f = (Fn)optin_sym("ipdm_optin_attention_sx_scratch_floats");
    return f ? f(B, heads, T) : 0;
}

int attention_sx_launch(const float *qkv, float *scratch, float *out, int B, int heads, int T, float scale, hipStream_t st)
{
    using Fn = int (*)(const float *, float *, float *, int, int, int, float, hipStream_t);
    Fn f = (Fn)optin_sym("ipdm_optin_attention_sx_launch");
    return f ? f(qkv, scratch, out, B, heads, T, scale, st) : IPDM_ERR_UNSUPPORTED;
}

}  // namespace ipdm
