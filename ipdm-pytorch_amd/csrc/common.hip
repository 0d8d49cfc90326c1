#include <cctype>
#include <cstdlib>
#include <mutex>
#include <strings.h>
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

// ---------------------------------------------------------------------------------------------- options
namespace {
struct OptDef { const char *name; int def; bool per_call; int lo = 0, hi = 1; };      // [lo, hi]: what ipdm_set_option accepts
const OptDef g_opt_def[OPT_COUNT] = {
    {"conv_no_up2", 0, true},
    {"conv_no_wup2", 0, true},
    {"conv_legacy", 0, false}, {"conv1x1_legacy", 0, false}, {"convs2_legacy", 0, false},
    {"conv_no_direct", 0, false}, {"direct_no_planar", 0, false},
    {"direct_max_cin", 160, false, 0, 160},      // (conv_direct's LDS staging and chunk loop are sized for at most 160 input channels)
    {"direct_no_s2", 0, false}, {"direct_no_skip_fuse", 0, true},
    {"conv_dbg", 0, true, 0, 0x3ff}, {"conv_vec4_strict", 0, false}, {"conv_no_splitk", 0, false},
    {"conv_no_wino", 0, true}, {"wino_v1", 0, true}, {"wino2_min_tiles", 192, true, 0, 1 << 24},
    {"conv1x1_no_quarter", 0, true}, {"conv_no_pw", 0, true}, {"pw_item", 0, true, 0, 2}, {"pw_force", 0, true}, {"conv_nm", 0, true, 0, 2},
    {"gn_two_stage", 0, true}, {"gn_unfused", 0, true},
    {"unet_transpose", -1, true, -1, 1},
    {"attn_no_kvsplit", 0, false}, {"attn_legacy", 0, false}, {"attn_no_zseq", 0, true},
    {"art_per_view", 0, true},
    {"conv_bf16x3", 0, true},
};
bool opt_value_ok(int i, int v)
{
    return v >= g_opt_def[i].lo && v <= g_opt_def[i].hi;
}
std::mutex g_opt_mu;
int g_opt_val[OPT_COUNT];
bool g_opt_init = false;

void opt_init_locked()
{
    if (g_opt_init) return;
    for (int i = 0; i < OPT_COUNT; ++i) {
        char env[64] = "IPDM_";
        size_t n = 5;
        for (const char *c = g_opt_def[i].name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)toupper((unsigned char)*c);
        env[n] = 0;
        const char *e = getenv(env);
        g_opt_val[i] = g_opt_def[i].def;
        if (e) {                       // presence switches a flag on; a number is taken as the value ("0" switches it off)
            char *end = nullptr;
            const long v = strtol(e, &end, 10);
            const int want = (end != e) ? (int)v : 1;
            if (opt_value_ok(i, want)) g_opt_val[i] = want;
            else fprintf(stderr, "libipdm_hip: %s=%s is outside the option's range [%d, %d]: ignored\n", env, e, g_opt_def[i].lo, g_opt_def[i].hi);
        }
    }
    g_opt_init = true;
}
}  // namespace

int opt(Opt o)
{
    std::lock_guard<std::mutex> lk(g_opt_mu);
    opt_init_locked();
    return g_opt_val[o];
}
bool opt_per_call(int o) { return g_opt_def[o].per_call; }
const char *opt_name(int o) { return g_opt_def[o].name; }
void opt_snapshot(int (&dst)[OPT_COUNT])
{
    std::lock_guard<std::mutex> lk(g_opt_mu);
    opt_init_locked();
    for (int i = 0; i < OPT_COUNT; ++i) dst[i] = g_opt_val[i];
}
std::vector<int> opt_per_call_values()
{
    std::lock_guard<std::mutex> lk(g_opt_mu);
    opt_init_locked();
    std::vector<int> v;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (g_opt_def[i].per_call) v.push_back(g_opt_val[i]);
    return v;
}
int opt_changed_since(const int (&rec)[OPT_COUNT])
{
    std::lock_guard<std::mutex> lk(g_opt_mu);
    opt_init_locked();
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!g_opt_def[i].per_call && g_opt_val[i] != rec[i]) return i;
    return -1;
}
static int opt_find(const char *name)
{
    if (!name) return -1;
    if (!strncmp(name, "IPDM_", 5) || !strncmp(name, "ipdm_", 5)) name += 5;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcasecmp(name, g_opt_def[i].name)) return i;
    return -1;
}

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
extern "C" int ipdm_abi_version(void) { return IPDM_ABI_VERSION; }      // 2: ipdm_profile_end takes its array length; ipdm_conv_kernel_code  3: ipdm_profile_begin_classes  4: profile class 6 (IPDM_PROF_CLASSES 7); the split-bf16 switches are gone  5: profile class 7 (conv_wup2; IPDM_PROF_CLASSES 8), kernel code 11, option conv_no_wup2

extern "C" int ipdm_set_option(const char *name, int value)
{
    const int i = ipdm::opt_find(name);
    IPDM_REQUIRE(i >= 0, "ipdm_set_option: unknown option '%s'", name ? name : "(null)");
    IPDM_REQUIRE(ipdm::opt_value_ok(i, value), "ipdm_set_option: %s = %d is outside [%d, %d]", ipdm::g_opt_def[i].name, value,
                 ipdm::g_opt_def[i].lo, ipdm::g_opt_def[i].hi);
    std::lock_guard<std::mutex> lk(ipdm::g_opt_mu);
    ipdm::opt_init_locked();
    ipdm::g_opt_val[i] = value;
    return IPDM_OK;
}

extern "C" int ipdm_get_option(const char *name, int *value)
{
    const int i = ipdm::opt_find(name);
    IPDM_REQUIRE(i >= 0 && value, "ipdm_get_option: unknown option '%s'", name ? name : "(null)");
    *value = ipdm::opt((ipdm::Opt)i);
    return IPDM_OK;
}
