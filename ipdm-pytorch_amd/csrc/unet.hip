// UNet epsilon-predictor executor for gfx950 (replaces UNetModel, Model/model.py:190-310).
//
// The module topology is rebuilt from the constructor arguments exactly as UNetModel.__init__ walks it
// (:224-281), parameters are addressed by the reference's state_dict key names (so reference
// checkpoints load unchanged) and repacked once into the kernels' layouts.  A forward is a fixed
// sequence of launches on the caller's stream over a caller-provided workspace:
//   time-embedding MLP + all per-block projections (2 tiny launches), then per block
//   GN statistics -> conv (GN+SiLU prologue, bias/residual epilogue) ..., flash attention.
// Activations are NCHW fp32; the workspace is carved by a first-fit arena whose schedule is replayed
// identically on every call (ipdm_unet_workspace_bytes runs the same walk without launching).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <tuple>
#include <vector>
#include "unet_kernels.h"

using namespace ipdm;

// ------------------------------------------------------------------------------------ small kernels
namespace {

__global__ void __launch_bounds__(256) temb_kernel(const float *__restrict__ freqs, int mc, float t,
                                                   const float *__restrict__ w0, const float *__restrict__ b0,
                                                   const float *__restrict__ w2, const float *__restrict__ b2,
                                                   float *__restrict__ tmp, float *__restrict__ silu_emb)
{
    // single workgroup: sinusoid[mc] -> Linear(mc,4mc) -> SiLU -> Linear(4mc,4mc) -> SiLU (the SiLU of every
    // ResidualBlock.time_emb, Model/model.py:105-108, is applied once here)
    extern __shared__ float sm[];
    float *e0 = sm;              // mc
    float *h1 = sm + mc;         // 4mc
    const int half = mc / 2, ted = 4 * mc;
    for (int i = threadIdx.x; i < mc; i += 256) {
        float v = 0.0f;
        if (i < half) v = cosf(t * freqs[i]);
        else if (i < 2 * half) v = sinf(t * freqs[i - half]);
        e0[i] = v;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < ted; r += 256) {
        float acc = 0.0f;
        for (int k = 0; k < mc; ++k) acc += w0[(size_t)r * mc + k] * e0[k];
        acc += b0[r];
        h1[r] = acc / (1.0f + expf(-acc));
    }
    __syncthreads();
    for (int r = threadIdx.x; r < ted; r += 256) {
        float acc = 0.0f;
        for (int k = 0; k < ted; ++k) acc += w2[(size_t)r * ted + k] * h1[k];
        acc += b2[r];
        tmp[r] = acc;
        silu_emb[r] = acc / (1.0f + expf(-acc));
    }
}

// one wave per output row
__global__ void __launch_bounds__(256) gemv_bias_kernel(const float *__restrict__ W, const float *__restrict__ b,
                                                        const float *__restrict__ base, const float *__restrict__ v,
                                                        float *__restrict__ y, int rows, int K)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float acc = 0.0f;
    for (int k = lane; k < K; k += 64) acc += W[(size_t)row * K + k] * v[k];
    acc = wave_sum(acc);
    if (lane == 0) y[row] = base[row] + (acc + b[row]);
}

__global__ void __launch_bounds__(256) concat_kernel(const float *__restrict__ x1, const float *__restrict__ x2,
                                                     float *__restrict__ out, int C1, int C2, long HW, long total)
{
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    const int Ct = C1 + C2;
    for (; i < total; i += stride) {
        const long hw = i % HW;
        const long nc = i / HW;
        const int c = (int)(nc % Ct);
        const long n = nc / Ct;
        out[i] = c < C1 ? x1[(n * C1 + c) * HW + hw] : x2[(n * C2 + (c - C1)) * HW + hw];
    }
}

// [N][H][W] -> [N][W][H] through 32x33 LDS tiles (both sides coalesced)
__global__ void __launch_bounds__(256) transpose_hw_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W)
{
    __shared__ float tile[32][33];
    const size_t plane = (size_t)H * W;
    const float *src = in + blockIdx.z * plane;
    float *dst = out + blockIdx.z * plane;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (y0 + r < H && x0 + tx < W) tile[r][tx] = src[(size_t)(y0 + r) * W + x0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (x0 + r < W && y0 + tx < H) dst[(size_t)(x0 + r) * H + y0 + tx] = tile[tx][r];
}

}  // namespace

namespace ipdm {
int temb_launch(const float *freqs, int mc, int t, const float *w0, const float *b0, const float *w2, const float *b2,
                float *tmp, float *silu_emb, hipStream_t st)
{
    hipLaunchKernelGGL(temb_kernel, dim3(1), dim3(256), (size_t)5 * mc * sizeof(float), st, freqs, mc, (float)t, w0, b0,
                       w2, b2, tmp, silu_emb);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}
int gemv_bias_launch(const float *W, const float *b, const float *base, const float *v, float *y, int rows, int K,
                     hipStream_t st)
{
    hipLaunchKernelGGL(gemv_bias_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, W, b, base, v, y, rows, K);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}
}  // namespace ipdm

// ------------------------------------------------------------------------------------ topology
namespace {

enum LayerKind { L_CONV, L_RES, L_ATTN, L_DOWN, L_UP };
struct Layer { LayerKind kind; int cin, cout; std::string prefix; };
struct Topology {
    std::vector<std::vector<Layer>> down, up;
    std::vector<Layer> middle;
    int out_ch_in;      // channels entering self.out
    int ted;            // time embedding dim
};

// norm_layer / factor (Model/model.py:69-90)
int gn_groups(int ch)
{
    if (ch % 32 == 0) return 32;
    if (ch < 32) return ch;
    std::vector<int> f;
    int lim = (int)std::sqrt((double)ch);
    for (int i = 1; i <= lim; ++i)
        if (ch % i == 0) {
            f.push_back(i);
            int t = ch / i;
            if (t != i) f.push_back(t);
        }
    int best = f[0];
    long bd = (long)(f[0] - 32) * (f[0] - 32);
    for (size_t k = 1; k < f.size(); ++k) {
        long dd = (long)(f[k] - 32) * (f[k] - 32);
        if (dd < bd) { bd = dd; best = f[k]; }      // first argmin
    }
    return best;
}

std::string pfx(const char *blk, int bi, int li)
{
    char buf[64];
    if (bi >= 0) snprintf(buf, sizeof buf, "%s.%d.%d", blk, bi, li);
    else snprintf(buf, sizeof buf, "%s.%d", blk, li);
    return buf;
}

bool in_attn(const ipdm_unet_cfg &c, int ds)
{
    for (int i = 0; i < c.n_attn; ++i)
        if (c.attention_resolutions[i] == ds) return true;
    return false;
}

// mirrors UNetModel.__init__ (Model/model.py:224-281); int(mult*model_channels) truncation included
Topology build_topology(const ipdm_unet_cfg &c)
{
    Topology t;
    const int mc = c.model_channels;
    t.ted = 4 * mc;
    int ch = (int)(c.channel_mult[0] * mc);
    t.down.push_back({Layer{L_CONV, c.in_channels, ch, pfx("down_blocks", 0, 0)}});
    std::vector<int> chans{ch};
    int ds = 1;
    const int nlev = c.n_mult - 1;
    for (int level = 0; level < nlev; ++level) {
        const double mult = c.channel_mult[level + 1];
        for (int r = 0; r < c.num_res_blocks; ++r) {
            const int bi = (int)t.down.size();
            std::vector<Layer> layers;
            const int co = (int)(mult * mc);
            layers.push_back(Layer{L_RES, ch, co, pfx("down_blocks", bi, 0)});
            ch = co;
            if (in_attn(c, ds)) layers.push_back(Layer{L_ATTN, ch, ch, pfx("down_blocks", bi, 1)});
            t.down.push_back(layers);
            chans.push_back(ch);
        }
        if (level != nlev - 1) {
            const int bi = (int)t.down.size();
            t.down.push_back({Layer{L_DOWN, ch, ch, pfx("down_blocks", bi, 0)}});
            chans.push_back(ch);
            ds *= 2;
        }
    }
    t.middle = {Layer{L_RES, ch, ch, pfx("middle_block", -1, 0)}, Layer{L_ATTN, ch, ch, pfx("middle_block", -1, 1)},
                Layer{L_RES, ch, ch, pfx("middle_block", -1, 2)}};
    for (int level = nlev - 1; level >= 0; --level) {
        const double mult = c.channel_mult[level + 1];
        for (int i = 0; i <= c.num_res_blocks; ++i) {
            const int bi = (int)t.up.size();
            std::vector<Layer> layers;
            const int skip = chans.back();
            chans.pop_back();
            const int co = (int)(mc * mult);
            layers.push_back(Layer{L_RES, ch + skip, co, pfx("up_blocks", bi, 0)});
            ch = co;
            if (in_attn(c, ds)) layers.push_back(Layer{L_ATTN, ch, ch, pfx("up_blocks", bi, (int)layers.size())});
            if (level && i == c.num_res_blocks) {
                layers.push_back(Layer{L_UP, ch, ch, pfx("up_blocks", bi, (int)layers.size())});
                ds /= 2;
            }
            t.up.push_back(layers);
        }
    }
    t.out_ch_in = ch;
    return t;
}

struct ParamInfo { std::string name; int shape[4]; int ndim; };

void add_p(std::vector<ParamInfo> &v, const std::string &name, int a, int b = 0, int c = 0, int d = 0)
{
    ParamInfo p;
    p.name = name;
    p.shape[0] = a; p.shape[1] = b ? b : 1; p.shape[2] = c ? c : 1; p.shape[3] = d ? d : 1;
    p.ndim = d ? 4 : (c ? 3 : (b ? 2 : 1));
    v.push_back(p);
}

void layer_params(std::vector<ParamInfo> &v, const Layer &l, int ted)
{
    const std::string &p = l.prefix;
    switch (l.kind) {
        case L_CONV:
            add_p(v, p + ".weight", l.cout, l.cin, 3, 3); add_p(v, p + ".bias", l.cout); break;
        case L_RES:
            add_p(v, p + ".conv1.0.weight", l.cin); add_p(v, p + ".conv1.0.bias", l.cin);
            add_p(v, p + ".conv1.2.weight", l.cout, l.cin, 3, 3); add_p(v, p + ".conv1.2.bias", l.cout);
            add_p(v, p + ".time_emb.1.weight", l.cout, ted); add_p(v, p + ".time_emb.1.bias", l.cout);
            add_p(v, p + ".conv2.0.weight", l.cout); add_p(v, p + ".conv2.0.bias", l.cout);
            add_p(v, p + ".conv2.2.weight", l.cout, l.cout, 3, 3); add_p(v, p + ".conv2.2.bias", l.cout);
            if (l.cin != l.cout) { add_p(v, p + ".shortcut.weight", l.cout, l.cin, 1, 1); add_p(v, p + ".shortcut.bias", l.cout); }
            break;
        case L_ATTN:
            add_p(v, p + ".norm.weight", l.cin); add_p(v, p + ".norm.bias", l.cin);
            add_p(v, p + ".qkv.weight", 3 * l.cin, l.cin, 1, 1);
            add_p(v, p + ".proj.weight", l.cin, l.cin, 1, 1); add_p(v, p + ".proj.bias", l.cin);
            break;
        case L_DOWN:
            add_p(v, p + ".op.weight", l.cout, l.cin, 3, 3); add_p(v, p + ".op.bias", l.cout); break;
        case L_UP:
            add_p(v, p + ".conv.weight", l.cout, l.cin, 3, 3); add_p(v, p + ".conv.bias", l.cout); break;
    }
}

std::vector<ParamInfo> list_params(const ipdm_unet_cfg &c, const Topology &t)
{
    std::vector<ParamInfo> v;
    const int mc = c.model_channels;
    add_p(v, "time_embed.0.weight", t.ted, mc); add_p(v, "time_embed.0.bias", t.ted);
    add_p(v, "time_embed.2.weight", t.ted, t.ted); add_p(v, "time_embed.2.bias", t.ted);
    for (auto &blk : t.down) for (auto &l : blk) layer_params(v, l, t.ted);
    for (auto &l : t.middle) layer_params(v, l, t.ted);
    for (auto &blk : t.up) for (auto &l : blk) layer_params(v, l, t.ted);
    add_p(v, "out.0.weight", t.out_ch_in); add_p(v, "out.0.bias", t.out_ch_in);
    add_p(v, "out.2.weight", c.out_channels, t.out_ch_in, 3, 3); add_p(v, "out.2.bias", c.out_channels);
    return v;
}

bool cfg_ok(const ipdm_unet_cfg *c)
{
    return c && c->in_channels > 0 && c->model_channels > 0 && c->out_channels > 0 && c->num_res_blocks > 0 &&
           c->num_heads > 0 && c->n_mult >= 2 && c->n_mult <= 16 && c->n_attn >= 0 && c->n_attn <= 16;
}

}  // namespace

extern "C" int ipdm_unet_param_count(const ipdm_unet_cfg *cfg)
{
    if (!cfg_ok(cfg)) { set_error("unet_param_count: bad config"); return IPDM_ERR_INVALID; }
    Topology t = build_topology(*cfg);
    return (int)list_params(*cfg, t).size();
}

extern "C" int ipdm_unet_param_info(const ipdm_unet_cfg *cfg, int32_t idx, char *name, int32_t name_cap, int32_t shape[4],
                                    int32_t *ndim)
{
    if (!cfg_ok(cfg)) { set_error("unet_param_info: bad config"); return IPDM_ERR_INVALID; }
    Topology t = build_topology(*cfg);
    auto v = list_params(*cfg, t);
    IPDM_REQUIRE(idx >= 0 && idx < (int)v.size() && name && shape && ndim, "unet_param_info: bad index %d", idx);
    snprintf(name, name_cap, "%s", v[idx].name.c_str());
    for (int i = 0; i < 4; ++i) shape[i] = v[idx].shape[i];
    *ndim = v[idx].ndim;
    return IPDM_OK;
}

// ------------------------------------------------------------------------------------ the net
namespace {

// w_t: the same weights with the two kernel axes swapped (3x3 only): what the convolution needs when the executor runs
// a forward on spatially TRANSPOSED activations (run_forward: orientation)
// w_up2 / w_up2_t: the Upsample convolutions' parity form (conv_pack_weights_up2), null elsewhere
struct ConvP { float *w = nullptr; float *w_t = nullptr; float *b = nullptr; int cin = 0, cout = 0, ks = 0, cout_pad = 0, interleave = 0;
               float *w_up2 = nullptr, *w_up2_t = nullptr;
               float *w_wup2 = nullptr, *w_wup2_t = nullptr;      // ... in the F(2x2,2x2) domain (conv_pack_weights_wup2), wide layers
               float *w_wino = nullptr, *w_wino_t = nullptr; };      // Winograd-domain weights (conv_pack_weights_wino), both orientations
struct NormP { float *g = nullptr, *b = nullptr; int ch = 0, groups = 0; };
struct ResP { NormP n1, n2; ConvP c1, c2, sc; bool has_sc = false; int bias_off = 0;      // bias_off into bias_eff
              float *b2sc = nullptr; };      // conv2's bias + the shortcut's (the fused form of the narrow levels: conv_direct_skip_ok)
struct AttnP { NormP n; ConvP qkv, proj; };

struct Arena {
    // first-fit free list over a byte range; replayed identically on every forward
    struct Blk { size_t off, size; };
    std::vector<Blk> free_;
    size_t cap = 0, high = 0;
    void reset(size_t capacity) { cap = capacity; free_.assign(1, Blk{0, capacity}); high = 0; }
    size_t alloc(size_t bytes)
    {
        bytes = align_up(bytes ? bytes : 1, 256);
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].size >= bytes) {
                size_t off = free_[i].off;
                free_[i].off += bytes;
                free_[i].size -= bytes;
                if (!free_[i].size) free_.erase(free_.begin() + i);
                if (off + bytes > high) high = off + bytes;
                return off;
            }
        return (size_t)-1;
    }
    void release(size_t off, size_t bytes)
    {
        bytes = align_up(bytes ? bytes : 1, 256);
        size_t i = 0;
        while (i < free_.size() && free_[i].off < off) ++i;
        free_.insert(free_.begin() + i, Blk{off, bytes});
        if (i + 1 < free_.size() && free_[i].off + free_[i].size == free_[i + 1].off) {
            free_[i].size += free_[i + 1].size;
            free_.erase(free_.begin() + i + 1);
        }
        if (i > 0 && free_[i - 1].off + free_[i - 1].size == free_[i].off) {
            free_[i - 1].size += free_[i].size;
            free_.erase(free_.begin() + i);
        }
    }
};

// st[]: where the per-tile partial sums of this tensor's channels live (written by the convolution that produced it,
// ConvArgs::stats); a materialised channel concat refers to the rows of its two sources (nst = 2).  nst = 0: none.
struct StatRef { size_t off = 0; int rows = 0, C = 0; };
struct Tensor { size_t off = (size_t)-1; size_t bytes = 0; int C = 0, H = 0, W = 0; int refs = 0; bool external = false; const float *ext = nullptr;
                size_t st_off = (size_t)-1, st_bytes = 0; StatRef st[2]; int nst = 0;
                bool planar = false; };      // stored parity-planar [B][C][2][2][H/2][W/2]: the output of an up2 convolution

}  // namespace

struct ipdm_unet {
    ipdm_unet_cfg cfg;
    Topology topo;
    std::vector<float *> owned;                 // device allocations
    std::map<std::string, ConvP> convs;
    std::map<std::string, ResP> res;
    std::map<std::string, AttnP> attn;
    NormP out_norm;
    ConvP out_conv;
    float *d_freqs = nullptr, *te_w0 = nullptr, *te_b0 = nullptr, *te_w2 = nullptr, *te_b2 = nullptr;
    float *temb_W = nullptr, *temb_b = nullptr, *conv1_b = nullptr;   // concatenated over all ResidualBlocks
    int temb_rows = 0;
    int max_gn_groups = 32, max_ch = 0;
    bool transposed = false;                    // this forward runs on [B,C,W,H] activations (run_forward: orientation)
    bool no_fused_stats = false;                // option gn_unfused: GroupNorm statistics by a pass over the activations (gn_partial)
    int opts[OPT_COUNT] = {};                   // the option table as it stood at ipdm_unet_create (weights were packed under it)

    // forward state
    Arena arena;
    char *ws = nullptr;
    bool dry = false;
    hipStream_t st = nullptr;
    int B = 0;
    float *bias_eff = nullptr, *gn_scale = nullptr, *gn_shift = nullptr;
    double *gn_part = nullptr;
    std::vector<Tensor *> live;
    // captured forwards (ipdm_unet_forward_graph): one executable graph per (t, batch, size, buffers)
    struct GraphKey {
        int t, B, H, W; const void *x, *eps, *ws;
        std::vector<int> mode;                  // the values of EVERY per-call option at capture time (opt_per_call_values)
        bool operator<(const GraphKey &o) const
        {
            return std::tie(t, B, H, W, x, eps, ws, mode) < std::tie(o.t, o.B, o.H, o.W, o.x, o.eps, o.ws, o.mode);
        }
    };
    std::map<GraphKey, hipGraphExec_t> graphs;
    std::map<GraphKey, int> graph_seen;
    std::map<GraphKey, size_t> graph_need;      // workspace bytes the captured walk addresses
};

namespace {

#ifndef IPDM_UNET_TRACE
#define IPDM_UNET_TRACE 0           // diagnostic builds only (tools/build_variants.sh unet.hip trace:-DIPDM_UNET_TRACE=1 ...; loaded through IPDM_LIB_PATH)
#endif
#if IPDM_UNET_TRACE
// IPDM_UNET_TRACE=1: the host waits for every convolution and sums its output; =2: a kernel behind every convolution adds the
// output's words into a device slot (no host wait: the launches stay back to back), ipdm_trace_dump() prints the slots.
__global__ void trace_sum_kernel(const unsigned *p, size_t n, unsigned long long *slot)
{
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += (unsigned long long)p[i] * (i % 1021 + 1);
    atomicAdd(slot, s);
}
std::vector<std::string> g_trace_desc;
unsigned long long *g_trace_slots = nullptr;
struct TraceRec { size_t off, n; int B, C, H, W; size_t off_x1 = 0, off_sc = 0, off_sh = 0; int C1 = 0, code = 0; const float *w = nullptr; };
std::vector<TraceRec> g_trace_rec;
char *g_trace_pool = nullptr;
size_t g_trace_used = 0;
const size_t g_trace_pool_bytes = (size_t)48 << 30;
void trace_conv(const ConvArgs &a, size_t n, hipStream_t st)
{
    char desc[256];
    snprintf(desc, sizeof(desc), "code %2d  %3d+%-3d -> %3d @%4dx%-4d s%d act %d res %d planar %d stats %d", conv_kernel_code(a), a.C1, a.C2, a.Cout, a.Ho, a.Wo,
             a.stride, a.act, a.res ? 1 : 0, a.x1_planar, a.stats ? a.stats_rows : 0);
#if IPDM_UNET_TRACE == 1
    std::vector<unsigned> h(n);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), a.out, n * 4, hipMemcpyDeviceToHost);
    unsigned long long sum = 0;
    for (size_t i = 0; i < n; ++i) sum += (unsigned long long)h[i] * (i % 1021 + 1);
    fprintf(stderr, "TRACE %4zu %s  %016llx\n", g_trace_desc.size(), desc, sum);
    g_trace_desc.push_back(desc);
#elif IPDM_UNET_TRACE == 3
    // a device copy of every convolution's output (one pool, carved up front: no allocation between the launches), compared by ipdm_trace_dump()
    if (!g_trace_pool) { (void)hipMalloc((void **)&g_trace_pool, g_trace_pool_bytes); }
    if (g_trace_used + n * 4 <= g_trace_pool_bytes) {
        (void)hipMemcpyAsync(g_trace_pool + g_trace_used, a.out, n * 4, hipMemcpyDeviceToDevice, st);
        TraceRec rec{g_trace_used, n, a.B, a.Cout, a.Ho, a.Wo};
        g_trace_used += (n * 4 + 255) / 256 * 256;
        rec.code = conv_kernel_code(a);
        const size_t nx = (size_t)a.B * a.C1 * a.Hs * a.Ws * 4, ng = (size_t)a.B * (a.C1 + a.C2) * 4;
        if (rec.code == 12 && a.C2 == 0 && a.act && g_trace_used + nx + 2 * ng + 1024 <= g_trace_pool_bytes) {      // the inputs too: ipdm_trace_dump() writes them out for the first wrong one
            rec.C1 = a.C1; rec.w = a.w_wino;
            rec.off_x1 = g_trace_used; (void)hipMemcpyAsync(g_trace_pool + g_trace_used, a.x1, nx, hipMemcpyDeviceToDevice, st); g_trace_used += (nx + 255) / 256 * 256;
            rec.off_sc = g_trace_used; (void)hipMemcpyAsync(g_trace_pool + g_trace_used, a.gn_scale, ng, hipMemcpyDeviceToDevice, st); g_trace_used += (ng + 255) / 256 * 256;
            rec.off_sh = g_trace_used; (void)hipMemcpyAsync(g_trace_pool + g_trace_used, a.gn_shift, ng, hipMemcpyDeviceToDevice, st); g_trace_used += (ng + 255) / 256 * 256;
        }
        g_trace_rec.push_back(rec);
        g_trace_desc.push_back(desc);
    }
#else
    if (!g_trace_slots) { (void)hipMalloc((void **)&g_trace_slots, 8192 * 8); (void)hipMemset(g_trace_slots, 0, 8192 * 8); }
    if (g_trace_desc.size() < 8192) {
        hipLaunchKernelGGL(trace_sum_kernel, dim3(512), dim3(256), 0, st, (const unsigned *)a.out, n, g_trace_slots + g_trace_desc.size());
        g_trace_desc.push_back(desc);
    }
#endif
}
#endif

int upload(ipdm_unet *net, const float *host, size_t n, float **out)
{
    float *d = nullptr;
    IPDM_HIP_CHECK(hipMalloc((void **)&d, (n ? n : 1) * sizeof(float)));
    if (n) IPDM_HIP_CHECK(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
    net->owned.push_back(d);
    *out = d;
    return IPDM_OK;
}

struct WeightMap {
    std::map<std::string, const float *> m;
    const float *get(const std::string &k) const
    {
        auto it = m.find(k);
        return it == m.end() ? nullptr : it->second;
    }
};

int make_conv(ipdm_unet *net, const WeightMap &wm, const std::string &wname, const std::string &bname, int cout, int cin,
              int ks, ConvP &out, int stride = 1, bool up = false)
{
    const float *w = wm.get(wname);
    IPDM_REQUIRE(w, "unet_create: missing parameter %s", wname.c_str());
    std::vector<float> packed;
    int cin_pad, cout_pad;
    out.interleave = conv_weight_interleave(cout, ks, stride);
    conv_pack_weights(w, cout, cin, ks, out.interleave, packed, cin_pad, cout_pad);
    out.cin = cin; out.cout = cout; out.ks = ks; out.cout_pad = cout_pad;
    int rc = upload(net, packed.data(), packed.size(), &out.w);
    if (rc) return rc;
    out.w_t = out.w;
    if (ks == 3) {
        std::vector<float> wt((size_t)cout * cin * 9);
        for (size_t oc = 0; oc < (size_t)cout * cin; ++oc)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) wt[oc * 9 + ky * 3 + kx] = w[oc * 9 + kx * 3 + ky];
        conv_pack_weights(wt.data(), cout, cin, ks, out.interleave, packed, cin_pad, cout_pad);
        rc = upload(net, packed.data(), packed.size(), &out.w_t);
        if (rc) return rc;
        if (!up && conv_wino_shape_ok(cout, cin, ks, stride, out.interleave)) {
            conv_pack_weights_wino(w, cout, cin, packed);
            if ((rc = upload(net, packed.data(), packed.size(), &out.w_wino))) return rc;
            conv_pack_weights_wino(wt.data(), cout, cin, packed);
            if ((rc = upload(net, packed.data(), packed.size(), &out.w_wino_t))) return rc;
        }
        if (up && (out.interleave == 2 || out.interleave == 4 || (out.interleave == 0 && cout <= 16))) {      // Upsample: the parity form of both orientations
            conv_pack_weights_up2(w, cout, cin, out.interleave, packed);
            if ((rc = upload(net, packed.data(), packed.size(), &out.w_up2))) return rc;
            conv_pack_weights_up2(wt.data(), cout, cin, out.interleave, packed);
            if ((rc = upload(net, packed.data(), packed.size(), &out.w_up2_t))) return rc;
            if (out.interleave && conv_wup2_shape_ok(cout, cin)) {
                conv_pack_weights_wup2(w, cout, cin, packed);
                if ((rc = upload(net, packed.data(), packed.size(), &out.w_wup2))) return rc;
                conv_pack_weights_wup2(wt.data(), cout, cin, packed);
                if ((rc = upload(net, packed.data(), packed.size(), &out.w_wup2_t))) return rc;
            }
        }
    }
    out.b = nullptr;
    if (!bname.empty()) {
        const float *b = wm.get(bname);
        IPDM_REQUIRE(b, "unet_create: missing parameter %s", bname.c_str());
        rc = upload(net, b, cout, &out.b);
    }
    return rc;
}

int make_norm(ipdm_unet *net, const WeightMap &wm, const std::string &p, int ch, NormP &out)
{
    const float *g = wm.get(p + ".weight"), *b = wm.get(p + ".bias");
    IPDM_REQUIRE(g && b, "unet_create: missing parameter %s.{weight,bias}", p.c_str());
    out.ch = ch;
    out.groups = gn_groups(ch);
    int rc = upload(net, g, ch, &out.g);
    if (rc) return rc;
    if (out.groups > net->max_gn_groups) net->max_gn_groups = out.groups;
    if (ch > net->max_ch) net->max_ch = ch;
    return upload(net, b, ch, &out.b);
}

}  // namespace

extern "C" int ipdm_unet_create(const ipdm_unet_cfg *cfg, const float *const *weights, int32_t n_weights, ipdm_unet **out)
{
    IPDM_REQUIRE(cfg_ok(cfg) && weights && out, "unet_create: bad argument");
    IPDM_REQUIRE(cfg->model_channels % 2 == 0, "unet_create: model_channels must be even");
    ipdm_unet *net = new ipdm_unet();
    opt_snapshot(net->opts);
    net->cfg = *cfg;
    net->topo = build_topology(*cfg);
    auto plist = list_params(*cfg, net->topo);
    if ((int)plist.size() != n_weights) {
        set_error("unet_create: expected %d parameters, got %d", (int)plist.size(), n_weights);
        delete net;
        return IPDM_ERR_INVALID;
    }
    WeightMap wm;
    for (size_t i = 0; i < plist.size(); ++i) {
        if (!weights[i]) { set_error("unet_create: parameter %s is null", plist[i].name.c_str()); delete net; return IPDM_ERR_INVALID; }
        wm.m[plist[i].name] = weights[i];
    }
    const int mc = cfg->model_channels, ted = net->topo.ted;
    int rc = 0;
#define TRY(x) do { rc = (x); if (rc) { ipdm_unet_destroy(net); return rc; } } while (0)
    // sinusoid frequencies in float32 exactly as timestep_embedding builds them (Model/model.py:24-27):
    // exp(-log(1e4) * arange(half) / half) evaluated in float32
    {
        const int half = mc / 2;
        std::vector<float> f(half);
        const float a = (float)(-std::log(10000.0));
        for (int k = 0; k < half; ++k) f[k] = expf((a * (float)k) / (float)half);
        TRY(upload(net, f.data(), half, &net->d_freqs));
    }
    TRY(upload(net, wm.get("time_embed.0.weight"), (size_t)ted * mc, &net->te_w0));
    TRY(upload(net, wm.get("time_embed.0.bias"), ted, &net->te_b0));
    TRY(upload(net, wm.get("time_embed.2.weight"), (size_t)ted * ted, &net->te_w2));
    TRY(upload(net, wm.get("time_embed.2.bias"), ted, &net->te_b2));

    std::vector<float> tW, tb, c1b;
    auto do_layer = [&](const Layer &l) -> int {
        const std::string &p = l.prefix;
        int r = 0;
        switch (l.kind) {
            case L_CONV: r = make_conv(net, wm, p + ".weight", p + ".bias", l.cout, l.cin, 3, net->convs[p]); break;
            case L_DOWN: r = make_conv(net, wm, p + ".op.weight", p + ".op.bias", l.cout, l.cin, 3, net->convs[p], 2); break;
            case L_UP: r = make_conv(net, wm, p + ".conv.weight", p + ".conv.bias", l.cout, l.cin, 3, net->convs[p], 1, true); break;
            case L_RES: {
                ResP &rp = net->res[p];
                if ((r = make_norm(net, wm, p + ".conv1.0", l.cin, rp.n1))) break;
                if ((r = make_conv(net, wm, p + ".conv1.2.weight", "", l.cout, l.cin, 3, rp.c1))) break;
                if ((r = make_norm(net, wm, p + ".conv2.0", l.cout, rp.n2))) break;
                if ((r = make_conv(net, wm, p + ".conv2.2.weight", p + ".conv2.2.bias", l.cout, l.cout, 3, rp.c2))) break;
                rp.has_sc = l.cin != l.cout;
                if (rp.has_sc && (r = make_conv(net, wm, p + ".shortcut.weight", p + ".shortcut.bias", l.cout, l.cin, 1, rp.sc))) break;
                if (rp.has_sc) {
                    const float *b2 = wm.get(p + ".conv2.2.bias"), *bs = wm.get(p + ".shortcut.bias");
                    std::vector<float> both(l.cout);
                    for (int c = 0; c < l.cout; ++c) both[c] = b2[c] + bs[c];
                    if ((r = upload(net, both.data(), both.size(), &rp.b2sc))) break;
                }
                const float *tw = wm.get(p + ".time_emb.1.weight"), *tbb = wm.get(p + ".time_emb.1.bias"),
                            *cb = wm.get(p + ".conv1.2.bias");
                if (!tw || !tbb || !cb) { set_error("unet_create: missing time_emb/conv1 bias of %s", p.c_str()); r = IPDM_ERR_INVALID; break; }
                rp.bias_off = (int)tb.size();
                tW.insert(tW.end(), tw, tw + (size_t)l.cout * ted);
                tb.insert(tb.end(), tbb, tbb + l.cout);
                c1b.insert(c1b.end(), cb, cb + l.cout);
                break;
            }
            case L_ATTN: {
                AttnP &ap = net->attn[p];
                if ((r = make_norm(net, wm, p + ".norm", l.cin, ap.n))) break;
                if ((r = make_conv(net, wm, p + ".qkv.weight", "", 3 * l.cin, l.cin, 1, ap.qkv))) break;
                r = make_conv(net, wm, p + ".proj.weight", p + ".proj.bias", l.cin, l.cin, 1, ap.proj);
                if (!r && l.cin / cfg->num_heads != 64 && l.cin / cfg->num_heads != 32) { set_error("unet_create: attention head dim %d unsupported (64 or 32)", l.cin / cfg->num_heads); r = IPDM_ERR_UNSUPPORTED; }
                break;
            }
        }
        return r;
    };
    for (auto &blk : net->topo.down) for (auto &l : blk) TRY(do_layer(l));
    for (auto &l : net->topo.middle) TRY(do_layer(l));
    for (auto &blk : net->topo.up) for (auto &l : blk) TRY(do_layer(l));
    TRY(make_norm(net, wm, "out.0", net->topo.out_ch_in, net->out_norm));
    TRY(make_conv(net, wm, "out.2.weight", "out.2.bias", cfg->out_channels, net->topo.out_ch_in, 3, net->out_conv));
    net->temb_rows = (int)tb.size();
    TRY(upload(net, tW.data(), tW.size(), &net->temb_W));
    TRY(upload(net, tb.data(), tb.size(), &net->temb_b));
    TRY(upload(net, c1b.data(), c1b.size(), &net->conv1_b));
#undef TRY
    // the largest GN input is a concatenated up-block input
    for (auto &blk : net->topo.up) for (auto &l : blk) if (l.cin > net->max_ch) net->max_ch = l.cin;
    *out = net;
    return IPDM_OK;
}

extern "C" int ipdm_unet_destroy(ipdm_unet *net)
{
    if (!net) return IPDM_OK;
    for (auto &kv : net->graphs) (void)hipGraphExecDestroy(kv.second);
    for (float *p : net->owned) (void)hipFree(p);
    delete net;
    return IPDM_OK;
}

// ------------------------------------------------------------------------------------ forward walk
namespace {

struct Fwd {
    ipdm_unet *net;
    int rc = 0;

    const float *ptr(const Tensor *t) const { return t->external ? t->ext : (const float *)(net->ws + t->off); }
    float *wptr(Tensor *t) const { return (float *)(net->ws + t->off); }

    Tensor *make(int C, int H, int W)
    {
        Tensor *t = new Tensor();
        t->C = C; t->H = H; t->W = W;
        t->bytes = (size_t)net->B * C * H * W * sizeof(float);
        t->off = net->arena.alloc(t->bytes);
        if (t->off == (size_t)-1) { set_error("unet_forward: workspace exhausted"); rc = IPDM_ERR_WORKSPACE; t->off = 0; }
        t->refs = 1;
        net->live.push_back(t);
        return t;
    }
    void retain(Tensor *t) { t->refs++; }
    void release(Tensor *t)
    {
        if (--t->refs == 0) {
            if (!t->external) net->arena.release(t->off, t->bytes);
            if (t->st_off != (size_t)-1) net->arena.release(t->st_off, t->st_bytes);
            for (size_t i = 0; i < net->live.size(); ++i)
                if (net->live[i] == t) { net->live.erase(net->live.begin() + i); break; }
            delete t;
        }
    }

    // a parity-planar tensor for a reader that takes NCHW only: returns a converted copy (caller releases), or null when
    // t is NCHW already
    Tensor *linear_copy(const Tensor *t)
    {
        if (!t || !t->planar) return nullptr;
        Tensor *o = make(t->C, t->H, t->W);
        if (!rc && !net->dry) rc = planar_to_linear_launch(ptr(t), wptr(o), (long)net->B * t->C, t->H, t->W, net->st);
        return o;
    }

    void gn(const Tensor *x1, const Tensor *x2, const NormP &np)
    {
        if (rc || net->dry) return;
        // statistics from the per-tile partial sums the producing convolutions left behind, when every source has them
        StatRef refs[2];
        int nref = 0;
        bool have = !net->no_fused_stats;
        for (const Tensor *t : {x1, x2}) {
            if (!t) continue;
            if (t->nst == 0 || nref + t->nst > 2) { have = false; break; }
            for (int k = 0; k < t->nst; ++k) refs[nref++] = t->st[k];
        }
        if (have) {
            GnTileArgs g;
            g.nsrc = nref;
            for (int k = 0; k < nref; ++k) { g.src[k].stats = (const float *)(net->ws + refs[k].off); g.src[k].rows = refs[k].rows; g.src[k].C = refs[k].C; }
            g.B = net->B; g.HW = (long)x1->H * x1->W; g.groups = np.groups; g.gamma = np.g; g.beta = np.b; g.eps = 1e-5f;
            g.partials = net->gn_part; g.scale = net->gn_scale; g.shift = net->gn_shift;
            rc = gn_tiles_launch(g, net->st);
            return;
        }
        GnArgs a;
        a.x1 = ptr(x1); a.x2 = x2 ? ptr(x2) : nullptr;
        a.C1 = x1->C; a.C2 = x2 ? x2->C : 0; a.B = net->B;
        a.HW = (long)x1->H * x1->W;
        a.groups = np.groups; a.gamma = np.g; a.beta = np.b; a.eps = 1e-5f;
        a.partials = net->gn_part; a.scale = net->gn_scale; a.shift = net->gn_shift;
        rc = gn_stats_launch(a, net->st);
    }

    // conv over (x1 [,x2]) optionally nearest-upsampled to (H,W); returns a new tensor
    // want_stats: the output feeds a GroupNorm -> the kernel also leaves per-tile partial sums of it (when it can)
    // sk_*: the block input and shortcut weights of a fused 1x1 shortcut (res_block decides; ConvArgs::sk_*)
    Tensor *conv(const Tensor *x1, const Tensor *x2, const ConvP &cp, int stride, int act, const float *bias,
                 const Tensor *res, int H, int W, float *ext_out = nullptr, bool want_stats = false,
                 const Tensor *sk_x1 = nullptr, const Tensor *sk_x2 = nullptr, const ConvP *sk = nullptr)
    {
        const int pad = cp.ks / 2;
        const int Ho = (H + 2 * pad - cp.ks) / stride + 1, Wo = (W + 2 * pad - cp.ks) / stride + 1;
        Tensor *o;
        if (ext_out) { o = new Tensor(); o->C = cp.cout; o->H = Ho; o->W = Wo; o->external = true; o->ext = ext_out; o->refs = 1; net->live.push_back(o); }
        else o = make(cp.cout, Ho, Wo);
        ConvArgs a;
        // (shape fields first: the statistics geometry depends on the kernel the dispatcher picks, in dry runs too)
        a.C1 = x1->C; a.C2 = x2 ? x2->C : 0; a.B = net->B; a.Cout = cp.cout; a.ksize = cp.ks; a.stride = stride; a.Ho = Ho; a.Wo = Wo;
        a.w_interleave = cp.interleave; a.cout_pad = cp.cout_pad;
        a.Hs = x1->H; a.Ws = x1->W; a.H = H; a.W = W; a.upsample = (H != x1->H || W != x1->W); a.act = act; a.res = res ? (const float *)(uintptr_t)256 : nullptr;
        a.w_up2 = net->transposed ? cp.w_up2_t : cp.w_up2;
        a.w_wup2 = net->transposed ? cp.w_wup2_t : cp.w_wup2;
        a.w_wino = net->transposed ? cp.w_wino_t : cp.w_wino;
        // parity-planar sources (outputs of up2 convolutions): x1 of the kernels that can read them, converted otherwise
        Tensor *lin1 = nullptr, *lin2 = linear_copy(x2), *linr = linear_copy(res);
        if (x1->planar && !conv_planar_ok(a)) lin1 = linear_copy(x1);
        if (lin1) x1 = lin1;
        if (lin2) x2 = lin2;
        if (linr) res = linr;
        a.x1_planar = x1->planar ? 1 : 0;
        Tensor *lins = sk ? linear_copy(sk_x2) : nullptr;
        if (lins) sk_x2 = lins;
        struct LinGuard { Fwd &f; Tensor *a, *b, *c, *d; ~LinGuard() { if (a) f.release(a); if (b) f.release(b); if (c) f.release(c); if (d) f.release(d); } }
            lin_guard{*this, lin1, lin2, linr, lins};
        if (sk) {
            a.sk_C1 = sk_x1->C; a.sk_C2 = sk_x2 ? sk_x2->C : 0; a.sk_cout_pad = sk->cout_pad; a.sk_planar = sk_x1->planar ? 1 : 0;
            a.sk_w = sk->w;
        }
        if (conv_up2_eligible(a) && !ext_out) o->planar = true;                 // wide levels: parity-planar output
        else if (!conv_direct_up2_eligible(a)) a.w_up2 = nullptr;               // (narrow levels: the direct kernel's parity form writes NCHW)
        // layers with too few tiles to fill the chip are split along K into a scratch buffer (conv_ws.hip)
        const size_t split_bytes = conv_split_ws_bytes(a);
        size_t split_off = (size_t)-1;
        if (split_bytes) {
            split_off = net->arena.alloc(split_bytes);
            if (split_off == (size_t)-1) { set_error("unet_forward: workspace exhausted"); rc = IPDM_ERR_WORKSPACE; split_off = 0; }
            a.split_ws = (float *)(net->ws + split_off);
            if (!a.split_ws) a.split_ws = reinterpret_cast<float *>((uintptr_t)256);
        }
        struct SplitGuard { Arena &ar; size_t off, bytes; ~SplitGuard() { if (off != (size_t)-1) ar.release(off, bytes); } } split_guard{net->arena, split_off, split_bytes};
        if (want_stats && !ext_out && !net->no_fused_stats) {
            const int rows = conv_stats_rows(a);
            if (rows > 0) {
                o->st_bytes = (size_t)net->B * rows * cp.cout * 2 * sizeof(float);
                o->st_off = net->arena.alloc(o->st_bytes);
                if (o->st_off == (size_t)-1) { set_error("unet_forward: workspace exhausted"); rc = IPDM_ERR_WORKSPACE; o->st_off = 0; }
                o->nst = 1;
                o->st[0].off = o->st_off; o->st[0].rows = rows; o->st[0].C = cp.cout;
            }
        }
        if (rc || net->dry) return o;
        if (o->nst) { a.stats = (float *)(net->ws + o->st_off); a.stats_rows = o->st[0].rows; }
        a.x1 = ptr(x1); a.x2 = x2 ? ptr(x2) : nullptr;
        a.C1 = x1->C; a.C2 = x2 ? x2->C : 0; a.B = net->B;
        a.Hs = x1->H; a.Ws = x1->W; a.H = H; a.W = W;
        a.upsample = (H != x1->H || W != x1->W);
        a.scale_y = (float)x1->H / (float)H; a.scale_x = (float)x1->W / (float)W;
        a.w = net->transposed ? cp.w_t : cp.w; a.cout_pad = cp.cout_pad; a.w_interleave = cp.interleave; a.bias = bias; a.Cout = cp.cout; a.ksize = cp.ks; a.stride = stride;
        a.Ho = Ho; a.Wo = Wo; a.act = act; a.gn_scale = net->gn_scale; a.gn_shift = net->gn_shift;
        a.res = res ? ptr(res) : nullptr;
        if (sk) { a.sk_x1 = ptr(sk_x1); a.sk_x2 = sk_x2 ? ptr(sk_x2) : nullptr; }
        a.out = ext_out ? ext_out : wptr(o);
        a.tiles_x = a.tiles_y = a.co_tiles = 0;
        rc = conv2d_launch(a, net->st);
#if IPDM_UNET_TRACE      // diagnostic build (tools/build_variants.sh unet.hip trace:-DIPDM_UNET_TRACE=1): a checksum of every convolution's output
        if (!rc) trace_conv(a, (size_t)net->B * cp.cout * Ho * Wo, net->st);
#endif
        return o;
    }

    // ResidualBlock.forward (Model/model.py:121-130); consumes nothing, returns new tensor
    Tensor *res_block(const Tensor *x1, const Tensor *x2, const Layer &l)
    {
        const ResP &rp = net->res[l.prefix];
        const int H = x1->H, W = x1->W;
        gn(x1, x2, rp.n1);
        Tensor *h1 = conv(x1, x2, rp.c1, 1, 2, net->bias_eff + rp.bias_off, nullptr, H, W, nullptr, true);
        Tensor *sc = nullptr;
        const Tensor *resid;
        // narrow levels: the 1x1 shortcut rides on conv2 as extra K chunks over the block input (conv_direct.hip)
        bool fuse = false;
        if (rp.has_sc && rp.b2sc && rp.sc.interleave == 0 && (!x2 || !x2->planar)) {
            ConvArgs q;
            q.C1 = rp.c2.cin; q.C2 = 0; q.B = net->B; q.Cout = rp.c2.cout; q.ksize = 3; q.stride = 1; q.Ho = H; q.Wo = W; q.Hs = H; q.Ws = W; q.H = H; q.W = W;
            q.upsample = 0; q.act = 2; q.res = nullptr; q.w_interleave = rp.c2.interleave; q.cout_pad = rp.c2.cout_pad; q.x1_planar = 0;
            q.w_up2 = nullptr; q.w_wino = nullptr;
            q.sk_C1 = x1->C; q.sk_C2 = x2 ? x2->C : 0; q.sk_cout_pad = rp.sc.cout_pad;
            fuse = conv_direct_skip_ok(q);
        }
        if (fuse) resid = nullptr;
        else if (rp.has_sc) { sc = conv(x1, x2, rp.sc, 1, 0, rp.sc.b, nullptr, H, W); resid = sc; }
        else if (x2) {   // identity shortcut over a concatenated input: materialise the concat
            sc = make(x1->C + x2->C, H, W);
            Tensor *l1 = linear_copy(x1), *l2 = linear_copy(x2);
            struct G { Fwd &f; Tensor *a, *b; ~G() { if (a) f.release(a); if (b) f.release(b); } } g_{*this, l1, l2};
            if (l1) x1 = l1;
            if (l2) x2 = l2;
            if (!rc && !net->dry) {
                const long total = (long)net->B * (x1->C + x2->C) * H * W;
                int g = cdiv(total, 1024); if (g > 4096) g = 4096;
                hipLaunchKernelGGL(concat_kernel, dim3(g), dim3(256), 0, net->st, ptr(x1), ptr(x2), wptr(sc), x1->C, x2->C, (long)H * W, total);
            }
            resid = sc;
        } else resid = x1;
        gn(h1, nullptr, rp.n2);
        Tensor *o = fuse ? conv(h1, nullptr, rp.c2, 1, 2, rp.b2sc, nullptr, H, W, nullptr, true, x1, x2, &rp.sc)
                         : conv(h1, nullptr, rp.c2, 1, 2, rp.c2.b, resid, H, W, nullptr, true);
        release(h1);
        if (sc) release(sc);
        return o;
    }

    // AttentionBlock.forward (Model/model.py:145-155)
    Tensor *attn_block(const Tensor *x, const Layer &l)
    {
        const AttnP &ap = net->attn[l.prefix];
        const int H = x->H, W = x->W;
        gn(x, nullptr, ap.n);
        Tensor *qkv = conv(x, nullptr, ap.qkv, 1, 1, nullptr, nullptr, H, W);
        Tensor *a = make(x->C, H, W);
        // scratch of the key-slice split (partial outputs of short sequences), a rule of the layer's shape
        const int heads = net->cfg.num_heads, hd = x->C / heads;
        const size_t sfl = attention_scratch_floats(net->B, heads, hd, H * W);
        Tensor *scr = sfl ? make((int)((sfl + (size_t)net->B * H * W - 1) / ((size_t)net->B * H * W)), H, W) : nullptr;
        if (!rc && !net->dry) rc = attention_launch(ptr(qkv), wptr(a), net->B, heads, hd, H * W, net->st, scr ? wptr(scr) : nullptr);
        if (scr) release(scr);
        release(qkv);
        Tensor *o = conv(a, nullptr, ap.proj, 1, 0, ap.proj.b, x, H, W, nullptr, true);
        release(a);
        return o;
    }

    // TimestepEmbedSequential.forward (Model/model.py:55-63) over one block's layers.
    // (x1,x2) is the (possibly concatenated) block input; the caller keeps ownership of x1/x2.
    Tensor *run_block(const std::vector<Layer> &layers, Tensor *x1, Tensor *x2, int size_h, int size_w)
    {
        Tensor *h = nullptr;
        Tensor *cat = nullptr;
        // the conv kernels walk K in chunks that must not straddle the two sources; the chunk depends on the kernels the
        // block's first layer runs on (3x3: 8, wave-specialised 1x1 shortcut: 32)
        int align = conv_k_chunk();
        if (x2 && !layers.empty() && layers[0].kind == L_RES) {
            const ResP &rp0 = net->res[layers[0].prefix];
            if (rp0.has_sc) align = std::max(align, conv_ws_k_chunk(1, rp0.sc.interleave));
            align = std::max(align, conv_ws_k_chunk(3, rp0.c1.interleave));
        }
        if (x2 && (x1->C % align) != 0) {
            // the conv kernel walks K in chunks that must not straddle the two sources: materialise torch.cat
            cat = make(x1->C + x2->C, x1->H, x1->W);
            Tensor *l1 = linear_copy(x1), *l2 = linear_copy(x2);
            if (!rc && !net->dry) {
                const long total = (long)net->B * cat->C * x1->H * x1->W;
                int g = cdiv(total, 1024); if (g > 4096) g = 4096;
                hipLaunchKernelGGL(concat_kernel, dim3(g), dim3(256), 0, net->st, ptr(l1 ? l1 : x1), ptr(l2 ? l2 : x2), wptr(cat), x1->C, x2->C, (long)x1->H * x1->W, total);
            }
            if (l1) release(l1);
            if (l2) release(l2);
            if (x1->nst == 1 && x2->nst == 1) { cat->nst = 2; cat->st[0] = x1->st[0]; cat->st[1] = x2->st[0]; }   // rows stay owned by x1 / x2
            x1 = cat;
            x2 = nullptr;
        }
        for (size_t i = 0; i < layers.size(); ++i) {
            const Layer &l = layers[i];
            const Tensor *in1 = h ? h : x1;
            const Tensor *in2 = h ? nullptr : x2;
            Tensor *o = nullptr;
            switch (l.kind) {
                case L_CONV: { const ConvP &cp = net->convs[l.prefix]; o = conv(in1, in2, cp, 1, 0, cp.b, nullptr, in1->H, in1->W, nullptr, true); break; }
                case L_DOWN: { const ConvP &cp = net->convs[l.prefix]; o = conv(in1, in2, cp, 2, 0, cp.b, nullptr, in1->H, in1->W, nullptr, true); break; }
                case L_UP: { const ConvP &cp = net->convs[l.prefix]; o = conv(in1, in2, cp, 1, 0, cp.b, nullptr, size_h, size_w, nullptr, true); break; }
                case L_RES: o = res_block(in1, in2, l); break;
                case L_ATTN: o = attn_block(in1, l); break;
            }
            if (h) release(h);
            h = o;
        }
        if (cat) release(cat);
        return h;
    }
};

size_t fixed_ws_bytes(const ipdm_unet *net, int B)
{
    size_t s = 0;
    s += align_up((size_t)net->topo.ted * 2 * sizeof(float), 256);              // emb tmp + silu(emb)
    s += align_up((size_t)(net->temb_rows ? net->temb_rows : 1) * sizeof(float), 256);   // bias_eff
    s += 2 * align_up(((size_t)B * net->max_ch + 64) * sizeof(float), 256);      // gn scale/shift (+ a K chunk of read-ahead)
    s += align_up(gn_partials_bytes(B, net->max_gn_groups), 256);
    return s;
}

// Padding waste of the 32-column x 8-row MFMA tiling summed over the resolution levels that run on it (more than 32
// channels), weighted by each level's share of the convolution FLOPs (pixels x channels^2): true when the transposed
// orientation wastes at least 2 % less.  IPDM_UNET_TRANSPOSE=0/1 forces the choice.
bool choose_transposed(const ipdm_unet *net, int H, int W)
{
    if (const int force = opt(OPT_UNET_TRANSPOSE); force >= 0) return force != 0;
    if (H == W) return false;
    double cost[2] = {0.0, 0.0};
    int h = H, w = W;
    const int nlev = net->cfg.n_mult - 1;
    for (int level = 0; level < nlev; ++level) {
        const double ch = (double)(int)(net->cfg.channel_mult[level + 1] * net->cfg.model_channels);
        if (ch > 32) {
            const double wgt = ch * ch;
            cost[0] += wgt * (double)(cdiv(w, 32) * 32) * (double)(cdiv(h, 8) * 8);
            cost[1] += wgt * (double)(cdiv(h, 32) * 32) * (double)(cdiv(w, 8) * 8);
        }
        if (level != nlev - 1) { h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1; }
    }
    return cost[1] < 0.98 * cost[0];
}

int run_forward(ipdm_unet *net, const float *d_x, int t, float *d_eps, int B, int H, int W, void *d_ws, size_t ws_bytes,
                hipStream_t st, bool dry, size_t *need)
{
    // an option that shaped the packed weights or the kernel choice may not change under a live handle
    if (const int ch = opt_changed_since(net->opts); ch >= 0) {
        set_error("unet_forward: option '%s' changed after ipdm_unet_create (was %d): create the network again", opt_name(ch), net->opts[ch]);
        return IPDM_ERR_INVALID;
    }
    net->B = B; net->st = st; net->dry = dry; net->ws = (char *)d_ws;
    net->no_fused_stats = opt(OPT_GN_UNFUSED) != 0;
    const size_t fixed = fixed_ws_bytes(net, B);
    if (!dry && ws_bytes < fixed) { set_error("unet_forward: workspace too small"); return IPDM_ERR_WORKSPACE; }
    // fixed region
    char *w = (char *)d_ws;
    float *emb_tmp = (float *)w; float *silu_emb = emb_tmp + net->topo.ted;
    w += align_up((size_t)net->topo.ted * 2 * sizeof(float), 256);
    net->bias_eff = (float *)w; w += align_up((size_t)(net->temb_rows ? net->temb_rows : 1) * sizeof(float), 256);
    net->gn_scale = (float *)w; w += align_up(((size_t)B * net->max_ch + 64) * sizeof(float), 256);
    net->gn_shift = (float *)w; w += align_up(((size_t)B * net->max_ch + 64) * sizeof(float), 256);
    net->gn_part = (double *)w;
    if (!dry) {
        // the convolutions' prologue reads a K chunk past [B, Ctot] without selects (channels beyond Cin carry zero weights,
        // but NaN * 0 is NaN).  For a layer with Ctot < max_ch that read-ahead lands INSIDE the arrays, at [B * Ctot,
        // B * Ctot + chunk), which no finalize of that layer writes: both arrays are zeroed whole, once per forward
        // ((B * max_ch + 64) floats each: a few KB)
        (void)hipMemsetAsync(net->gn_scale, 0, ((size_t)B * net->max_ch + 64) * sizeof(float), st);
        (void)hipMemsetAsync(net->gn_shift, 0, ((size_t)B * net->max_ch + 64) * sizeof(float), st);
    }
    net->ws = (char *)d_ws + fixed;
    net->arena.reset(dry ? (size_t)1 << 46 : ws_bytes - fixed);
    for (Tensor *tt : net->live) delete tt;
    net->live.clear();

    // ---- orientation.  The network is equivariant under a transposition of the image axes once every 3x3 kernel is
    // transposed too (GroupNorm, 1x1, attention and nearest resampling do not care), and the MFMA convolutions tile the
    // image in 32-column x 8-row blocks: for 2000x912 sinograms the heaviest level (128 channels at 500x228) pads 228
    // columns to 256 (12 %), while 228 ROWS pad to 232 and 500 columns to 512 (4 % in all).  So the forward may run on
    // transposed activations: x is transposed once on the way in, eps once on the way out (two 7 MB-per-slice passes
    // beside ~100 ms of convolutions); the choice depends on (H, W) only.
    net->transposed = choose_transposed(net, H, W);
    Fwd f{net};
    if (!dry) {
        f.rc = temb_launch(net->d_freqs, net->cfg.model_channels, t, net->te_w0, net->te_b0, net->te_w2, net->te_b2, emb_tmp, silu_emb, st);
        if (!f.rc && net->temb_rows)
            f.rc = gemv_bias_launch(net->temb_W, net->temb_b, net->conv1_b, silu_emb, net->bias_eff, net->temb_rows, net->topo.ted, st);
    }
    Tensor *x;
    if (net->transposed) {
        x = f.make(net->cfg.in_channels, W, H);
        if (!f.rc && !dry)
            hipLaunchKernelGGL(transpose_hw_kernel, dim3(cdiv(W, 32), cdiv(H, 32), B * net->cfg.in_channels), dim3(256), 0, st, d_x,
                               f.wptr(x), H, W);
        std::swap(H, W);
    } else {
        x = new Tensor();
        x->C = net->cfg.in_channels; x->H = H; x->W = W; x->external = true; x->ext = d_x; x->refs = 1;
        net->live.push_back(x);
    }

    std::vector<Tensor *> hs;
    Tensor *h = x;
    for (auto &blk : net->topo.down) {
        Tensor *o = f.run_block(blk, h, nullptr, 0, 0);
        f.release(h);              // drops the block input unless it is held as a skip
        h = o;
        f.retain(h);
        hs.push_back(h);
    }
    {
        Tensor *o = f.run_block(net->topo.middle, h, nullptr, 0, 0);
        f.release(h);
        h = o;
    }
    Tensor *h_ = hs.back(); hs.pop_back();     // refs held by hs transfer to h_
    for (auto &blk : net->topo.up) {
        Tensor *skip = h_;
        bool popped = false;
        if (!hs.empty()) { h_ = hs.back(); hs.pop_back(); popped = true; }   // quirk of Model/model.py:304-309
        Tensor *o = f.run_block(blk, h, skip, h_->H, h_->W);
        f.release(h);
        if (popped) f.release(skip);    // the consumed skip; when nothing was popped, `skip` is still h_ (reused next)
        h = o;
    }
    f.release(h_);
    // out = Conv(SiLU(GN(h)))  (Model/model.py:277-281,310)
    f.gn(h, nullptr, net->out_norm);
    Tensor *eps;
    if (net->transposed) {
        eps = f.conv(h, nullptr, net->out_conv, 1, 2, net->out_conv.b, nullptr, h->H, h->W);
        if (!f.rc && !dry)
            hipLaunchKernelGGL(transpose_hw_kernel, dim3(cdiv(W, 32), cdiv(H, 32), B * net->cfg.out_channels), dim3(256), 0, st,
                               f.ptr(eps), d_eps, H, W);       // (H, W are the transposed extents here)
    } else {
        eps = f.conv(h, nullptr, net->out_conv, 1, 2, net->out_conv.b, nullptr, h->H, h->W, dry ? reinterpret_cast<float *>((uintptr_t)256) : d_eps);
    }
    f.release(h);
    f.release(eps);
    if (need) *need = fixed + net->arena.high;
    for (Tensor *tt : net->live) delete tt;
    net->live.clear();
    return f.rc;
}

}  // namespace

extern "C" size_t ipdm_unet_workspace_bytes(ipdm_unet *net, int32_t B, int32_t H, int32_t W)
{
    if (!net || B <= 0 || H <= 0 || W <= 0) return 0;
    size_t need = 0;
    run_forward(net, nullptr, 0, nullptr, B, H, W, nullptr, 0, nullptr, true, &need);
    return need;
}

extern "C" int ipdm_unet_forward(ipdm_unet *net, const float *d_x, int32_t t, float *d_eps, int32_t B, int32_t H, int32_t W,
                                 void *d_ws, size_t ws_bytes, void *stream)
{
    IPDM_REQUIRE(net && d_x && d_eps && d_ws && B > 0 && H > 0 && W > 0 && t >= 0, "unet_forward: bad argument");
    return run_forward(net, d_x, t, d_eps, B, H, W, d_ws, ws_bytes, (hipStream_t)stream, false, nullptr);
}

// The same forward replayed from a captured hipGraph.  A forward allocates nothing and never synchronises (the workspace
// walk is replayed identically), so its ~400 launches can be recorded once per (t, B, H, W, buffers) and re-issued with one
// hipGraphLaunch: what a reverse loop with a static step list wants when the batch is small and the launches are short.
// The first call with a new key runs eagerly (it also sets the one-time kernel attributes), the second one captures.
extern "C" int ipdm_unet_forward_graph(ipdm_unet *net, const float *d_x, int32_t t, float *d_eps, int32_t B, int32_t H, int32_t W,
                                       void *d_ws, size_t ws_bytes, void *stream)
{
    IPDM_REQUIRE(net && d_x && d_eps && d_ws && B > 0 && H > 0 && W > 0 && t >= 0, "unet_forward_graph: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (prof_enabled() || !st) return run_forward(net, d_x, t, d_eps, B, H, W, d_ws, ws_bytes, st, false, nullptr);   // (the legacy default stream cannot capture)
    if (const int ch = opt_changed_since(net->opts); ch >= 0) {
        set_error("unet_forward_graph: option '%s' changed after ipdm_unet_create", opt_name(ch));
        return IPDM_ERR_INVALID;
    }
    // every per-call switch may change the recorded launches: the whole vector of their values is part of the key (a packed
    // bit field aliased values that did not fit their bits, and missed switches added later)
    const std::vector<int> mode = opt_per_call_values();
    const ipdm_unet::GraphKey key{t, B, H, W, d_x, d_eps, d_ws, mode};
    auto it = net->graphs.find(key);
    if (it != net->graphs.end()) {
        // the recorded launches address the workspace as it was walked at capture time: same pointer (key), and it must
        // still be as large as that walk needs
        IPDM_REQUIRE(ws_bytes >= net->graph_need[key], "unet_forward_graph: workspace smaller (%zu) than at capture (%zu)", ws_bytes,
                     net->graph_need[key]);
        IPDM_HIP_CHECK(hipGraphLaunch(it->second, st));
        return IPDM_OK;
    }
    if (net->graph_seen.size() >= 1024) net->graph_seen.clear();      // callers that pass fresh buffers every time never capture
    if (net->graph_seen[key]++ == 0) return run_forward(net, d_x, t, d_eps, B, H, W, d_ws, ws_bytes, st, false, nullptr);
    if (net->graphs.size() >= 256) {            // buffers keep changing under the caller: start over rather than grow
        for (auto &kv : net->graphs) (void)hipGraphExecDestroy(kv.second);
        net->graphs.clear();
        net->graph_seen.clear();
        net->graph_need.clear();
    }
    hipGraph_t graph = nullptr;
    IPDM_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = run_forward(net, d_x, t, d_eps, B, H, W, d_ws, ws_bytes, st, false, nullptr);
    const hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        if (!rc) { set_error("unet_forward_graph: capture failed: %s", hipGetErrorString(e)); return IPDM_ERR_HIP; }
        return rc;
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { set_error("unet_forward_graph: instantiate failed: %s", hipGetErrorString(ei)); return IPDM_ERR_HIP; }
    net->graphs[key] = exec;
    {
        size_t need = 0;
        run_forward(net, nullptr, 0, nullptr, B, H, W, nullptr, 0, nullptr, true, &need);
        net->graph_need[key] = need;
    }
    IPDM_HIP_CHECK(hipGraphLaunch(exec, st));
    return IPDM_OK;
}

// the Winograd-domain weights of a test / benchmark layer, when its shape can use them (device pointer in *out, else null)
static int upload_wino(const float *w_host, int Cout, int Cin, int ks, int stride, int interleave, float **out)
{
    *out = nullptr;
    if (!conv_wino_shape_ok(Cout, Cin, ks, stride, interleave)) return IPDM_OK;
    std::vector<float> u;
    conv_pack_weights_wino(w_host, Cout, Cin, u);
    IPDM_HIP_CHECK(hipMalloc((void **)out, u.size() * sizeof(float)));
    IPDM_HIP_CHECK(hipMemcpy(*out, u.data(), u.size() * sizeof(float), hipMemcpyHostToDevice));
    return IPDM_OK;
}

// ------------------------------------------------------------------------------------ op-level entry (tests)
extern "C" int ipdm_op_conv2d(const float *d_x1, int32_t C1, const float *d_x2, int32_t C2, int32_t B, int32_t Hs, int32_t Ws,
                              int32_t H, int32_t W, const float *w_host, const float *b_host, int32_t Cout, int32_t ksize,
                              int32_t stride, int32_t act, int32_t groups, const float *gamma_host, const float *beta_host,
                              const float *d_res, float *d_out, void *stream)
{
    IPDM_REQUIRE(d_x1 && w_host && d_out, "op_conv2d: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int Cin = C1 + C2;
    std::vector<float> packed;
    int cin_pad, cout_pad;
    const int interleave = conv_weight_interleave(Cout, ksize, stride);
    conv_pack_weights(w_host, Cout, Cin, ksize, interleave, packed, cin_pad, cout_pad);
    float *d_w = nullptr, *d_b = nullptr, *d_g = nullptr, *d_be = nullptr, *d_sc = nullptr, *d_sh = nullptr, *d_split = nullptr, *d_wino = nullptr;
    double *d_part = nullptr;
    if (int rcw = upload_wino(w_host, Cout, Cin, ksize, stride, interleave, &d_wino)) return rcw;
    IPDM_HIP_CHECK(hipMalloc((void **)&d_w, packed.size() * sizeof(float)));
    IPDM_HIP_CHECK(hipMemcpy(d_w, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    if (b_host) {
        IPDM_HIP_CHECK(hipMalloc((void **)&d_b, Cout * sizeof(float)));
        IPDM_HIP_CHECK(hipMemcpy(d_b, b_host, Cout * sizeof(float), hipMemcpyHostToDevice));
    }
    int rc = IPDM_OK;
    if (act) {
        IPDM_REQUIRE(gamma_host && beta_host && groups > 0, "op_conv2d: GN prologue needs gamma/beta/groups");
        IPDM_HIP_CHECK(hipMalloc((void **)&d_g, Cin * sizeof(float)));
        IPDM_HIP_CHECK(hipMalloc((void **)&d_be, Cin * sizeof(float)));
        IPDM_HIP_CHECK(hipMemcpy(d_g, gamma_host, Cin * sizeof(float), hipMemcpyHostToDevice));
        IPDM_HIP_CHECK(hipMemcpy(d_be, beta_host, Cin * sizeof(float), hipMemcpyHostToDevice));
        IPDM_HIP_CHECK(hipMalloc((void **)&d_sc, ((size_t)B * Cin + 64) * sizeof(float)));      // (+ a K chunk of read-ahead,
        IPDM_HIP_CHECK(hipMalloc((void **)&d_sh, ((size_t)B * Cin + 64) * sizeof(float)));      //  zeroed: NaN * 0 weight = NaN)
        IPDM_HIP_CHECK(hipMemsetAsync(d_sc + (size_t)B * Cin, 0, 64 * sizeof(float), st));
        IPDM_HIP_CHECK(hipMemsetAsync(d_sh + (size_t)B * Cin, 0, 64 * sizeof(float), st));
        IPDM_HIP_CHECK(hipMalloc((void **)&d_part, gn_partials_bytes(B, groups)));
        GnArgs g;
        g.x1 = d_x1; g.x2 = d_x2; g.C1 = C1; g.C2 = C2; g.B = B; g.HW = (long)Hs * Ws; g.groups = groups;
        g.gamma = d_g; g.beta = d_be; g.eps = 1e-5f; g.partials = d_part; g.scale = d_sc; g.shift = d_sh;
        rc = gn_stats_launch(g, st);
    }
    if (!rc) {
        const int pad = ksize / 2;
        ConvArgs a;
        a.x1 = d_x1; a.x2 = d_x2; a.C1 = C1; a.C2 = C2; a.B = B; a.Hs = Hs; a.Ws = Ws; a.H = H; a.W = W;
        a.upsample = (H != Hs || W != Ws);
        a.scale_y = (float)Hs / (float)H; a.scale_x = (float)Ws / (float)W;
        a.w = d_w; a.w_wino = d_wino; a.cout_pad = cout_pad; a.w_interleave = interleave; a.bias = d_b; a.Cout = Cout; a.ksize = ksize; a.stride = stride;
        a.Ho = (H + 2 * pad - ksize) / stride + 1; a.Wo = (W + 2 * pad - ksize) / stride + 1;
        a.act = act; a.gn_scale = d_sc; a.gn_shift = d_sh; a.res = d_res; a.out = d_out;
        a.tiles_x = a.tiles_y = a.co_tiles = 0;
        if (conv_split_ws_bytes(a)) IPDM_HIP_CHECK(hipMalloc((void **)&d_split, conv_split_ws_bytes(a)));
        a.split_ws = d_split;
        rc = conv2d_launch(a, st);
    }
    IPDM_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(d_split);
    (void)hipFree(d_w); (void)hipFree(d_b); (void)hipFree(d_g); (void)hipFree(d_be); (void)hipFree(d_sc); (void)hipFree(d_sh); (void)hipFree(d_wino);
    (void)hipFree(d_part);
    return rc;
}

// x -> convA (+bias, +residual) -> GroupNorm(+SiLU) from convA's FUSED per-tile statistics -> convB 3x3: the
// statistics hand-over between a producing convolution and the GroupNorm that follows it, as the executor wires it.
// Test entry: Upsample (nearest 2x + 3x3 conv, in its parity form when eligible) -> GroupNorm(+SiLU) over cat(mid, skip)
// -> conv B reading mid as stored (parity-planar after an up2 convolution).  d_mid receives mid as NCHW for checking.
extern "C" int ipdm_op_up_conv_chain(const float *d_x, int32_t C, int32_t B, int32_t Hs, int32_t Ws, const float *wA_host,
                                     const float *bA_host, int32_t CA, const float *d_skip, int32_t C2, int32_t groups,
                                     const float *gamma_host, const float *beta_host, int32_t act, const float *wB_host,
                                     const float *bB_host, int32_t CB, int32_t ksB, float *d_mid, float *d_out,
                                     int32_t *used_up2, void *stream)
{
    IPDM_REQUIRE(d_x && wA_host && wB_host && gamma_host && beta_host && d_mid && d_out && groups > 0 && (C2 == 0 || d_skip),
                 "op_up_conv_chain: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int H = 2 * Hs, W = 2 * Ws, Cc = CA + C2;
    std::vector<float> pA, pU, pB, pW;
    int cinp, coutpA, coutpB;
    const int ilA = conv_weight_interleave(CA, 3, 1), ilB = conv_weight_interleave(CB, ksB, 1);
    conv_pack_weights(wA_host, CA, C, 3, ilA, pA, cinp, coutpA);
    if (ilA == 2 || ilA == 4 || (ilA == 0 && CA <= 16)) conv_pack_weights_up2(wA_host, CA, C, ilA, pU);
    if (ilA && conv_wup2_shape_ok(CA, C)) conv_pack_weights_wup2(wA_host, CA, C, pW);
    conv_pack_weights(wB_host, CB, Cc, ksB, ilB, pB, cinp, coutpB);
    std::vector<void *> tofree;
    auto dev = [&](const void *h, size_t bytes, void **out) -> int {
        void *d = nullptr;
        IPDM_HIP_CHECK(hipMalloc(&d, bytes ? bytes : 4));
        if (h) IPDM_HIP_CHECK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
        else IPDM_HIP_CHECK(hipMemset(d, 0, bytes ? bytes : 4));      // (scale / shift read-ahead padding must not hold NaNs)
        tofree.push_back(d);
        *out = d;
        return IPDM_OK;
    };
    float *d_wW = nullptr;
    float *d_wA, *d_wU = nullptr, *d_wB, *d_bA = nullptr, *d_bB = nullptr, *d_g, *d_be, *d_sc, *d_sh, *d_stats = nullptr, *d_pl = nullptr, *d_lin = nullptr;
    double *d_part;
    int rc = dev(pA.data(), pA.size() * 4, (void **)&d_wA);
    if (!rc && !pU.empty()) rc = dev(pU.data(), pU.size() * 4, (void **)&d_wU);
    if (!rc && !pW.empty()) rc = dev(pW.data(), pW.size() * 4, (void **)&d_wW);
    if (!rc) rc = dev(pB.data(), pB.size() * 4, (void **)&d_wB);
    if (!rc && bA_host) rc = dev(bA_host, CA * 4, (void **)&d_bA);
    if (!rc && bB_host) rc = dev(bB_host, CB * 4, (void **)&d_bB);
    if (!rc) rc = dev(gamma_host, Cc * 4, (void **)&d_g);
    if (!rc) rc = dev(beta_host, Cc * 4, (void **)&d_be);
    if (!rc) rc = dev(nullptr, ((size_t)B * Cc + 64) * 4, (void **)&d_sc);
    if (!rc) rc = dev(nullptr, ((size_t)B * Cc + 64) * 4, (void **)&d_sh);
    if (!rc) rc = dev(nullptr, gn_partials_bytes(B, groups), (void **)&d_part);
    if (!rc) rc = dev(nullptr, (size_t)B * CA * H * W * 4, (void **)&d_pl);
    ConvArgs a;
    a.x1 = d_x; a.x2 = nullptr; a.C1 = C; a.C2 = 0; a.B = B; a.Hs = Hs; a.Ws = Ws; a.H = H; a.W = W; a.upsample = 1;
    a.scale_y = (float)Hs / (float)H; a.scale_x = (float)Ws / (float)W; a.w = d_wA; a.w_up2 = d_wU; a.w_wup2 = d_wW; a.cout_pad = coutpA; a.w_interleave = ilA;
    a.bias = d_bA; a.Cout = CA; a.ksize = 3; a.stride = 1; a.Ho = H; a.Wo = W; a.act = 0; a.gn_scale = a.gn_shift = nullptr; a.res = nullptr;
    a.out = d_pl; a.tiles_x = a.tiles_y = a.co_tiles = 0;
    const bool up2 = conv_up2_eligible(a);
    // 2: the direct kernel's parity form (NCHW output); 3: the F(2x2,2x2) form of the wide layers (conv_wup2.hip)
    if (used_up2) *used_up2 = up2 ? (conv_wup2_eligible(a) ? 3 : 1) : (conv_direct_up2_eligible(a) ? 2 : 0);
    const int rows = C2 == 0 ? conv_stats_rows(a) : 0;      // fused statistics when the GroupNorm covers mid alone
    if (!rc && rows > 0) {
        rc = dev(nullptr, (size_t)B * rows * CA * 2 * 4, (void **)&d_stats);
        if (!rc) IPDM_HIP_CHECK(hipMemsetAsync(d_stats, 0xff, (size_t)B * rows * CA * 2 * 4, st));     // NaN: unwritten rows show
        a.stats = d_stats; a.stats_rows = rows;
    }
    if (!rc) rc = conv2d_launch(a, st);
    if (!rc && up2) rc = planar_to_linear_launch(d_pl, d_mid, (long)B * CA, H, W, st);
    else if (!rc) IPDM_HIP_CHECK(hipMemcpyAsync(d_mid, d_pl, (size_t)B * CA * H * W * 4, hipMemcpyDeviceToDevice, st));
    if (!rc && rows > 0) {
        GnTileArgs g;
        g.nsrc = 1; g.src[0].stats = d_stats; g.src[0].rows = rows; g.src[0].C = CA; g.B = B; g.HW = (long)H * W; g.groups = groups;
        g.gamma = d_g; g.beta = d_be; g.eps = 1e-5f; g.partials = d_part; g.scale = d_sc; g.shift = d_sh;
        rc = gn_tiles_launch(g, st);
    } else if (!rc) {
        GnArgs g;       // (sums over a plane: the parity-planar order of mid does not matter)
        g.x1 = d_pl; g.x2 = d_skip; g.C1 = CA; g.C2 = C2; g.B = B; g.HW = (long)H * W; g.groups = groups; g.gamma = d_g; g.beta = d_be;
        g.eps = 1e-5f; g.partials = d_part; g.scale = d_sc; g.shift = d_sh;
        rc = gn_stats_launch(g, st);
    }
    if (!rc) {
        ConvArgs b;
        b.x1 = d_pl; b.x2 = d_skip; b.C1 = CA; b.C2 = C2; b.B = B; b.Hs = H; b.Ws = W; b.H = H; b.W = W; b.upsample = 0;
        b.scale_y = b.scale_x = 1.f; b.w = d_wB; b.cout_pad = coutpB; b.w_interleave = ilB; b.bias = d_bB; b.Cout = CB; b.ksize = ksB;
        b.stride = 1; b.Ho = H; b.Wo = W; b.act = act; b.gn_scale = d_sc; b.gn_shift = d_sh; b.res = nullptr; b.out = d_out;
        b.tiles_x = b.tiles_y = b.co_tiles = 0;
        float *d_wino = nullptr;
        if (!rc) rc = upload_wino(wB_host, CB, Cc, ksB, 1, ilB, &d_wino);
        if (d_wino) tofree.push_back(d_wino);
        b.w_wino = d_wino;
        if (up2 && !conv_planar_ok(b)) {      // a reader that takes NCHW only: convert, as the executor does
            rc = dev(nullptr, (size_t)B * CA * H * W * 4, (void **)&d_lin);
            if (!rc) rc = planar_to_linear_launch(d_pl, d_lin, (long)B * CA, H, W, st);
            b.x1 = d_lin;
        } else b.x1_planar = up2 ? 1 : 0;
        if (!rc && conv_split_ws_bytes(b)) { float *d_sp; rc = dev(nullptr, conv_split_ws_bytes(b), (void **)&d_sp); if (!rc) b.split_ws = d_sp; }
        if (!rc) rc = conv2d_launch(b, st);
    }
    (void)hipStreamSynchronize(st);
    for (void *d : tofree) (void)hipFree(d);
    return rc;
}

extern "C" int ipdm_op_conv_gn_conv(const float *d_x, int32_t C, int32_t B, int32_t H, int32_t W, const float *wA_host,
                                    const float *bA_host, int32_t CA, int32_t ksA, int32_t strideA, const float *d_resA,
                                    int32_t groups, const float *gamma_host, const float *beta_host, int32_t act,
                                    const float *wB_host, const float *bB_host, int32_t CB, float *d_mid, float *d_out,
                                    int32_t *fused_rows, void *stream)
{
    IPDM_REQUIRE(d_x && wA_host && wB_host && gamma_host && beta_host && d_mid && d_out && groups > 0, "op_conv_gn_conv: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int padA = ksA / 2;
    const int Hm = (H + 2 * padA - ksA) / strideA + 1, Wm = (W + 2 * padA - ksA) / strideA + 1;
    std::vector<float> pA, pB;
    int cinp, coutpA, coutpB;
    const int ilA = conv_weight_interleave(CA, ksA, strideA), ilB = conv_weight_interleave(CB, 3, 1);
    conv_pack_weights(wA_host, CA, C, ksA, ilA, pA, cinp, coutpA);
    conv_pack_weights(wB_host, CB, CA, 3, ilB, pB, cinp, coutpB);
    std::vector<void *> tofree;
    auto dev = [&](const void *h, size_t bytes, void **out) -> int {
        void *d = nullptr;
        IPDM_HIP_CHECK(hipMalloc(&d, bytes ? bytes : 4));
        if (h) IPDM_HIP_CHECK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
        else IPDM_HIP_CHECK(hipMemset(d, 0, bytes ? bytes : 4));      // (scale / shift read-ahead padding must not hold NaNs)
        tofree.push_back(d);
        *out = d;
        return IPDM_OK;
    };
    float *d_wA, *d_wB, *d_bA = nullptr, *d_bB = nullptr, *d_g, *d_be, *d_sc, *d_sh, *d_stats = nullptr;
    double *d_part;
    int rc = dev(pA.data(), pA.size() * 4, (void **)&d_wA);
    if (!rc) rc = dev(pB.data(), pB.size() * 4, (void **)&d_wB);
    if (!rc && bA_host) rc = dev(bA_host, CA * 4, (void **)&d_bA);
    if (!rc && bB_host) rc = dev(bB_host, CB * 4, (void **)&d_bB);
    if (!rc) rc = dev(gamma_host, CA * 4, (void **)&d_g);
    if (!rc) rc = dev(beta_host, CA * 4, (void **)&d_be);
    if (!rc) rc = dev(nullptr, ((size_t)B * CA + 64) * 4, (void **)&d_sc);
    if (!rc) rc = dev(nullptr, ((size_t)B * CA + 64) * 4, (void **)&d_sh);
    if (!rc) rc = dev(nullptr, gn_partials_bytes(B, groups), (void **)&d_part);
    ConvArgs a;
    a.x1 = d_x; a.x2 = nullptr; a.C1 = C; a.C2 = 0; a.B = B; a.Hs = H; a.Ws = W; a.H = H; a.W = W; a.upsample = 0;
    a.scale_y = a.scale_x = 1.f; a.w = d_wA; a.cout_pad = coutpA; a.w_interleave = ilA; a.bias = d_bA; a.Cout = CA; a.ksize = ksA;
    a.stride = strideA; a.Ho = Hm; a.Wo = Wm; a.act = 0; a.gn_scale = a.gn_shift = nullptr; a.res = d_resA; a.out = d_mid;
    a.tiles_x = a.tiles_y = a.co_tiles = 0;
    float *d_winoA = nullptr, *d_winoB = nullptr;
    if (!rc) rc = upload_wino(wA_host, CA, C, ksA, strideA, ilA, &d_winoA);
    if (d_winoA) tofree.push_back(d_winoA);
    if (!rc) rc = upload_wino(wB_host, CB, CA, 3, 1, ilB, &d_winoB);
    if (d_winoB) tofree.push_back(d_winoB);
    a.w_wino = d_winoA;
    if (!rc && conv_split_ws_bytes(a)) { float *d_sp; rc = dev(nullptr, conv_split_ws_bytes(a), (void **)&d_sp); if (!rc) a.split_ws = d_sp; }
    const int rows = conv_stats_rows(a);
    if (fused_rows) *fused_rows = rows;
    if (!rc && rows > 0) {
        rc = dev(nullptr, (size_t)B * rows * CA * 2 * 4, (void **)&d_stats);
        if (!rc) IPDM_HIP_CHECK(hipMemsetAsync(d_stats, 0xff, (size_t)B * rows * CA * 2 * 4, st));     // NaN: unwritten rows show
        a.stats = d_stats; a.stats_rows = rows;
    }
    if (!rc) rc = conv2d_launch(a, st);
    if (!rc && rows > 0) {
        GnTileArgs g;
        g.nsrc = 1; g.src[0].stats = d_stats; g.src[0].rows = rows; g.src[0].C = CA; g.B = B; g.HW = (long)Hm * Wm; g.groups = groups;
        g.gamma = d_g; g.beta = d_be; g.eps = 1e-5f; g.partials = d_part; g.scale = d_sc; g.shift = d_sh;
        rc = gn_tiles_launch(g, st);
    } else if (!rc) {
        GnArgs g;
        g.x1 = d_mid; g.x2 = nullptr; g.C1 = CA; g.C2 = 0; g.B = B; g.HW = (long)Hm * Wm; g.groups = groups; g.gamma = d_g; g.beta = d_be;
        g.eps = 1e-5f; g.partials = d_part; g.scale = d_sc; g.shift = d_sh;
        rc = gn_stats_launch(g, st);
    }
    if (!rc) {
        ConvArgs b;
        b.x1 = d_mid; b.x2 = nullptr; b.C1 = CA; b.C2 = 0; b.B = B; b.Hs = Hm; b.Ws = Wm; b.H = Hm; b.W = Wm; b.upsample = 0;
        b.scale_y = b.scale_x = 1.f; b.w = d_wB; b.cout_pad = coutpB; b.w_interleave = ilB; b.bias = d_bB; b.Cout = CB; b.ksize = 3;
        b.stride = 1; b.Ho = Hm; b.Wo = Wm; b.act = act; b.gn_scale = d_sc; b.gn_shift = d_sh; b.res = nullptr; b.out = d_out;
        b.tiles_x = b.tiles_y = b.co_tiles = 0;
        b.w_wino = d_winoB;
        rc = conv2d_launch(b, st);
    }
    (void)hipStreamSynchronize(st);
    for (void *d : tofree) (void)hipFree(d);
    return rc;
}

// ------------------------------------------------------------------------------------ micro-benchmark entry
// Times `iters` launches of one conv configuration on random data (kernel tuning; not on the product path).
extern "C" int ipdm_bench_conv2d(int32_t B, int32_t C1, int32_t C2, int32_t H, int32_t W, int32_t Cout, int32_t ksize,
                                 int32_t stride, int32_t act, int32_t with_res, int32_t iters, float *avg_ms)
{
    IPDM_REQUIRE(avg_ms && iters > 0, "bench_conv2d: bad argument");
    const bool x1_planar = (act & 256) != 0;          // tuning aid: time the kernel's parity-planar reader path (x1 as an up2 output)
    const bool up = (act & 512) != 0;                 // tuning aid: an Upsample layer (nearest 2x + 3x3) whose SOURCE is H x W
    act &= 255;
    IPDM_REQUIRE(!up || (ksize == 3 && stride == 1 && !C2 && !with_res && !act && !x1_planar), "bench_conv2d: bad Upsample configuration");
    const int Cin = C1 + C2, pad = ksize / 2;
    const int Hv = up ? 2 * H : H, Wv = up ? 2 * W : W;
    const int Ho = (Hv + 2 * pad - ksize) / stride + 1, Wo = (Wv + 2 * pad - ksize) / stride + 1;
    std::vector<float> w((size_t)Cout * Cin * ksize * ksize), packed;
    for (size_t i = 0; i < w.size(); ++i) w[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    int cin_pad, cout_pad;
    const int interleave = conv_weight_interleave(Cout, ksize, stride);
    conv_pack_weights(w.data(), Cout, Cin, ksize, interleave, packed, cin_pad, cout_pad);
    float *d_w, *d_x1, *d_x2 = nullptr, *d_out, *d_res = nullptr, *d_sc, *d_sh, *d_b, *d_wino = nullptr;
    if (int rcw = upload_wino(w.data(), Cout, Cin, ksize, stride, interleave, &d_wino)) return rcw;
    IPDM_HIP_CHECK(hipMalloc((void **)&d_w, packed.size() * 4));
    IPDM_HIP_CHECK(hipMemcpy(d_w, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
    IPDM_HIP_CHECK(hipMalloc((void **)&d_x1, (size_t)B * C1 * H * W * 4));
    ipdm_randn(d_x1, B, (int64_t)C1 * H * W, 1, 0, 0, nullptr);
    if (C2) { IPDM_HIP_CHECK(hipMalloc((void **)&d_x2, (size_t)B * C2 * H * W * 4)); ipdm_randn(d_x2, B, (int64_t)C2 * H * W, 2, 0, 0, nullptr); }
    IPDM_HIP_CHECK(hipMalloc((void **)&d_out, (size_t)B * Cout * Ho * Wo * 4));
    if (with_res) { IPDM_HIP_CHECK(hipMalloc((void **)&d_res, (size_t)B * Cout * Ho * Wo * 4)); ipdm_randn(d_res, B, (int64_t)Cout * Ho * Wo, 3, 0, 0, nullptr); }
    IPDM_HIP_CHECK(hipMalloc((void **)&d_sc, ((size_t)B * Cin + 64) * 4));
    IPDM_HIP_CHECK(hipMalloc((void **)&d_sh, ((size_t)B * Cin + 64) * 4));
    IPDM_HIP_CHECK(hipMalloc((void **)&d_b, (size_t)Cout * 4));
    IPDM_HIP_CHECK(hipMemset(d_sc + (size_t)B * Cin, 0, 64 * 4));
    IPDM_HIP_CHECK(hipMemset(d_sh + (size_t)B * Cin, 0, 64 * 4));
    ipdm_randn(d_sc, 1, (int64_t)B * Cin, 4, 0, 0, nullptr);
    ipdm_randn(d_sh, 1, (int64_t)B * Cin, 5, 0, 0, nullptr);
    ipdm_randn(d_b, 1, Cout, 6, 0, 0, nullptr);
    float *d_wU = nullptr, *d_wW = nullptr;
    if (up) {
        std::vector<float> pk;
        if (interleave == 2 || interleave == 4) {
            conv_pack_weights_up2(w.data(), Cout, Cin, interleave, pk);
            IPDM_HIP_CHECK(hipMalloc((void **)&d_wU, pk.size() * 4));
            IPDM_HIP_CHECK(hipMemcpy(d_wU, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
            if (conv_wup2_shape_ok(Cout, Cin)) {
                conv_pack_weights_wup2(w.data(), Cout, Cin, pk);
                IPDM_HIP_CHECK(hipMalloc((void **)&d_wW, pk.size() * 4));
                IPDM_HIP_CHECK(hipMemcpy(d_wW, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
            }
        }
    }
    ConvArgs a;
    a.x1 = d_x1; a.x2 = d_x2; a.C1 = C1; a.C2 = C2; a.B = B; a.Hs = H; a.Ws = W; a.H = Hv; a.W = Wv; a.upsample = up ? 1 : 0;
    a.w_up2 = d_wU; a.w_wup2 = d_wW;
    a.scale_y = a.scale_x = up ? 0.5f : 1.f; a.w = d_w; a.w_wino = d_wino; a.cout_pad = cout_pad; a.w_interleave = interleave; a.bias = d_b; a.Cout = Cout; a.ksize = ksize; a.stride = stride;
    a.Ho = Ho; a.Wo = Wo; a.act = act; a.gn_scale = d_sc; a.gn_shift = d_sh; a.res = d_res; a.out = d_out;
    a.tiles_x = a.tiles_y = a.co_tiles = 0;
    if (x1_planar) { IPDM_REQUIRE(conv_planar_ok(a), "bench_conv2d: this shape has no parity-planar reader"); a.x1_planar = 1; }
    float *d_split = nullptr, *d_stats = nullptr;
    if (conv_split_ws_bytes(a)) IPDM_HIP_CHECK(hipMalloc((void **)&d_split, conv_split_ws_bytes(a)));
    a.split_ws = d_split;
    if (up && conv_stats_rows(a) > 0) {       // (every Upsample of the networks feeds a GroupNorm: timed with its fused statistics)
        a.stats_rows = conv_stats_rows(a);
        IPDM_HIP_CHECK(hipMalloc((void **)&d_stats, (size_t)B * a.stats_rows * Cout * 2 * 4));
        a.stats = d_stats;
    }
    int rc = 0;
    const bool stamps = (opt(OPT_CONV_DBG) & 24) != 0;
    if (stamps) { IPDM_HIP_CHECK(hipMalloc((void **)&a.dbg_buf, 4096 * 8 * 8)); IPDM_HIP_CHECK(hipMemset(a.dbg_buf, 0, 4096 * 8 * 8)); }
    for (int i = 0; i < 3 && !rc; ++i) rc = conv2d_launch(a, nullptr);
    hipEvent_t e0, e1;
    IPDM_HIP_CHECK(hipEventCreate(&e0));
    IPDM_HIP_CHECK(hipEventCreate(&e1));
    IPDM_HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters && !rc; ++i) rc = conv2d_launch(a, nullptr);
    IPDM_HIP_CHECK(hipEventRecord(e1, nullptr));
    IPDM_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    IPDM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    if (stamps) {   // consumer wave 0 of every workgroup: cycles in MFMA section / epilogue / barrier wait / total (last launch)
        std::vector<unsigned long long> h(4096 * 8);
        IPDM_HIP_CHECK(hipMemcpy(h.data(), a.dbg_buf, h.size() * 8, hipMemcpyDeviceToHost));
        double s4[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int nz = 0;
        for (int g = 0; g < 4096; ++g) if (h[g * 8 + 3]) { for (int k = 0; k < 8; ++k) s4[k] += (double)h[g * 8 + k]; ++nz; }
        if (nz) fprintf(stderr, "  stamps over %d workgroups (s_memtime ticks, avg): consumer mfma %.0f epilogue %.0f barrier %.0f total %.0f | "
                        "producer issue %.0f wait %.0f math %.0f store %.0f\n", nz, s4[0] / nz, s4[1] / nz, s4[2] / nz, s4[3] / nz,
                        s4[4] / nz, s4[5] / nz, s4[6] / nz, s4[7] / nz);
        (void)hipFree(a.dbg_buf);
    }
    (void)hipFree(d_split);
    (void)hipFree(d_wino);
    (void)hipFree(d_wU); (void)hipFree(d_wW); (void)hipFree(d_stats);
    (void)hipFree(d_w); (void)hipFree(d_x1); (void)hipFree(d_x2); (void)hipFree(d_out); (void)hipFree(d_res); (void)hipFree(d_sc);
    (void)hipFree(d_sh); (void)hipFree(d_b); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

extern "C" int32_t ipdm_conv_layout_code(int32_t Cout, int32_t ksize, int32_t stride)
{
    return conv_weight_interleave(Cout, ksize, stride);
}

static int32_t conv_kernel_code_impl(int32_t B, int32_t Cout, int32_t Cin, int32_t ksize, int32_t stride, int32_t H, int32_t W, bool with_stats);
extern "C" int32_t ipdm_conv_kernel_code(int32_t B, int32_t Cout, int32_t Cin, int32_t ksize, int32_t stride, int32_t H, int32_t W)
{
    return conv_kernel_code_impl(B, Cout, Cin, ksize, stride, H, W, false);
}
// ... for a layer whose output feeds a GroupNorm (the executor asks it for fused statistics): the kernel rule of such a layer
// looks at the layer alone, never at the batch (conv_pw_stats_layer), so the answer can differ from the plain query's
extern "C" int32_t ipdm_conv_kernel_code_stats(int32_t B, int32_t Cout, int32_t Cin, int32_t ksize, int32_t stride, int32_t H, int32_t W)
{
    return conv_kernel_code_impl(B, Cout, Cin, ksize, stride, H, W, true);
}
static int32_t conv_kernel_code_impl(int32_t B, int32_t Cout, int32_t Cin, int32_t ksize, int32_t stride, int32_t H, int32_t W, bool with_stats)
{
    if (B <= 0 || Cout <= 0 || Cin <= 0 || H <= 0 || W <= 0 || (ksize != 1 && ksize != 3) || stride < 1 || stride > 2) return -1;
    static float dummy;                    // (only tested for null by the eligibility rules)
    const int pad = ksize / 2;
    ConvArgs a;
    a.x1 = &dummy; a.x2 = nullptr; a.C1 = Cin; a.C2 = 0; a.B = B; a.Hs = H; a.Ws = W; a.H = H; a.W = W; a.upsample = 0;
    a.scale_y = a.scale_x = 1.f; a.w = &dummy; a.bias = nullptr; a.Cout = Cout; a.ksize = ksize; a.stride = stride;
    a.w_interleave = conv_weight_interleave(Cout, ksize, stride);
    const int group = a.w_interleave ? 32 * a.w_interleave : 64;      // (conv_pack_weights)
    a.cout_pad = (Cout + group - 1) / group * group;
    a.Ho = (H + 2 * pad - ksize) / stride + 1; a.Wo = (W + 2 * pad - ksize) / stride + 1;
    a.act = 0; a.gn_scale = a.gn_shift = nullptr; a.res = nullptr; a.out = &dummy;
    a.tiles_x = a.tiles_y = a.co_tiles = 0;
    a.w_wino = conv_wino_shape_ok(Cout, Cin, ksize, stride, a.w_interleave) ? &dummy : nullptr;
    a.split_ws = &dummy;
    if (with_stats) { a.stats = &dummy; a.stats_rows = conv_stats_rows(a); }
    return conv_kernel_code(a);
}

extern "C" int ipdm_bench_attention(int32_t B, int32_t heads, int32_t d, int32_t T, int32_t iters, float *avg_ms)
{
    IPDM_REQUIRE(avg_ms && iters > 0, "bench_attention: bad argument");
    float *d_qkv, *d_out;
    IPDM_HIP_CHECK(hipMalloc((void **)&d_qkv, (size_t)B * heads * 3 * d * T * 4));
    IPDM_HIP_CHECK(hipMalloc((void **)&d_out, (size_t)B * heads * d * T * 4));
    ipdm_randn(d_qkv, B, (int64_t)heads * 3 * d * T, 9, 0, 0, nullptr);
    int rc = 0;
    float *d_scr = nullptr;
    if (attention_scratch_floats(B, heads, d, T)) IPDM_HIP_CHECK(hipMalloc((void **)&d_scr, attention_scratch_floats(B, heads, d, T) * 4));
    for (int i = 0; i < 2 && !rc; ++i) rc = attention_launch(d_qkv, d_out, B, heads, d, T, nullptr, d_scr);
    hipEvent_t e0, e1;
    IPDM_HIP_CHECK(hipEventCreate(&e0));
    IPDM_HIP_CHECK(hipEventCreate(&e1));
    IPDM_HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters && !rc; ++i) rc = attention_launch(d_qkv, d_out, B, heads, d, T, nullptr, d_scr);
    IPDM_HIP_CHECK(hipEventRecord(e1, nullptr));
    IPDM_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    IPDM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    (void)hipFree(d_qkv); (void)hipFree(d_out); (void)hipFree(d_scr); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

#if IPDM_UNET_TRACE == 2
extern "C" void ipdm_trace_dump(void)
{
    std::vector<unsigned long long> h(g_trace_desc.size());
    (void)hipDeviceSynchronize();
    if (!h.empty()) (void)hipMemcpy(h.data(), g_trace_slots, h.size() * 8, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h.size(); ++i) fprintf(stderr, "TRACE %4zu %s  %016llx\n", i, g_trace_desc[i].c_str(), h[i]);
    g_trace_desc.clear();
    if (g_trace_slots) (void)hipMemset(g_trace_slots, 0, 8192 * 8);
}
#endif

#if IPDM_UNET_TRACE == 3
// after the SECOND forward: the first convolution whose output differs between the two forwards, and where
extern "C" void ipdm_trace_dump(void)
{
    static size_t first_count = 0;
    (void)hipDeviceSynchronize();
    if (!first_count) { first_count = g_trace_rec.size(); return; }
    if (g_trace_rec.size() != 2 * first_count) { fprintf(stderr, "TRACE3 %zu records after the second forward, %zu after the first (pool %zu of %zu bytes)\n", g_trace_rec.size(), first_count, g_trace_used, g_trace_pool_bytes); return; }
    for (size_t i = 0; i < first_count; ++i) {
        const TraceRec &r1 = g_trace_rec[i], &r2 = g_trace_rec[first_count + i];
        std::vector<float> h1(r1.n), h2(r2.n);
        (void)hipMemcpy(h1.data(), g_trace_pool + r1.off, r1.n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h2.data(), g_trace_pool + r2.off, r2.n * 4, hipMemcpyDeviceToHost);
        if (!memcmp(h1.data(), h2.data(), r1.n * 4)) continue;
        size_t bad = 0;
        std::map<long, int> by_tile;      // (n, cout tile of 128, tile row of 4, tile column of 32) -> differing elements
        fprintf(stderr, "TRACE3 first differing convolution #%zu: %s\n", i, g_trace_desc[i].c_str());
        if (r1.C1 && getenv("IPDM_TRACE_DUMP_DIR")) {      // both outputs, the input, the GroupNorm table and the Winograd-domain weights: tools/experiments/analyze_trace3.py
            const std::string dir = getenv("IPDM_TRACE_DUMP_DIR");
            auto put = [&](const char *name, const void *host, size_t bytes) { FILE *f = fopen((dir + "/" + name).c_str(), "wb"); if (f) { fwrite(host, 1, bytes, f); fclose(f); } };
            put("out1.bin", h1.data(), r1.n * 4); put("out2.bin", h2.data(), r2.n * 4);
            std::vector<float> t((size_t)r1.B * r1.C1 * r1.H * r1.W);
            (void)hipMemcpy(t.data(), g_trace_pool + r1.off_x1, t.size() * 4, hipMemcpyDeviceToHost); put("x1.bin", t.data(), t.size() * 4);
            t.resize((size_t)r1.B * r1.C1);
            (void)hipMemcpy(t.data(), g_trace_pool + r1.off_sc, t.size() * 4, hipMemcpyDeviceToHost); put("sc.bin", t.data(), t.size() * 4);
            (void)hipMemcpy(t.data(), g_trace_pool + r1.off_sh, t.size() * 4, hipMemcpyDeviceToHost); put("sh.bin", t.data(), t.size() * 4);
            t.resize((size_t)(r1.C1 / 8) * (r1.C / 64) * 8192);
            (void)hipMemcpy(t.data(), r1.w, t.size() * 4, hipMemcpyDeviceToHost); put("u.bin", t.data(), t.size() * 4);
            char meta[128]; snprintf(meta, sizeof(meta), "%d %d %d %d %d %zu\n", r1.B, r1.C1, r1.C, r1.H, r1.W, i); put("meta.txt", meta, strlen(meta));
        }
        for (size_t j = 0; j < r1.n; ++j) {
            if (!memcmp(&h1[j], &h2[j], 4)) continue;
            const int x = (int)(j % r1.W), y = (int)(j / r1.W % r1.H), c = (int)(j / ((size_t)r1.W * r1.H) % r1.C), n = (int)(j / ((size_t)r1.W * r1.H * r1.C));
            if (bad < 8) fprintf(stderr, "TRACE3   n %d cout %3d y %3d x %3d   first %.9g   second %.9g\n", n, c, y, x, h1[j], h2[j]);
            by_tile[(((long)n * 64 + c / 128) * 4096 + y / 4) * 4096 + x / 32]++;
            ++bad;
        }
        {
            std::map<int, int> ys, cs, xs;
            for (size_t j = 0; j < r1.n; ++j) {
                if (!memcmp(&h1[j], &h2[j], 4)) continue;
                xs[(int)(j % r1.W)]++; ys[(int)(j / r1.W % r1.H)]++; cs[(int)(j / ((size_t)r1.W * r1.H) % r1.C)]++;
            }
            for (auto *m : {&ys, &cs, &xs}) {
                fprintf(stderr, "TRACE3   %s:", m == &ys ? "rows y" : m == &cs ? "couts" : "columns x");
                for (auto &kv : *m) fprintf(stderr, " %d(%d)", kv.first, kv.second);
                fprintf(stderr, "\n");
            }
        }
        fprintf(stderr, "TRACE3 %zu of %zu elements differ, in %zu (sample, cout tile, 4-row, 32-column) tiles:\n", bad, r1.n, by_tile.size());
        int shown = 0;
        for (auto &kv : by_tile) {
            if (shown++ >= 24) break;
            fprintf(stderr, "TRACE3   sample %ld cout tile %ld rows %ld.. columns %ld..: %d elements\n", kv.first / 4096 / 4096 / 64, kv.first / 4096 / 4096 % 64,
                    kv.first / 4096 % 4096 * 4, kv.first % 4096 * 32, kv.second);
        }
        return;
    }
    fprintf(stderr, "TRACE3 the two forwards agree in every convolution\n");
}
#endif
