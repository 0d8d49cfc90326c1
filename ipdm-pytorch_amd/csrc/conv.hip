// 2-D convolution (3x3 s1 / 3x3 s2 / 1x1) as an implicit GEMM on the exact-f32 MFMA of gfx950.
//
// Replaces the nn.Conv2d calls of the reference UNet (Model/model.py:101,113,117,142,143,165,180,
// 227,280) together with what surrounds them, fused:
//   prologue : GroupNorm(+SiLU) of the input applied while the tile is staged into LDS
//              (norm_layer + nn.SiLU in front of every conv, :99-100,111-112,141,278-279);
//              channel concat of two sources (torch.cat, :306) and nearest up-sampling to an
//              explicit size (F.interpolate, :168) are folded into the staging addresses;
//   epilogue : bias (which already carries the time-embedding projection, :128) and the residual
//              add (:130,155).
//
// GEMM view (per sample):  D[cout, pixel] = sum_{cin,ky,kx} W[cout,(cin,ky,kx)] * X[(cin,ky,kx), pixel]
//   A operand = weights (M = cout), B operand = input (N = pixel) so that one accumulator register
//   of v_mfma_f32_32x32x2_f32 holds 32 consecutive pixels of one cout row -> 128-B coalesced NCHW
//   stores.  The MFMA's two k-lanes are two consecutive input channels at the same tap.
//   Numerics: the f32 MFMA is an exact k-ordered fmaf chain (no reduced precision anywhere).
//
// Tiling: workgroup = 4 waves; tile = (4*NB rows) x 32 cols of output pixels x (32*MB) couts; wave w
//   owns rows [w*NB, (w+1)*NB) -> MB x NB accumulator tiles of 32x32.  K is walked in chunks of KC
//   input channels: the (halo'ed) input tile [KC][IN_ROWS][IN_COLS] and the weight slab
//   [KC][taps][32*MB] are double-buffered in LDS; chunk c+1 is prefetched into registers while
//   chunk c feeds the MFMAs.  All LDS operand reads are 32 consecutive dwords per half-wave
//   (conflict-free for ds_read_b32).
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KS, int STRIDE, int MB, int NB>
struct ConvTile {
    static constexpr int KC = 8;
    static constexpr int TAPS = KS * KS;
    static constexpr int TH = 4 * NB;                       // output rows per workgroup
    static constexpr int TW = 32;                           // output cols per workgroup
    static constexpr int IN_ROWS = (TH - 1) * STRIDE + KS;
    static constexpr int IN_COLS = (TW - 1) * STRIDE + KS;
    static constexpr int IN_CH = IN_ROWS * IN_COLS;         // floats per channel
    static constexpr int IN_TILE = KC * IN_CH;
    static constexpr int BN = 32 * MB;                      // couts per workgroup
    static constexpr int W_TILE = KC * TAPS * BN;
    static constexpr int IN_PER_THREAD = (IN_TILE + 255) / 256;
    static constexpr int W_VEC_PER_THREAD = (W_TILE / 4 + 255) / 256;
    static constexpr size_t LDS_BYTES = (size_t)2 * (IN_TILE + W_TILE) * sizeof(float);
};

__device__ inline float silu_f(float v) { return v / (1.0f + expf(-v)); }

template <int KS, int STRIDE, int MB, int NB>
__global__ void __launch_bounds__(256, 2) conv_igemm_kernel(ConvArgs a)
{
    using T = ConvTile<KS, STRIDE, MB, NB>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BUF = T::IN_TILE + T::W_TILE;   // floats per stage: [input tile | weight slab]

    // ---- work-item decode (XCD-aware: blocks sharing an XCD get a contiguous run of work items,
    //      and consecutive work items are the cout tiles of one pixel tile -> shared input in L2)
    const int nwg = gridDim.x;
    int wid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int co_tiles = a.co_tiles;
    const int co_t = wid % co_tiles;
    int rest = wid / co_tiles;
    const int tx = rest % a.tiles_x; rest /= a.tiles_x;
    const int ty = rest % a.tiles_y;
    const int n = rest / a.tiles_y;
    const int oy0 = ty * T::TH, ox0 = tx * T::TW, co0 = co_t * T::BN;
    const int iy0 = oy0 * STRIDE - KS / 2, ix0 = ox0 * STRIDE - KS / 2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ctot = a.C1 + a.C2;
    const size_t src_plane = (size_t)a.Hs * a.Ws;

    // ---- per-thread staging descriptors (independent of the channel chunk)
    int in_off[T::IN_PER_THREAD];      // source offset inside a channel plane, or -1 (zero padding)
#pragma unroll
    for (int e = 0; e < T::IN_PER_THREAD; ++e) {
        const int idx = tid + e * 256;
        const int sp = idx % T::IN_CH;
        const int r = sp / T::IN_COLS, c = sp % T::IN_COLS;
        const int iy = iy0 + r, ix = ix0 + c;
        int off = -1;
        if (idx < T::IN_TILE && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
            int sy = iy, sx = ix;
            if (a.upsample) {   // F.interpolate(mode="nearest"): src = min(floor(dst * (in/out) in f32), in-1)
                sy = min((int)floorf((float)iy * a.scale_y), a.Hs - 1);
                sx = min((int)floorf((float)ix * a.scale_x), a.Ws - 1);
            }
            off = sy * a.Ws + sx;
        }
        in_off[e] = off;
    }

    float in_reg[T::IN_PER_THREAD];
    float4 w_reg[T::W_VEC_PER_THREAD];

    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int e = 0; e < T::IN_PER_THREAD; ++e) {
            const int idx = tid + e * 256;
            const int c = c0 + idx / T::IN_CH;
            float v = 0.0f;
            if (in_off[e] >= 0 && c < Ctot) {
                const float *src = (c < a.C1) ? a.x1 + ((size_t)n * a.C1 + c) * src_plane
                                              : a.x2 + ((size_t)n * a.C2 + (c - a.C1)) * src_plane;
                v = src[in_off[e]];
                if (a.act) {
                    v = v * a.gn_scale[(size_t)n * Ctot + c] + a.gn_shift[(size_t)n * Ctot + c];
                    if (a.act == 2) v = silu_f(v);
                }
            }
            in_reg[e] = v;
        }
        // weights packed [Cin_pad][TAPS][Cout_pad] (Cout_pad multiple of 64, Cin_pad of KC): one KC chunk
        // of a BN-wide cout slab = KC*TAPS rows of BN floats
#pragma unroll
        for (int e = 0; e < T::W_VEC_PER_THREAD; ++e) {
            const int v4 = tid + e * 256;
            if (v4 < T::W_TILE / 4) {
                const int row = v4 / (T::BN / 4), col4 = v4 % (T::BN / 4);
                w_reg[e] = *reinterpret_cast<const float4 *>(a.w + ((size_t)c0 * T::TAPS + row) * a.cout_pad + co0 + col4 * 4);
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int e = 0; e < T::IN_PER_THREAD; ++e) {
            const int idx = tid + e * 256;
            if (idx < T::IN_TILE) lds[buf * BUF + idx] = in_reg[e];
        }
#pragma unroll
        for (int e = 0; e < T::W_VEC_PER_THREAD; ++e) {
            const int v4 = tid + e * 256;
            if (v4 < T::W_TILE / 4) *reinterpret_cast<float4 *>(lds + buf * BUF + T::IN_TILE + v4 * 4) = w_reg[e];
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.0f;

    const int nchunks = (Ctot + T::KC - 1) / T::KC;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    const int lk = lane >> 5, l31 = lane & 31;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nchunks) load_chunk((ch + 1) * T::KC);
        const float *ib = lds + cur * BUF;
        const float *wb = ib + T::IN_TILE;
        const int kc_eff = min(T::KC, Ctot - ch * T::KC);
        const int npairs = (kc_eff + 1) >> 1;
        for (int cp = 0; cp < npairs; ++cp) {
            const int c = cp * 2 + lk;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    float av[MB], bv[NB];
#pragma unroll
                    for (int m = 0; m < MB; ++m) av[m] = wb[(c * T::TAPS + ky * KS + kx) * T::BN + m * 32 + l31];
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        bv[q] = ib[c * T::IN_CH + ((wave * NB + q) * STRIDE + ky) * T::IN_COLS + l31 * STRIDE + kx];
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[q], acc[m][q], 0, 0, 0);
                }
        }
        if (ch + 1 < nchunks) store_chunk(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: + bias (+ residual), coalesced NCHW stores (32 consecutive pixels per half-wave)
    const int ox = ox0 + l31;
    const size_t out_plane = (size_t)a.Ho * a.Wo;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int oy = oy0 + wave * NB + q;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co < a.Cout && oy < a.Ho && ox < a.Wo) {
                    const size_t o = ((size_t)n * a.Cout + co) * out_plane + (size_t)oy * a.Wo + ox;
                    float v = acc[m][q][r];
                    if (a.bias) v += a.bias[co];
                    if (a.res) v += a.res[o];
                    a.out[o] = v;
                }
            }
        }
}

template <int KS, int STRIDE, int MB, int NB>
static int launch_conv(const ConvArgs &args, hipStream_t st)
{
    constexpr int prof_cls = (KS == 3 && STRIDE == 1 && MB == 2) ? 0 : 1;
    using T = ConvTile<KS, STRIDE, MB, NB>;
    ConvArgs a = args;
    a.tiles_x = cdiv(a.Wo, T::TW);
    a.tiles_y = cdiv(a.Ho, T::TH);
    a.co_tiles = cdiv(a.Cout, T::BN);
    const long nwg = (long)a.tiles_x * a.tiles_y * a.co_tiles * a.B;
    static bool attr_set = false;
    if (!attr_set) {
        IPDM_HIP_CHECK(hipFuncSetAttribute((const void *)conv_igemm_kernel<KS, STRIDE, MB, NB>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES));
        attr_set = true;
    }
    const bool prof = prof_enabled();
    if (prof) prof_before(prof_cls, st);
    hipLaunchKernelGGL((conv_igemm_kernel<KS, STRIDE, MB, NB>), dim3((unsigned)nwg), dim3(256), T::LDS_BYTES, st, a);
    if (prof) prof_after(prof_cls, 2.0 * a.B * a.Ho * a.Wo * (double)a.Cout * (a.C1 + a.C2) * KS * KS, st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

namespace ipdm {

int conv2d_launch(const ConvArgs &a, hipStream_t st)
{
    IPDM_REQUIRE(a.x1 && a.w && a.out && a.B > 0 && a.Cout > 0 && a.C1 > 0, "conv2d: bad argument");
    IPDM_REQUIRE(a.C2 == 0 || a.x2, "conv2d: second source missing");
    IPDM_REQUIRE(!a.act || (a.gn_scale && a.gn_shift), "conv2d: GN prologue without scale/shift");
    const bool wide = a.Cout > 32;
    if (a.ksize == 3 && a.stride == 1) return wide ? launch_conv<3, 1, 2, 2>(a, st) : launch_conv<3, 1, 1, 2>(a, st);
    if (a.ksize == 3 && a.stride == 2) return wide ? launch_conv<3, 2, 2, 1>(a, st) : launch_conv<3, 2, 1, 1>(a, st);
    if (a.ksize == 1 && a.stride == 1) return wide ? launch_conv<1, 1, 2, 2>(a, st) : launch_conv<1, 1, 1, 2>(a, st);
    set_error("conv2d: unsupported ksize=%d stride=%d", a.ksize, a.stride);
    return IPDM_ERR_UNSUPPORTED;
}

// Repack reference-layout weights [Cout][Cin][k][k] (host) -> [Cin_pad][k*k][Cout_pad] (host), zero padded.
void conv_pack_weights(const float *w, int Cout, int Cin, int ks, std::vector<float> &packed, int &cin_pad, int &cout_pad)
{
    cin_pad = (Cin + 7) / 8 * 8;
    cout_pad = (Cout + 63) / 64 * 64;
    const int taps = ks * ks;
    packed.assign((size_t)cin_pad * taps * cout_pad, 0.0f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < taps; ++t)
                packed[((size_t)ci * taps + t) * cout_pad + co] = w[((size_t)co * Cin + ci) * taps + t];
}

}  // namespace ipdm
