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
//   [KC][taps][32*MB] are double-buffered in LDS.
// Pipeline per chunk: (1) issue the RAW global loads of chunk c+1 (branch-free: scalar channel base +
//   per-thread clamped spatial offset, all in flight together), (2) run the MFMAs of chunk c out of
//   LDS, (3) only then apply GroupNorm/SiLU (scale/shift are wave-uniform per channel -> scalar
//   operands) and write chunk c+1 to the other LDS buffer, (4) one barrier.  The global-load latency
//   is therefore hidden behind ~9k cycles of MFMA work.  All LDS operand reads are 32 consecutive
//   dwords per half-wave (conflict-free ds_read_b32).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KS, int STRIDE, int MB, int NB, int KC_>
struct ConvTile {
    static constexpr int KC = KC_;
    static constexpr int TAPS = KS * KS;
    static constexpr int TH = 4 * NB;                       // output rows per workgroup
    static constexpr int TW = 32;                           // output cols per workgroup
    static constexpr int IN_ROWS = (TH - 1) * STRIDE + KS;
    static constexpr int IN_COLS = (TW - 1) * STRIDE + KS;
    static constexpr int IN_CH = IN_ROWS * IN_COLS;         // floats per channel
    static constexpr int IN_TILE = KC * IN_CH;
    static constexpr int BN = 32 * MB;                      // couts per workgroup
    static constexpr int W_TILE = KC * TAPS * BN;
    static constexpr int SP = (IN_CH + 255) / 256;          // spatial positions per thread per channel
    static constexpr int W_VEC = (W_TILE / 4 + 255) / 256;  // float4 weight loads per thread per chunk
    static constexpr size_t LDS_BYTES = (size_t)2 * (IN_TILE + W_TILE) * sizeof(float);
};

// 32-lane sums (DPP row rotations + ds_swizzle across the two rows of a half): fused GroupNorm statistics of the output
#define IPDM_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))
__device__ inline float igemm_sum_half(float x)
{
    x += IPDM_DPP_F(x, 0x121);
    x += IPDM_DPP_F(x, 0x122);
    x += IPDM_DPP_F(x, 0x124);
    x += IPDM_DPP_F(x, 0x128);
    x += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));
    return x;
}
#undef IPDM_DPP_F

// SiLU with the hardware exp2/rcp (each ~1 ulp): |err| ~ 3e-7 relative, far inside the parity budget
__device__ inline float silu_fast(float v)
{
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * v);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int KS, int STRIDE, int MB, int NB, int KC>
__global__ void __launch_bounds__(256, 2) conv_igemm_kernel(ConvArgs a)
{
    using T = ConvTile<KS, STRIDE, MB, NB, KC>;
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
    const int co_t = wid % a.co_tiles;
    int rest = wid / a.co_tiles;
    const int tx = rest % a.tiles_x; rest /= a.tiles_x;
    const int ty = rest % a.tiles_y;
    const int n = rest / a.tiles_y;
    const int oy0 = ty * T::TH, ox0 = tx * T::TW, co0 = co_t * T::BN;
    const int iy0 = oy0 * STRIDE - KS / 2, ix0 = ox0 * STRIDE - KS / 2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 5, l31 = lane & 31;
    const int Ctot = a.C1 + a.C2;
    const int src_plane = a.Hs * a.Ws;

    // ---- per-thread spatial descriptors (the same for every channel): clamped source offset + validity
    int sp_off[T::SP];
    bool sp_ok[T::SP];
#pragma unroll
    for (int j = 0; j < T::SP; ++j) {
        const int sp = tid + j * 256;
        const int r = sp / T::IN_COLS, c = sp % T::IN_COLS;
        const int iy = iy0 + r, ix = ix0 + c;
        const bool ok = sp < T::IN_CH && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        int sy = min(max(iy, 0), a.H - 1), sx = min(max(ix, 0), a.W - 1);
        if (a.upsample) {   // F.interpolate(mode="nearest"): src = min(floor(dst * (in/out) in f32), in-1)
            sy = min((int)floorf((float)sy * a.scale_y), a.Hs - 1);
            sx = min((int)floorf((float)sx * a.scale_x), a.Ws - 1);
        }
        sp_off[j] = sy * a.Ws + sx;
        sp_ok[j] = ok;
    }

    float raw[KC][T::SP];
    f32x4 w_reg[T::W_VEC];

    // (1) raw loads of one chunk: channel base is wave-uniform (scalar), offsets are per-thread constants
    auto load_chunk = [&](int c0) __attribute__((always_inline)) {
        // the launcher guarantees a chunk never straddles the two sources (C1 % KC == 0 when C2 > 0)
        const bool from1 = c0 < a.C1;
        const float *base = from1 ? a.x1 + ((size_t)n * a.C1 + c0) * src_plane
                                  : a.x2 + ((size_t)n * a.C2 + (c0 - a.C1)) * src_plane;
        const int cend = (from1 ? a.C1 : Ctot) - c0;           // channels of this source left from c0
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const float *pc = base + (size_t)min(c, cend - 1) * src_plane;   // clamp: never reads past the tensor
#pragma unroll
            for (int j = 0; j < T::SP; ++j) raw[c][j] = pc[sp_off[j]];
        }
#pragma unroll
        for (int e = 0; e < T::W_VEC; ++e) {
            const int v4 = min(tid + e * 256, T::W_TILE / 4 - 1);      // clamped: unconditional load keeps w_reg in VGPRs
            const int row = v4 / (T::BN / 4), col4 = v4 % (T::BN / 4);
            w_reg[e] = *reinterpret_cast<const f32x4 *>(a.w + ((size_t)c0 * T::TAPS + row) * a.cout_pad + co0 + col4 * 4);
        }
    };
    // (3) GroupNorm/SiLU + zero padding, then LDS.  Split into KC*SP + W_VEC independent items so that the
    //     main loop can interleave them with the MFMAs of the current chunk (no VALU-only bubble per chunk).
    constexpr int N_IN_ITEMS = KC * T::SP;
    constexpr int N_ITEMS = N_IN_ITEMS + T::W_VEC;
    auto store_item = [&](int buf, int c0, int item) __attribute__((always_inline)) {
        float *ib = lds + buf * BUF;
        if (item < N_IN_ITEMS) {
            const int c = item / T::SP, j = item % T::SP;
            const int nvalid = Ctot - c0;                      // channels >= nvalid are zero padding
            const int sp = tid + j * 256;
            float v = raw[c][j];
            if (a.act) {                                        // wave-uniform channel -> scalar loads of scale/shift
                const int cc = min(c, nvalid - 1);
                v = v * a.gn_scale[(size_t)n * Ctot + c0 + cc] + a.gn_shift[(size_t)n * Ctot + c0 + cc];
                if (a.act == 2) v = silu_fast(v);
            }
            v = (sp_ok[j] && c < nvalid) ? v : 0.0f;
            if (T::IN_CH % 256 == 0 || sp < T::IN_CH) ib[c * T::IN_CH + sp] = v;
        } else {
            const int e = item - N_IN_ITEMS;
            const int v4 = tid + e * 256;
            if (v4 < T::W_TILE / 4) *reinterpret_cast<f32x4 *>(ib + T::IN_TILE + v4 * 4) = w_reg[e];
        }
    };
    auto store_chunk = [&](int buf, int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int item = 0; item < N_ITEMS; ++item) store_item(buf, c0, item);
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.0f;

    const int nchunks = (Ctot + KC - 1) / KC;
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        const bool more = ch + 1 < nchunks;
        if (more) load_chunk((ch + 1) * KC);
        const float *ib = lds + cur * BUF;
        const float *wb = ib + T::IN_TILE;
        const int kc_eff = min(KC, Ctot - ch * KC);
        const int npairs = (kc_eff + 1) >> 1;
        // Operand registers: B = the ROWS x KS input values of one channel pair this wave touches (loaded one
        // pair ahead), A = the MB weight values of one tap (loaded one tap ahead).  LDS latency is hidden behind
        // the 4 MFMAs (256 cycles) of the current tap; sched_barrier pins that order.
        constexpr int ROWS = (NB - 1) * STRIDE + KS;
        constexpr int NP = KC / 2;
        auto read_b = [&](int cp, float (&Bv)[ROWS][KS]) __attribute__((always_inline)) {
            const int c = cp * 2 + lk;
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
                    Bv[r][kx] = ib[c * T::IN_CH + (wave * NB * STRIDE + r) * T::IN_COLS + l31 * STRIDE + kx];
        };
        auto read_a = [&](int cp, int t, float (&A)[MB]) __attribute__((always_inline)) {
            const int c = cp * 2 + lk;
#pragma unroll
            for (int m = 0; m < MB; ++m) A[m] = wb[(c * T::TAPS + t) * T::BN + m * 32 + l31];
        };
        // one full chunk (KC channels), compile-time schedule; with_store: the staging items of chunk ch+1 are
        // spread between the MFMA groups from the second channel pair on (their loads were issued >2k cycles ago)
        auto full_chunk = [&](auto with_store) __attribute__((always_inline)) {
            constexpr bool WS = decltype(with_store)::value;
            constexpr int SLOTS = (NP > 1 ? NP - 1 : 1) * T::TAPS;
            constexpr int PER_SLOT = (N_ITEMS + SLOTS - 1) / SLOTS;
            float b_cur[ROWS][KS], b_nxt[ROWS][KS], a_c[MB], a_n[MB];
            read_b(0, b_cur);
            read_a(0, 0, a_c);
#pragma unroll
            for (int cp = 0; cp < NP; ++cp) {
                if (cp + 1 < NP) read_b(cp + 1, b_nxt);
#pragma unroll
                for (int t = 0; t < T::TAPS; ++t) {
                    if (t + 1 < T::TAPS) read_a(cp, t + 1, a_n);
                    else if (cp + 1 < NP) read_a(cp + 1, 0, a_n);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[m], b_cur[q * STRIDE + t / KS][t % KS], acc[m][q], 0, 0, 0);
                    if (WS) {
                        const int slot = (NP > 1 ? cp - 1 : cp) * T::TAPS + t;
                        if (slot >= 0) {
#pragma unroll
                            for (int k = 0; k < PER_SLOT; ++k)
                                if (slot * PER_SLOT + k < N_ITEMS) store_item(cur ^ 1, (ch + 1) * KC, slot * PER_SLOT + k);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MB; ++m) a_c[m] = a_n[m];
                }
                if (cp + 1 < NP) {
#pragma unroll
                    for (int r = 0; r < ROWS; ++r)
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) b_cur[r][kx] = b_nxt[r][kx];
                }
            }
        };
        bool stored = false;
        if (kc_eff == KC) {
            if (more) { full_chunk(std::true_type{}); stored = true; }
            else full_chunk(std::false_type{});
        } else {
            // partial chunk (Cin not a multiple of KC): plain schedule
            for (int cp = 0; cp < npairs; ++cp) {
                float b_cur[ROWS][KS], a_c[MB];
                read_b(cp, b_cur);
#pragma unroll
                for (int t = 0; t < T::TAPS; ++t) {
                    read_a(cp, t, a_c);
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[m], b_cur[q * STRIDE + t / KS][t % KS], acc[m][q], 0, 0, 0);
                }
            }
        }
        if (more && !stored) store_chunk(cur ^ 1, (ch + 1) * KC);
        __syncthreads();
    }

    // ---- epilogue: + bias (+ residual), coalesced NCHW stores (32 consecutive pixels per half-wave)
    const int ox = ox0 + l31;
    const size_t out_plane = (size_t)a.Ho * a.Wo;
    const bool full = (oy0 + T::TH <= a.Ho) && (ox0 + T::TW <= a.Wo) && (co0 + T::BN <= a.Cout);   // workgroup-uniform
    // fused GroupNorm statistics of the output (ConvArgs::stats): one row of per-cout partial sums per pixel row and
    // tile column, as conv_ws writes them; this kernel only serves a few narrow layers, so the plain form: a 32-lane sum
    // per accumulator register, one lane of each half stores its cout's pair
    const int tile_col = ox0 / T::TW;
    auto stat_row = [&](float v, bool ok, int oy, int co) __attribute__((always_inline)) {
        const float x = ok ? v : 0.0f;
        const float s1 = igemm_sum_half(x), s2 = igemm_sum_half(x * x);
        if (l31 == 0 && oy < a.Ho && co < a.Cout) {
            float *d = a.stats + (((size_t)n * a.stats_rows + (size_t)oy * a.tiles_x + tile_col) * a.Cout + co) * 2;
            d[0] = s1;
            d[1] = s2;
        }
    };
    if (full) {
        // fast path: no bounds checks, all residual loads of a 32x32 tile in flight together
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = a.bias ? a.bias[co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk] : 0.0f;
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int oy = oy0 + wave * NB + q;
                const size_t o0 = ((size_t)n * a.Cout + co0 + m * 32 + 4 * lk) * out_plane + (size_t)oy * a.Wo + ox;
                float rv[16];
                if (a.res) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = a.res[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * out_plane];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[m][q][r] + bv[r];
                    if (a.res) v += rv[r];
                    a.out[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * out_plane] = v;
                    if (a.stats) stat_row(v, true, oy, co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk);
                }
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int oy = oy0 + wave * NB + q;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    const bool ok = co < a.Cout && oy < a.Ho && ox < a.Wo;
                    float v = 0.0f;
                    if (ok) {
                        const size_t o = ((size_t)n * a.Cout + co) * out_plane + (size_t)oy * a.Wo + ox;
                        v = acc[m][q][r];
                        if (a.bias) v += a.bias[co];
                        if (a.res) v += a.res[o];
                        a.out[o] = v;
                    }
                    if (a.stats) stat_row(v, ok, oy, co);
                }
            }
    }
}

template <int KS, int STRIDE, int MB, int NB, int KC>
static int launch_conv(const ConvArgs &args, hipStream_t st)
{
    using T = ConvTile<KS, STRIDE, MB, NB, KC>;
    constexpr int prof_cls = (KS == 3 && STRIDE == 1 && MB == 2) ? 0 : 1;
    ConvArgs a = args;
    a.tiles_x = cdiv(a.Wo, T::TW);
    a.tiles_y = cdiv(a.Ho, T::TH);
    a.co_tiles = cdiv(a.Cout, T::BN);
    IPDM_REQUIRE(a.C2 == 0 || a.C1 % KC == 0, "conv2d: concat split %d not a multiple of the K chunk %d", a.C1, KC);
    IPDM_REQUIRE((long)a.C1 * a.Hs * a.Ws < (1L << 31) && (long)(a.C2 + 1) * a.Hs * a.Ws < (1L << 31),
                 "conv2d: per-sample tensor exceeds 32-bit offsets");
    const long nwg = (long)a.tiles_x * a.tiles_y * a.co_tiles * a.B;
    if (int rc = ensure_dynamic_lds((const void *)conv_igemm_kernel<KS, STRIDE, MB, NB, KC>, T::LDS_BYTES)) return rc;
    const bool prof = prof_enabled();
    if (prof) prof_before(prof_cls, st);
    hipLaunchKernelGGL((conv_igemm_kernel<KS, STRIDE, MB, NB, KC>), dim3((unsigned)nwg), dim3(256), T::LDS_BYTES, st, a);
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
    // wide 3x3 convolutions (>80 % of the path's FLOPs) run on the persistent wave-specialised kernel (conv_ws.hip);
    // when their weights were packed for it (conv_weight_interleave); IPDM_CONV_LEGACY=1 at pack time keeps them here
    if (a.w_interleave) return conv_pw_eligible(a) ? conv2d_pw_launch(a, st) : conv2d_ws_launch(a, st);      // (1x1: conv_pw.hip)
    if (!opt(OPT_CONV_NO_DIRECT) && conv_direct_eligible(a)) return conv2d_direct_launch(a, st);
    if (a.ksize == 3 && a.stride == 1) return wide ? launch_conv<3, 1, 2, 2, 8>(a, st) : launch_conv<3, 1, 1, 2, 8>(a, st);
    if (a.ksize == 3 && a.stride == 2) return wide ? launch_conv<3, 2, 2, 1, 8>(a, st) : launch_conv<3, 2, 1, 1, 8>(a, st);
    if (a.ksize == 1 && a.stride == 1) return wide ? launch_conv<1, 1, 2, 2, 8>(a, st) : launch_conv<1, 1, 1, 2, 8>(a, st);
    set_error("conv2d: unsupported ksize=%d stride=%d", a.ksize, a.stride);
    return IPDM_ERR_UNSUPPORTED;
}

// mirrors the dispatch of conv2d_launch: WHICH kernel this convolution runs on now (ipdm_conv_kernel_code)
int conv_kernel_code(const ConvArgs &a)
{
    if (a.w_interleave) {
        if (conv_pw_eligible(a)) return 10;
        if (conv_up2_eligible(a)) return conv_wup2_eligible(a) ? 11 : 7;
        if (conv_wino_eligible(a)) {
            if (a.split_ws && conv_split(a) > 1) return 9;
            if (conv_wino3_eligible(a)) return 12;
            return (!opt(OPT_WINO_V1) && conv_wino2_eligible(a)) ? 2 : 1;
        }
        return (a.split_ws && conv_ws_split(a) > 1) ? 4 : 3;
    }
    if (!opt(OPT_CONV_NO_DIRECT) && conv_direct_eligible(a)) return conv_nm_eligible(a) ? 6 : 5;
    return 8;
}

// mirrors the dispatch of conv2d_launch: rows of fused output statistics of the kernel this convolution runs on
int conv_stats_rows(const ConvArgs &a)
{
    if (a.w_interleave) return conv_pw_stats_layer(a) ? conv_pw_stats_rows(a) : conv_ws_stats_rows(a);      // (asked for layers that want statistics)
    if (!opt(OPT_CONV_NO_DIRECT) && conv_direct_eligible(a)) return conv_direct_stats_rows(a);
    return a.Ho * cdiv(a.Wo, 32);                                       // the 4-wave kernels below: a row per pixel row and tile column
}

int conv_split(const ConvArgs &a)
{
    if (!a.w_interleave || conv_pw_layer_ok(a)) return 1;      // (conv_pw's layers are never K-split ones)
    if (conv_wino_eligible(a)) return conv_ws_split(a) > 1 ? conv_wino_split(a) : 1;      // (K slices inside conv_wino2)
    return conv_ws_split(a);
}
size_t conv_split_ws_bytes(const ConvArgs &a)
{
    const int S = conv_split(a);
    return S > 1 ? (size_t)S * a.B * a.Cout * a.Ho * a.Wo * sizeof(float) : 0;
}

int conv_k_chunk() { return 8; }
int conv_ws_k_chunk(int ks, int interleave) { return (interleave && ks == 1) ? 32 : 8; }

// Up-sampling convolution (nearest 2x, then 3x3, zero padding 1): output pixel (2y + a, 2x + b) reads the source pixels
// (y + i + a - 1, x + j + b - 1), i, j in {0, 1}; the 3x3 taps that land on the same source pixel are
//   a = 0:  i = 0 <- ky {0},     i = 1 <- ky {1, 2}          a = 1:  i = 0 <- ky {0, 1},   i = 1 <- ky {2}
// (the same for b / kx).  Their sum (in double, rounded once) is the 2x2 weight of that parity.
void conv_pack_weights_up2(const float *w, int Cout, int Cin, int interleave, std::vector<float> &packed)
{
    packed.clear();
    std::vector<float> w2((size_t)Cout * Cin * 4), one;
    for (int par = 0; par < 4; ++par) {
        const int a = par >> 1, b = par & 1;
        for (size_t oc = 0; oc < (size_t)Cout * Cin; ++oc)
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) {
                    const int ky0 = a == 0 ? (i == 0 ? 0 : 1) : (i == 0 ? 0 : 2), ky1 = a == 0 ? (i == 0 ? 0 : 2) : (i == 0 ? 1 : 2);
                    const int kx0 = b == 0 ? (j == 0 ? 0 : 1) : (j == 0 ? 0 : 2), kx1 = b == 0 ? (j == 0 ? 0 : 2) : (j == 0 ? 1 : 2);
                    double acc = 0.0;
                    for (int ky = ky0; ky <= ky1; ++ky)
                        for (int kx = kx0; kx <= kx1; ++kx) acc += (double)w[oc * 9 + ky * 3 + kx];
                    w2[oc * 4 + i * 2 + j] = (float)acc;
                }
        int cin_pad, cout_pad;
        conv_pack_weights(w2.data(), Cout, Cin, 2, interleave, one, cin_pad, cout_pad);
        packed.insert(packed.end(), one.begin(), one.end());
    }
}

bool conv_planar_ok(const ConvArgs &a)
{
    if (a.upsample || (a.Hs & 1) || (a.Ws & 1)) return false;
    // (a layer of the pointwise kernel may still land on conv_ws.hip -- a low-fill launch without fused statistics -- so the
    //  producer's layout decision follows the STRICTER reader: conv_ws_planar_ok; conv_pw itself only needs even Ho / Wo)
    if (a.w_interleave) return conv_ws_planar_ok(a);                    // the wave-specialised kernels (conv_ws.hip)
    // (the stride-2 direct kernel and the 4-wave kernels below read NCHW only)
    return a.stride == 1 && !opt(OPT_CONV_NO_DIRECT) && !opt(OPT_DIRECT_NO_PLANAR) && conv_direct_eligible(a);
}

namespace {
__global__ void __launch_bounds__(256) planar_to_linear_kernel(const float *__restrict__ src, float *__restrict__ dst, long planes, int H, int W)
{
    const long total = planes * H * W;
    const int h2 = H >> 1, w2 = W >> 1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)(i / W % H);
        const long pl = i / ((long)H * W);
        dst[i] = src[pl * H * W + ((long)((y & 1) * 2 + (x & 1)) * h2 + (y >> 1)) * w2 + (x >> 1)];
    }
}
}  // namespace

int planar_to_linear_launch(const float *src, float *dst, long planes, int H, int W, hipStream_t st)
{
    IPDM_REQUIRE(!(H & 1) && !(W & 1), "planar_to_linear: odd size %dx%d", H, W);
    const long total = planes * H * W;
    int g = (int)((total + 1023) / 1024); if (g > 8192) g = 8192; if (g < 1) g = 1;
    hipLaunchKernelGGL(planar_to_linear_kernel, dim3(g), dim3(256), 0, st, src, dst, planes, H, W);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

int conv_weight_interleave(int Cout, int ks, int stride)
{
    const bool legacy = opt(OPT_CONV_LEGACY) != 0, legacy1 = opt(OPT_CONV1X1_LEGACY) != 0, legacy2 = opt(OPT_CONVS2_LEGACY) != 0;
    if (legacy || Cout <= 32 || (ks != 3 && ks != 1) || (ks == 1 && (legacy1 || stride != 1)) || stride > 2 || (stride == 2 && legacy2))
        return 0;
    return Cout > 96 ? 4 : 2;       // 128-cout tiles (MB=4,NB=2) for the wide layers, 64-cout x 16-row tiles (MB=2,NB=4) otherwise
}

// Repack reference-layout weights [Cout][Cin][k][k] (host) -> [Cin_pad][k*k][Cout_pad] (host), zero padded.
// interleave = MB > 0: inside every group of 32*MB couts the order is [l = co%32][m = co/32], so that the MB values one
// MFMA lane needs are adjacent in the LDS slab (one ds_read_b64/b128 per tap in conv_ws.hip).
void conv_pack_weights(const float *w, int Cout, int Cin, int ks, int interleave, std::vector<float> &packed, int &cin_pad,
                       int &cout_pad)
{
    const int group = interleave ? 32 * interleave : 64;
    const int kc = conv_ws_k_chunk(ks, interleave);           // channels per K chunk of the kernel that will read the slab
    cin_pad = (Cin + kc - 1) / kc * kc;
    cout_pad = (Cout + group - 1) / group * group;
    const int taps = ks * ks;
    packed.assign((size_t)cin_pad * taps * cout_pad, 0.0f);
    for (int co = 0; co < Cout; ++co) {
        int pos = co;
        if (interleave) {
            const int g = co / group, within = co % group;
            pos = g * group + (within % 32) * interleave + within / 32;
        }
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < taps; ++t)
                packed[((size_t)ci * taps + t) * cout_pad + pos] = w[((size_t)co * Cin + ci) * taps + t];
    }
}

}  // namespace ipdm
