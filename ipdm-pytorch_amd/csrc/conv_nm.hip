// The narrow stride-1 convolutions of the projection-domain UNet (8 / 16 output channels at 2000x912 and 1000x456, 3x3 and
// 1x1, <= 32 input channels) on the 16-cout MFMA, v_mfma_f32_16x16x4_f32 (gfx950) -- round 3.
//
// conv_direct.hip evaluates these layers on the packed-f32 vector ALU.  That unit has the matrix pipe's f32 peak, but a
// 16->16 3x3 layer needs 4608 flops per pixel for 192 bytes: at 6 TB/s that is 144 TFLOP/s of a 157 TFLOP/s unit which also
// has to do the GroupNorm+SiLU prologue, the LDS traffic and the addressing -- the VALU kernel sits at 0.39 of the HBM peak
// (roofline_hbm of the bench line) and is VALU-bound (59 % of its issue slots are the FMAs).  On the MFMA the same flops
// cost a quarter of the issue slots and nothing else: M = the 16 couts of ONE instruction (8-cout layers use half of it and
// are still HBM-bound: 2304 executed flops per 96 bytes), N = 16 pixels, K = 4 input channels of one tap.
//
// MEASURED, and therefore OPT-IN (option conv_nm = 1: the 16-cout layers, 2: every eligible layer; default 0): in the
// micro-benchmark (tools/nm_check.py, B = 8, operands hot in the 256 MB memory-side cache) the 16-cout layers run 1.15-1.24x
// faster than on conv_direct.hip (16->16 @1000x456 +res 0.223 vs 0.275 ms) and the 8-cout layers 0.73-0.86x (half of M is
// padding: the same matrix-pipe cycles as a perfect v_pk_fma_f32 loop); inside the network, where those layers stream from
// HBM, the narrow family's total did not move (960.6 vs 961.5 ms per two bench steps with the 16-cout layers switched over),
// and a lone slice (B = 1) loses 20 % on them (504 wave-strips for 1024 SIMDs).  Kept as the measured alternative; parity
// covered by test_narrow_convolutions_on_the_16_cout_mfma_opt_in.
//
// No LDS staging of the input, no producer waves, no barriers: every wave is independent.
//   * a wave owns a 64 x 16 pixel strip (the statistics-row geometry of conv_direct.hip: one row of fused GroupNorm
//     partial sums per 64 x 16 tile, so the two kernels are interchangeable behind the executor);
//   * lane = (n = lane & 15, k = lane >> 4): pixels 4n .. 4n+3 of the strip row, channel 4g + k of channel group g.  One
//     16-byte buffer load per lane, row and group (16 lanes x 16 B = 256 contiguous bytes per channel row) plus one dword
//     load for the two halo columns of the strip (lanes 0 and 15 only); GroupNorm(+SiLU) ONCE per loaded element;
//   * the B operand of block j (pixels 4n + j, j = 0..3: the MFMA's N index is a labelling, stride-4 pixel sets are as
//     good as runs) at horizontal tap dx is REGISTER j + dx of {L, p0, p1, p2, p3, R}; L / R are the neighbouring lane's
//     p3 / p0 (one DPP row shift each, the halo dword at the row ends) -- no LDS, no shuffles per tap;
//   * vertical taps: input row r contributes to output rows r - ky through three rolling accumulator sets (row loop unrolled
//     by 3), an output row is finished two input rows later: + bias (+ residual, prefetched a row ahead) -> the D layout
//     (cout 4k + reg, pixel 4n + j) gives every lane 4 consecutive pixels of 4 couts: 16-byte stores, 256-byte segments;
//   * A operand (weights): [group][tap][lane] in LDS, one ds_read_b32 per 4 MFMAs (the 4 pixel blocks share it);
//   * fused GroupNorm statistics of the output: in-lane over the strip's 16 rows, one DPP reduction over n at the end,
//     stored straight from registers.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT_W = 64, NT_H = 16;                     // a wave's strip
constexpr int NOOB = 0x7fffffff;

inline double nm_bytes(const ConvArgs &a)
{
    return 4.0 * a.B * ((double)(a.C1 + a.C2) * a.Hs * a.Ws + (double)a.Cout * a.Ho * a.Wo * (a.res ? 2 : 1));
}

__device__ inline float nload(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ inline f32x4 nload4(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// G: channel groups of 4 (Ctot = 4 G); KS: 3 (padding 1) or 1.  Workgroup = 4 waves = 4 strips stacked vertically (64 x 64).
template <int G, int KS>
__global__ void __launch_bounds__(256, (G <= 3 ? 3 : 2)) conv_nm_kernel(ConvArgs a)
{
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    constexpr int ROWS = NT_H + 2 * PAD;                 // input rows of a strip
    __shared__ float w_lds[G * TAPS * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, k = lane >> 4;
    const int smp = blockIdx.z;
    const int x0 = blockIdx.x * NT_W, oy0 = (blockIdx.y * 4 + wave) * NT_H;
    const int Ctot = a.C1 + a.C2;
    const int plane = a.Hs * a.Ws, plane_bytes = plane * 4;

    // weights: A[m = cout][kk = channel in group] of (group g, tap t) sits in lane (m, kk) = (lane & 15, lane >> 4)
    // packed layout: [cin][tap][cout_pad] (plain), zero-padded couts
    for (int e = tid; e < G * TAPS * 64; e += 256) {
        const int l = e & 63, gt = e >> 6, g = gt / TAPS, t = gt - g * TAPS;
        w_lds[e] = a.w[((size_t)(4 * g + (l >> 4)) * TAPS + t) * a.cout_pad + (l & 15)];
    }
    __syncthreads();
    if (oy0 >= a.Ho) return;                             // (wave-uniform: the strips below the image; no barrier follows)

    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.x1 + (size_t)smp * a.C1 * plane), 0, a.C1 * plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.x2 ? a.x2 + (size_t)smp * a.C2 * plane : a.x1), 0,
                                                                         a.x2 ? a.C2 * plane_bytes : 0, 0x00020000);
    // per-lane offsets, fixed for the kernel: the lane's channel inside a group and its 4 pixels; the halo dword of the two
    // end lanes (left of pixel 0 / right of pixel 63; everything else out of range = no memory traffic)
    const bool has_left = x0 > 0, has_right = x0 + NT_W < a.W;                 // (uniform) the strip's halo columns exist
    const int voff = k * plane_bytes + n * 16;
    int voff_e = NOOB;
    if (KS == 3) {
        if (n == 0 && has_left) voff_e = k * plane_bytes;                      // byte (x0 - 1) through the scalar offset - 4
        if (n == 15 && has_right) voff_e = k * plane_bytes + (has_left ? 65 : 64) * 4;
    }
    // columns of the lane's run that lie inside the image (ragged right strip only)
    const bool edge_strip = x0 + NT_W > a.W;                                     // (uniform)
    bool cok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cok[j] = x0 + 4 * n + j < a.W;

    // GroupNorm scale / shift of the lane's channel per group
    float sc[G], sh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        sc[g] = a.act ? a.gn_scale[(size_t)smp * Ctot + 4 * g + k] : 1.0f;
        sh[g] = a.act ? a.gn_shift[(size_t)smp * Ctot + 4 * g + k] : 0.0f;
    }

    // input row r of the strip = image row oy0 - PAD + r; raw values of the NEXT row to be consumed, per group
    f32x4 xr[G];
    float xe[G];
    auto issue_row = [&](int g, int r) __attribute__((always_inline)) {
        const int y = oy0 - PAD + r;
        const bool rowok = y >= 0 && y < a.H && r < ROWS;                       // (uniform)
        const bool from1 = 4 * g < a.C1;                                         // (uniform; C1 % 4 == 0)
        const __amdgpu_buffer_rsrc_t rs = from1 ? rs1 : rs2;
        const int cb = (from1 ? 4 * g : 4 * g - a.C1) * plane_bytes;
        const int so = cb + (min(max(y, 0), a.H - 1) * a.Ws + x0) * 4;
        xr[g] = nload4(rs, rowok ? voff : NOOB, so);
        if (KS == 3) xe[g] = nload(rs, rowok ? voff_e : NOOB, has_left ? so - 4 : so);
    };
#pragma unroll
    for (int g = 0; g < G; ++g) issue_row(g, 0);

    // output side
    const int out_plane = a.Ho * a.Wo, plane4 = out_plane * 4;
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (size_t)smp * a.Cout * out_plane), 0, a.Cout * plane4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((a.res ? a.res : a.out) + (size_t)smp * a.Cout * out_plane), 0,
                                                                            a.res ? a.Cout * plane4 : 0, 0x00020000);
    const bool couts_ok = 4 * k < a.Cout;                                        // (Cout % 4 == 0) the lane's four couts exist
    const bool vec = (a.Wo & 3) == 0;                                            // (uniform) runs are whole and 16-byte aligned
    const bool run_ok = x0 + 4 * n + 3 < a.Wo;
    const int voff_o = (couts_ok && run_ok && vec) ? (4 * k * out_plane + 4 * n) * 4 : NOOB;
    const int voff_o1 = couts_ok ? (4 * k * out_plane + 4 * n) * 4 : NOOB;       // element-wise path (widths that are not multiples of 4)
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = (a.bias && couts_ok) ? a.bias[4 * k + r] : 0.0f;
    float st1[4] = {0, 0, 0, 0}, st2[4] = {0, 0, 0, 0};

    f32x4 acc[3][4];                                     // [output row mod 3][pixel block j]: D regs = couts 4k .. 4k+3
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[s][j] = f32x4{0, 0, 0, 0};
    f32x4 rv[4];                                         // residual of the output row that finishes with this input row

    auto finish_row = [&](int o, f32x4 (&ac)[4]) __attribute__((always_inline)) {     // output row o of the strip (0..15), uniform
        const int oy = oy0 + o;
        const bool rowok = oy < a.Ho;                                            // (uniform)
        const int so = (oy * a.Wo + x0) * 4;
        if (rowok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 v = {ac[0][r] + bias[r], ac[1][r] + bias[r], ac[2][r] + bias[r], ac[3][r] + bias[r]};
                if (vec) {
                    v += rv[r];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, voff_o, so + r * plane4, 0);
                    if (a.stats) {
                        if (edge_strip && voff_o == NOOB) v = f32x4{0, 0, 0, 0};
                        st1[r] += (v[0] + v[1]) + (v[2] + v[3]);
                        st2[r] += fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int vo = (x0 + 4 * n + j < a.Wo) ? voff_o1 + 4 * j : NOOB;
                        float e = v[j];
                        if (a.res) e += nload(r_rsrc, vo, so + r * plane4);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e), o_rsrc, vo, so + r * plane4, 0);
                        v[j] = vo != NOOB ? e : 0.0f;
                    }
                    if (a.stats) {
                        st1[r] += (v[0] + v[1]) + (v[2] + v[3]);
                        st2[r] += fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ac[j] = f32x4{0, 0, 0, 0};
    };

    // one input row: R3 = r mod 3 (compile time: the accumulator sets rotate)
    auto do_row = [&](auto r3_tag, int r) __attribute__((always_inline)) {
        constexpr int R3 = decltype(r3_tag)::value;
        const int y = oy0 - PAD + r;
        const bool rowok = y >= 0 && y < a.H;                                    // (uniform) a row of zero padding contributes nothing
        // the residual of the output row this input row completes (o = r - 2 PAD), ahead of the MFMAs
        const int o_done = r - 2 * PAD;
        if (o_done >= 0 && oy0 + o_done < a.Ho && vec) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rv[q] = nload4(r_rsrc, voff_o, ((oy0 + o_done) * a.Wo + x0) * 4 + q * plane4);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            // activate the landed row of group g, then refill its registers with the next row
            f32x4 p = xr[g];
            float e = KS == 3 ? xe[g] : 0.0f;
            if (r + 1 < ROWS) issue_row(g, r + 1);
            if (rowok) {
                if (a.act) {
                    p = p * sc[g] + sh[g];
                    e = fmaf(e, sc[g], sh[g]);
                    if (a.act == 2) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) p[j] = p[j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * p[j]));
                        e = e * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * e));
                    }
                }
                if (edge_strip) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) p[j] = cok[j] ? p[j] : 0.0f;
                }
                float X[6];
                X[1] = p[0]; X[2] = p[1]; X[3] = p[2]; X[4] = p[3];
                if (KS == 3) {
                    // L: the left neighbour's p3 (row_shr:1), lane 0 of the row keeps the halo dword (zero padding at the image
                    // edge: voff_e out of range there reads 0 -- but an activated 0 is not 0: masked); R likewise.  (Operands are
                    // the scalar copies: __builtin_bit_cast straight from a vector element yields element 0 with this compiler.)
                    const float el = (n == 0 && has_left) ? e : 0.0f, er = (n == 15 && has_right) ? e : 0.0f;
                    X[0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, el), __builtin_bit_cast(int, X[4]), 0x111, 0xf, 0xf, false));
                    X[5] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, er), __builtin_bit_cast(int, X[1]), 0x101, 0xf, 0xf, false));
                }
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    // input row r, vertical tap ky -> output row o = r - ky: skip rows outside the strip (uniform)
                    const int o = r - ky;
                    if (o < 0 || o >= NT_H) continue;
                    f32x4 (&ac)[4] = acc[(R3 + 3 - ky) % 3];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const float wv = w_lds[(g * TAPS + ky * KS + kx) * 64 + lane];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            ac[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, X[(KS == 3 ? j + kx : j + 1)], ac[j], 0, 0, 0);
                    }
                }
            }
        }
        if (o_done >= 0) finish_row(o_done, acc[(R3 + 3 - 2 * PAD) % 3]);
    };

    static_assert(ROWS % 3 == 0 || KS == 1, "conv_nm: the row loop is unrolled by 3");
    if (KS == 3) {
        for (int r = 0; r < ROWS; r += 3) {
            do_row(std::integral_constant<int, 0>{}, r);
            do_row(std::integral_constant<int, 1>{}, r + 1);
            do_row(std::integral_constant<int, 2>{}, r + 2);
        }
    } else {
        for (int r = 0; r < ROWS; ++r) do_row(std::integral_constant<int, 0>{}, r);
    }

    if (a.stats) {
        // over the 16 lanes n of the lane's DPP row (= one k: the same four couts)
#define IPDM_ROR(v, c) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (c), 0xf, 0xf, false))
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s1 = st1[r], s2 = st2[r];
            s1 += IPDM_ROR(s1, 0x121); s2 += IPDM_ROR(s2, 0x121);
            s1 += IPDM_ROR(s1, 0x122); s2 += IPDM_ROR(s2, 0x122);
            s1 += IPDM_ROR(s1, 0x124); s2 += IPDM_ROR(s2, 0x124);
            s1 += IPDM_ROR(s1, 0x128); s2 += IPDM_ROR(s2, 0x128);
            if (n == 0 && couts_ok) {
                const int row = (oy0 / NT_H) * gridDim.x + blockIdx.x;
                *reinterpret_cast<f32x2 *>(a.stats + (((size_t)smp * a.stats_rows + row) * a.Cout + 4 * k + r) * 2) = f32x2{s1, s2};
            }
        }
#undef IPDM_ROR
    }
}

template <int G, int KS>
int launch_nm(const ConvArgs &a, hipStream_t st)
{
    dim3 grid(cdiv(a.Wo, NT_W), cdiv(cdiv(a.Ho, NT_H), 4), a.B);
    const bool prof = prof_enabled();
    if (prof) prof_before(4, st);
    hipLaunchKernelGGL((conv_nm_kernel<G, KS>), grid, dim3(256), 0, st, a);
    if (prof) prof_after(4, nm_bytes(a), st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

template <int KS>
int launch_nm_g(const ConvArgs &a, hipStream_t st)
{
    switch ((a.C1 + a.C2) / 4) {
    case 1: return launch_nm<1, KS>(a, st);
    case 2: return launch_nm<2, KS>(a, st);
    case 3: return launch_nm<3, KS>(a, st);
    case 4: return launch_nm<4, KS>(a, st);
    case 6: return launch_nm<6, KS>(a, st);
    case 8: return launch_nm<8, KS>(a, st);
    }
    return IPDM_ERR_UNSUPPORTED;
}

}  // namespace

namespace ipdm {

// Which narrow layers run here: stride 1, 3x3 or 1x1, whole groups of 4 input channels (<= 32, the concat boundary on a
// group), 8..16 couts in whole groups of 4, NCHW sources.  A rule of the layer alone, never of the batch.
static bool conv_nm_eligible_impl(const ConvArgs &a)
{
    if (opt(OPT_CONV_NM) <= 0) return false;              // opt-in (see the header: no gain inside the network)
    const int Ctot = a.C1 + a.C2, g = Ctot / 4;
    if ((a.ksize != 3 && a.ksize != 1) || a.stride != 1 || a.upsample || a.x1_planar || a.H != a.Ho || a.W != a.Wo) return false;
    // (8-cout layers fill half of the instruction's M = 16 and measured slower than conv_direct.hip: only on request)
    if (a.w_interleave != 0 || a.cout_pad < 16 || a.Cout <= (opt(OPT_CONV_NM) >= 2 ? 4 : 8) || a.Cout > 16 || a.Cout % 4) return false;
    if (Ctot % 4 || (a.C2 && a.C1 % 4) || !(g == 1 || g == 2 || g == 3 || g == 4 || g == 6 || g == 8)) return false;
    return !conv_direct_up2_eligible(a);
}

static int conv2d_nm_launch_impl(const ConvArgs &a, hipStream_t st)
{
    IPDM_REQUIRE(conv_nm_eligible_impl(a), "conv2d_nm: layer not eligible");
    IPDM_REQUIRE(!a.stats || a.stats_rows == conv_direct_stats_rows(a), "conv2d_nm: statistics rows %d != %d", a.stats_rows, conv_direct_stats_rows(a));
    IPDM_REQUIRE((long)a.C1 * a.Hs * a.Ws < (1L << 29) && (long)(a.C2 + 1) * a.Hs * a.Ws < (1L << 29) && (long)a.Cout * a.Ho * a.Wo < (1L << 29),
                 "conv2d_nm: per-sample tensor exceeds the 2 GiB buffer-addressing range");
    return a.ksize == 3 ? launch_nm_g<3>(a, st) : launch_nm_g<1>(a, st);
}

}  // namespace ipdm

// entry points of libipdm_hip_optin.so (csrc/optin.hip)
// the copy of the product library this object's references resolved against (optin.hip compares it with its own)
extern "C" const void *ipdm_optin_bound_to(void) { return (const void *)&ipdm_last_error; }
// ... and the layout this object was COMPILED against: ABI version, sizeof(ConvArgs), number of option slots (a stale copy of
// this library next to a newer product would otherwise misread ConvArgs and the option table silently)
extern "C" void ipdm_optin_layout(int *out3) { out3[0] = IPDM_ABI_VERSION; out3[1] = (int)sizeof(ipdm::ConvArgs); out3[2] = (int)ipdm::OPT_COUNT; }
extern "C" int ipdm_optin_conv_nm_eligible(const ipdm::ConvArgs *a) { return ipdm::conv_nm_eligible_impl(*a) ? 1 : 0; }
extern "C" int ipdm_optin_conv2d_nm_launch(const ipdm::ConvArgs *a, hipStream_t st) { return ipdm::conv2d_nm_launch_impl(*a, st); }
