// Direct convolutions for the narrow layers of the projection-domain UNet (4/8/16 channels at 2000x912 and 1000x456, the
// stem, the eps output conv): 3x3 / 1x1 stride 1, the 3x3 stride-2 Downsample and the Upsample layer in its parity form,
// Cout <= 16 and Cin <= 160 (the 144->16 up-block conv included).
//
// These layers carry 2 % of the FLOPs but took 21 % of a proj forward on the 32-cout MFMA tiles (4-16x padding
// waste; profiles/r01c layer sweep).  Their arithmetic intensity (Cin*Cout*18 / ((Cin+Cout)*4) = 9-36 FLOP/B) is at
// or below the HBM ridge, and the packed-f32 VALU has the MFMA's f32 peak, so the unit is v_pk_fma_f32:
//   * workgroup = 256 threads = a 64 x 16 pixel tile; a thread owns 4 consecutive pixels of one row for ALL couts
//     (4*CO accumulators), so an input value is read from LDS once per thread and used 9*CO/ (3 overlap) times;
//   * the haloed input tile [4 or 8 ch][18][68] is staged through LDS (raw-buffer loads) with the same fused prologue as
//     the MFMA kernels (nearest up-sampling + concat as addressing, GroupNorm(+SiLU), zero padding after the activation;
//     a parity-planar x1 through a second offset set, template PLANAR);
//   * weights are wave-uniform: scalar loads, a cout pair is the SGPR operand of one v_pk_fma_f32 with the input value
//     broadcast (round 2; as LDS broadcast reads they made the CU's one LDS pipe the bound of the loop);
//   * epilogue: + bias (+ residual), 16-byte stores of the 4 pixels per cout, fused GroupNorm statistics of the output.
// One pass per tile, 3-6 workgroups per CU (__launch_bounds__'s second argument is waves per SIMD in HIP).  Two persistent
// forms were built and measured slower: producer / consumer wave specialisation (25-35 %), and a loop over (tile, chunk)
// steps that issues the loads of step s+1 before the FMAs of step s (8->8 @2000x912 0.53 vs 0.45 ms, 16->16 @1000x456
// 0.43 vs 0.30 ms: the 40 prefetched values cost a wave of occupancy per SIMD, and this VALU-bound loop lives on occupancy).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// algorithmic HBM bytes of one launch (bench.py's roofline_hbm: these layers are bandwidth-bound): every input, residual and
// output element once
inline double direct_bytes(const ConvArgs &a)
{
    return 4.0 * a.B * ((double)(a.C1 + a.C2 + (a.sk_w ? a.sk_C1 + a.sk_C2 : 0)) * a.Hs * a.Ws + (double)a.Cout * a.Ho * a.Wo * (a.res ? 2 : 1));
}

// The family holds two kinds of launches: the layers of the 4/8/16-channel levels (up to 32 input channels with a
// concatenated skip: 9-36 FLOP/B, BANDWIDTH-bound: class 4, recorded with their algorithmic bytes) and the readers of the
// 128-channel level -- 128 + 16 -> 16 at 1000x456, 65 FLOP/B against a ridge of 20 -- which are bound by the f32 VALU
// (class 6, recorded with their flops)
inline bool direct_compute_bound(const ConvArgs &a) { return a.C1 + a.C2 >= 64; }
inline double direct_flops(const ConvArgs &a, int taps)
{
    return 2.0 * a.B * a.Ho * a.Wo * a.Cout * ((double)(a.C1 + a.C2) * taps + (a.sk_w ? a.sk_C1 + a.sk_C2 : 0));
}
inline int direct_class(const ConvArgs &a) { return direct_compute_bound(a) ? 6 : 4; }
inline double direct_work(const ConvArgs &a, int taps) { return direct_compute_bound(a) ? direct_flops(a, taps) : direct_bytes(a); }

constexpr int DT_W = 64, DT_H = 16;

// Which tile a workgroup takes.  Workgroups are handed to the 8 XCDs round-robin in launch order (x fastest), so with
// tile = blockIdx the horizontal and vertical neighbours of every tile run on OTHER XCDs and each XCD's L2 fetches the
// halo columns and rows again: a 64-pixel row segment with its two halo pixels touches four 128-byte lines, two of them
// for one pixel each (rocprofv3 FETCH_SIZE of the 8 -> 8 layer: 1.55x its algorithmic reads).  Here XCD x takes the x-th
// contiguous eighth of the tile sequence in row-major order, so neighbouring tiles share an L2 and run close in time: fetch
// traffic -40 % (profiles/r03k_hbm_by_kernel.csv), kernel time -1.6 % -- the re-fetches had come from the memory-side cache
// and the kernel is issue-bound.  A bijection for any grid size.
struct DTile { int bx, by, n; };
__device__ inline DTile direct_tile()
{
    const int gx = gridDim.x, gy = gridDim.y;
    const int total = gx * gy * gridDim.z;
    const int L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int xcd = L & 7, idx = L >> 3, q = total >> 3, r = total & 7;
    const int t = xcd * q + min(xcd, r) + idx;
    DTile d;
    d.bx = t % gx;
    const int rest = t / gx;
    d.by = rest % gy;
    d.n = rest / gy;
    return d;
}
constexpr int DIN_P1 = 68;     // LDS row pitch at stride 1: 16-byte aligned runs of 4 (+ halo); stride 2: 132

// 16-lane sums: DPP row rotations; every lane of the row ends with the total
#define IPDM_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))
__device__ inline float sum_lanes_row(float x)           // over the 16 lanes of the lane's DPP row
{
    x += IPDM_DPP_F(x, 0x121);                           // row_ror:1
    x += IPDM_DPP_F(x, 0x122);                           // row_ror:2
    x += IPDM_DPP_F(x, 0x124);                           // row_ror:4
    x += IPDM_DPP_F(x, 0x128);                           // row_ror:8
    return x;
}
#undef IPDM_DPP_F

// out-of-range per-lane offsets of a raw buffer read 0 (idle staging slots)
constexpr int DOOB = 0x7fffffff;
__device__ inline float dload(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// ---- epilogue shared by the kernels of this file: + bias (+ residual); 4 consecutive pixels per cout; fused GroupNorm
// statistics of the output.  ty: the thread's row inside the 64x16 tile; in_lds: the (dead) staging area, >= 2*CO*256 floats
template <int CO>
__device__ inline void direct_epilogue(const ConvArgs &a, f32x2 (&acc)[4][CO / 2], int n, int ox0, int oy0, int tx, int ty, float *in_lds)
{
    const int tid = threadIdx.x;
    const int oy = oy0 + ty, ox = ox0 + tx * 4;
    const bool in_img = oy < a.Ho && ox < a.Wo;
    const size_t out_plane = (size_t)a.Ho * a.Wo;
    const bool vec = (a.Wo & 3) == 0;                      // ox is a multiple of 4: whole run inside, 16-byte aligned
    const bool full_tile = oy0 + DT_H <= a.Ho && ox0 + DT_W <= a.Wo;     // uniform: no pixel of the tile needs masking
    // Statistics: this loop is VALU-bound, so the 256-way sums go through LDS (its pipe is idle here) instead of DPP
    // butterflies: every thread parks its 2*CO in-lane sums in LDS ([value][thread], over the input staging area, which
    // is dead by now), 16 threads per value then add 16 entries each and one DPP row reduction finishes the job.
    float *st = in_lds;
    if (a.stats) __syncthreads();                          // every thread is done reading the last chunk's tile
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        if (co < a.Cout) {
            const float b = a.bias ? a.bias[co] : 0.0f;
            const size_t o = ((size_t)n * a.Cout + co) * out_plane + (size_t)oy * a.Wo + ox;
            f32x4 v = {acc[0][co / 2][co & 1] + b, acc[1][co / 2][co & 1] + b, acc[2][co / 2][co & 1] + b, acc[3][co / 2][co & 1] + b};
            if (in_img) {
                if (vec) {
                    if (a.res) v += *reinterpret_cast<const f32x4 *>(a.res + o);
                    *reinterpret_cast<f32x4 *>(a.out + o) = v;
                } else {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if (ox + p < a.Wo) {
                            v[p] += a.res ? a.res[o + p] : 0.0f;
                            a.out[o + p] = v[p];
                        }
                }
            }
            if (a.stats) {
                if (!full_tile) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) v[p] = (in_img && ox + p < a.Wo) ? v[p] : 0.0f;
                }
                st[(2 * co) * 256 + tid] = (v[0] + v[1]) + (v[2] + v[3]);
                st[(2 * co + 1) * 256 + tid] = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
            }
        }
    }
    if (a.stats) {
        __syncthreads();
        const int j = tid & 15;                            // 16 threads per value (cout k/2, sum or sum of squares)
#pragma unroll
        for (int k = tid >> 4; k < 2 * CO; k += 16) {
            if ((k >> 1) < a.Cout) {
                const f32x4 *src = reinterpret_cast<const f32x4 *>(st + k * 256 + j * 16);
                const f32x4 p0 = src[0], p1 = src[1], p2 = src[2], p3 = src[3];
                float s = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) +
                          (((p2[0] + p2[1]) + (p2[2] + p2[3])) + ((p3[0] + p3[1]) + (p3[2] + p3[3])));
                s = sum_lanes_row(s);
                if (j == 0) {
                    const int row = (oy0 / DT_H) * gridDim.x + ox0 / DT_W;
                    a.stats[(((size_t)n * a.stats_rows + row) * a.Cout + (k >> 1)) * 2 + (k & 1)] = s;
                }
            }
        }
    }
}

// DKC: input channels staged per pass.  4 for CO <= 8 (20 KB of LDS and ~90 VGPRs: 5 workgroups per CU instead of 3;
// 8->8 @2000x912 0.36 vs 0.45 ms), 8 for CO = 16 (whose 64 accumulators bound the occupancy anyway: 4 was 8 % slower).
// (the second launch bound keeps the register budget of the main loop -- 5 workgroups per CU for CO = 8, 6 for CO = 4 --
// when the statistics epilogue is compiled in: this loop lives on occupancy)
// PLANAR: x1 is stored parity-planar (ConvArgs::x1_planar: the output of an up-sampling convolution in its parity form)
// STRIDE 2 (Downsample of the narrow levels, 3x3): the same 64x16 OUTPUT tile over a 33 x 129 input window, 2 channels per pass
// SKIP (1: NCHW, 2: its x1 parity-planar): the ResidualBlock's 1x1 shortcut over the block input (Model/model.py:116-130,
// ConvArgs::sk_*) as EXTRA K chunks of this (the block's second) 3x3 convolution -- centre tap only, no prologue, into the same
// accumulators.  The shortcut's launch, the write of its output and the read of that output as this layer's residual
// disappear, and with them one staging pass over the block input's tile.
template <int CO, int KS, int DKC, bool PLANAR, int STRIDE = 1, int SKIP = 0>
__global__ void __launch_bounds__(256, (STRIDE == 2 ? (CO <= 8 ? 4 : 3) : (CO <= 4 ? 6 : (CO <= 8 ? 5 : 3)))) conv_direct_kernel(ConvArgs a)
{
    static_assert(SKIP == 0 || (KS == 3 && STRIDE == 1 && !PLANAR), "conv_direct: the fused shortcut rides on a 3x3 stride-1 NCHW layer");
    constexpr int DIN_H = (DT_H - 1) * STRIDE + KS, DIN_W = (DT_W - 1) * STRIDE + KS, DIN_P = STRIDE == 1 ? DIN_P1 : 132;
    constexpr int DIN_CH = DIN_H * DIN_P, TAPS = KS * KS, PAD = KS / 2;
    static_assert(STRIDE == 1 || (KS == 3 && !PLANAR), "conv_direct: stride 2 is the 3x3 Downsample");
    __shared__ __attribute__((aligned(16))) float in_lds[DKC * DIN_CH];
    // weights: wave-uniform addresses in the constant address space -> scalar loads; a cout pair is the SGPR operand of
    // one v_pk_fma_f32 (as LDS broadcast reads they cost 9*CO/4 ds_read_b128 per channel and wave, and the LDS pipe --
    // one per CU, shared by the 4 SIMDs -- was the bound of this loop, not the VALU)
    typedef const __attribute__((address_space(4))) float cfloat;
    cfloat *wk = (cfloat *)(unsigned long long)a.w;
    cfloat *gsc = (cfloat *)(unsigned long long)a.gn_scale, *gsh = (cfloat *)(unsigned long long)a.gn_shift;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const DTile dt = direct_tile();
    const int n = dt.n;
    const int ox0 = dt.bx * DT_W, oy0 = dt.by * DT_H;
    const int Ctot = a.C1 + a.C2;
    const int src_plane = a.Hs * a.Ws;

    f32x2 acc[4][CO / 2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < CO / 2; ++c) acc[p][c] = f32x2{0.0f, 0.0f};

    // staging descriptors: element e of the tile = (row r, col c) for every channel of a chunk
    constexpr int NSP = (DIN_H * DIN_W + 255) / 256;
    int sp_src[NSP], sp_srcp[NSP], sp_dst[NSP];      // sp_srcp: the same element inside a parity-planar x1
    bool sp_ok[NSP];
#pragma unroll
    for (int j = 0; j < NSP; ++j) {
        const int e = tid + j * 256;
        const int r = e / DIN_W, c = e % DIN_W;
        const int iy = oy0 * STRIDE - PAD + r, ix = ox0 * STRIDE - PAD + c;
        sp_ok[j] = e < DIN_H * DIN_W && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        int sy = min(max(iy, 0), a.H - 1), sx = min(max(ix, 0), a.W - 1);
        if (a.upsample) {   // F.interpolate(mode="nearest"): src = min(floor(dst * (in/out) in f32), in-1)
            sy = min((int)floorf((float)sy * a.scale_y), a.Hs - 1);
            sx = min((int)floorf((float)sx * a.scale_x), a.Ws - 1);
        }
        // byte offsets inside a channel plane (buffer loads: no 64-bit address arithmetic on the VALU, which bounds this kernel)
        sp_src[j] = e < DIN_H * DIN_W ? (sy * a.Ws + sx) * 4 : DOOB;
        sp_srcp[j] = !(PLANAR || SKIP == 2) ? sp_src[j]
                   : (e < DIN_H * DIN_W ? ((((sy & 1) * 2 + (sx & 1)) * (a.Hs >> 1) + (sy >> 1)) * (a.Ws >> 1) + (sx >> 1)) * 4 : DOOB);
        sp_dst[j] = e < DIN_H * DIN_W ? r * DIN_P + c : -1;
    }
    const int plane_bytes = src_plane * 4;
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.x1 + (size_t)n * a.C1 * src_plane), 0, a.C1 * plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.x2 ? a.x2 + (size_t)n * a.C2 * src_plane : a.x1), 0, a.x2 ? a.C2 * plane_bytes : 0, 0x00020000);

    // one pass over the channels of a source pair: the layer's own input (all taps, fused prologue), or the block input of
    // the fused shortcut (SK: centre tap, raw values)
    auto run_chunks = [&](auto sk_tag) __attribute__((always_inline)) {
    constexpr bool SK = decltype(sk_tag)::value;
    const int C1v = SK ? a.sk_C1 : a.C1, Ctotv = SK ? a.sk_C1 + a.sk_C2 : Ctot;
    const int actv = SK ? 0 : a.act;
    const __amdgpu_buffer_rsrc_t rsA = SK ? __builtin_amdgcn_make_buffer_rsrc((void *)(a.sk_x1 + (size_t)n * a.sk_C1 * src_plane), 0, a.sk_C1 * plane_bytes, 0x00020000) : rsrc1;
    const __amdgpu_buffer_rsrc_t rsB = SK ? __builtin_amdgcn_make_buffer_rsrc((void *)(a.sk_x2 ? a.sk_x2 + (size_t)n * a.sk_C2 * src_plane : a.sk_x1), 0,
                                                                               a.sk_x2 ? a.sk_C2 * plane_bytes : 0, 0x00020000) : rsrc2;
    constexpr bool PL = SK ? SKIP == 2 : PLANAR;
    cfloat *wkv = SK ? (cfloat *)(unsigned long long)a.sk_w : wk;
    const int cpadv = SK ? a.sk_cout_pad : a.cout_pad;
    for (int c0 = 0; c0 < Ctotv; c0 += DKC) {
        const int kc = min(DKC, Ctotv - c0);
        __syncthreads();                                   // previous chunk fully consumed
        // all global loads of the chunk first (one latency), then the transform
        float raw[DKC][NSP];
#pragma unroll
        for (int c = 0; c < DKC; ++c) {
            const int cg = min(c0 + c, Ctotv - 1);         // channels beyond Cin re-read the last one; they are not consumed
            // (no branch around the loads: all loads of the chunk must issue back to back; the source is picked by scalar
            //  selects, and only a parity-planar x1 costs a per-load select of the offset)
            const bool from1 = cg < C1v;
            const __amdgpu_buffer_rsrc_t r = from1 ? rsA : rsB;
            const int so = (from1 ? cg : cg - C1v) * plane_bytes;
#pragma unroll
            for (int j = 0; j < NSP; ++j) raw[c][j] = dload(r, PL && from1 ? sp_srcp[j] : sp_src[j], so);
        }
#pragma unroll
        for (int c = 0; c < DKC; ++c) {
            if (c < kc) {
                float sc = 1.0f, sh = 0.0f;
                // (constant address space: scalar loads.  As vector loads they sit behind the chunk's 40 tile loads in the
                //  in-order memory counter and every channel's transform waits for all of them)
                if (actv) { sc = gsc[(size_t)n * Ctot + c0 + c]; sh = gsh[(size_t)n * Ctot + c0 + c]; }
#pragma unroll
                for (int j = 0; j < NSP; ++j) {
                    float v = raw[c][j];
                    if (actv) {
                        v = v * sc + sh;
                        // SiLU.  (A packed two-element form of this transform measured 1 % slower per forward; separate code
                        // paths per prologue with the exponent's argument as its own multiply-add -- 6 instead of 8 VALU
                        // per value -- gained 1-4 % on most layers but cost the 18-chunk parity-planar reader 65 %.)
                        if (actv == 2) v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
                    }
                    if (sp_dst[j] >= 0) in_lds[c * DIN_CH + sp_dst[j]] = sp_ok[j] ? v : 0.0f;
                }
            }
        }
        __syncthreads();
        for (int c = 0; c < kc; ++c) {
            const float *ip = in_lds + c * DIN_CH + ty * STRIDE * DIN_P + tx * 4 * STRIDE;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                if (SK && ky != 1) continue;
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(ip + ky * DIN_P);
                f32x2 hi = {0.0f, 0.0f};
                if (KS > 1 && STRIDE == 1) hi = *reinterpret_cast<const f32x2 *>(ip + ky * DIN_P + 4);
                f32x4 mid = {0.0f, 0.0f, 0.0f, 0.0f};
                float last = 0.0f;
                if (STRIDE == 2) { mid = *reinterpret_cast<const f32x4 *>(ip + ky * DIN_P + 4); last = ip[ky * DIN_P + 8]; }
                // stride 1: the 6 columns x .. x+5 of the 4 outputs' windows; stride 2: the 9 columns 2x .. 2x+8
                const float iv[9] = {lo[0], lo[1], lo[2], lo[3], STRIDE == 1 ? hi[0] : mid[0], STRIDE == 1 ? hi[1] : mid[1], mid[2], mid[3], last};
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    if (SK && kx != 1) continue;
                    // packed [Cin_pad8][taps][cout_pad] (plain layout), cout_pad >= CO; the shortcut's: [Cin_pad8][1][cout_pad]
                    cfloat *wp = wkv + (SK ? (size_t)(c0 + c) : (size_t)(c0 + c) * TAPS + ky * KS + kx) * cpadv;
                    f32x2 wv[CO / 2];
#pragma unroll
                    for (int q = 0; q < CO / 2; ++q) wv[q] = f32x2{wp[2 * q], wp[2 * q + 1]};
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int q = 0; q < CO / 2; ++q) acc[p][q] += wv[q] * iv[p * STRIDE + kx];
                }
            }
        }
    }

    };
    run_chunks(std::false_type{});
    if (SKIP) run_chunks(std::true_type{});

    static_assert(2 * CO * 256 <= DKC * DIN_CH, "conv_direct: statistics staging does not fit the input tile area");
    direct_epilogue<CO>(a, acc, n, ox0, oy0, tx, ty, in_lds);
}

// The up-sampling convolution (nearest 2x + 3x3, zero padding 1) of a narrow level in its PARITY FORM (conv.hip,
// conv_pack_weights_up2): output pixel (2y + a, 2x + b) = sum over i, j in {0, 1} of W'[a][b][i][j] . src(y + i + a - 1,
// x + j + b - 1), 4 multiply-adds per input channel instead of 9, and the staged tile is the 10 x 34 SOURCE window of the
// 64 x 16 outputs instead of an 18 x 66 window of the up-sampled image.  A thread owns 4 consecutive output pixels
// (2 source columns x both column parities) of one row; the rows of a wave all have the same parity (waves 0 and 2 take
// the even rows of their half of the tile, 1 and 3 the odd ones), so the weights stay wave-uniform scalar operands.
// Output: plain NCHW.  No prologue (the Upsample layer has none).
template <int CO, int DKC>
__global__ void __launch_bounds__(256, (CO <= 8 ? 5 : 3)) conv_direct_up2_kernel(ConvArgs a)
{
    constexpr int SH = DT_H / 2 + 2, SW = DT_W / 2 + 2, SP = 36, SCH = SH * SP;      // source window, LDS pitch (8-byte aligned runs)
    constexpr int LDS_FLOATS = (DKC * SCH > 2 * CO * 256) ? DKC * SCH : 2 * CO * 256;   // (the statistics staging reuses it)
    __shared__ __attribute__((aligned(16))) float in_lds[LDS_FLOATS];
    typedef const __attribute__((address_space(4))) float cfloat;
    cfloat *wk = (cfloat *)(unsigned long long)a.w_up2;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = lane & 15;
    const int ty = (wave >> 1) * 8 + (lane >> 4) * 2 + (wave & 1);      // row parity = wave & 1
    const int par_a = __builtin_amdgcn_readfirstlane(wave & 1);      // (uniform, and the compiler must know: scalar weight loads)
    const DTile dt = direct_tile();
    const int n = dt.n;
    const int ox0 = dt.bx * DT_W, oy0 = dt.by * DT_H;
    const int sx0 = (ox0 >> 1) - 1, sy0 = (oy0 >> 1) - 1;              // source coordinates of LDS (0, 0)
    const int Cin = a.C1, src_plane = a.Hs * a.Ws, plane_bytes = src_plane * 4;
    const int cin_pad = (Cin + 7) / 8 * 8;                              // rows of one parity's weight slab

    f32x2 acc[4][CO / 2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < CO / 2; ++c) acc[p][c] = f32x2{0.0f, 0.0f};

    constexpr int NSP = (SH * SW + 255) / 256;
    int sp_src[NSP], sp_dst[NSP];
#pragma unroll
    for (int j = 0; j < NSP; ++j) {
        const int e = tid + j * 256;
        const int r = e / SW, c = e % SW;
        const int sy = sy0 + r, sx = sx0 + c;
        const bool ok = e < SH * SW && sy >= 0 && sy < a.Hs && sx >= 0 && sx < a.Ws;      // outside the source: the conv's zero padding
        sp_src[j] = ok ? (sy * a.Ws + sx) * 4 : DOOB;
        sp_dst[j] = e < SH * SW ? r * SP + c : -1;
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.x1 + (size_t)n * Cin * src_plane), 0, Cin * plane_bytes, 0x00020000);

    for (int c0 = 0; c0 < Cin; c0 += DKC) {
        const int kc = min(DKC, Cin - c0);
        __syncthreads();
        float raw[DKC][NSP];
#pragma unroll
        for (int c = 0; c < DKC; ++c) {
            const int so = min(c0 + c, Cin - 1) * plane_bytes;
#pragma unroll
            for (int j = 0; j < NSP; ++j) raw[c][j] = dload(rsrc, sp_src[j], so);
        }
#pragma unroll
        for (int c = 0; c < DKC; ++c)
#pragma unroll
            for (int j = 0; j < NSP; ++j)
                if (sp_dst[j] >= 0) in_lds[c * SCH + sp_dst[j]] = raw[c][j];
        __syncthreads();
        for (int c = 0; c < kc; ++c) {
            // source columns 2 tx .. 2 tx + 3 of the window (x - 1 .. x + 2 for the thread's source pixels x, x + 1)
            const float *ip = in_lds + c * SCH + ((ty >> 1) + par_a) * SP + tx * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x2 lo = *reinterpret_cast<const f32x2 *>(ip + i * SP), hi = *reinterpret_cast<const f32x2 *>(ip + i * SP + 2);
                const float iv[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        // [parity a*2+b][Cin_pad8][2x2][cout_pad], plain layout
                        cfloat *wp = wk + (((size_t)(par_a * 2 + b) * cin_pad + c0 + c) * 4 + i * 2 + j) * a.cout_pad;
                        f32x2 wv[CO / 2];
#pragma unroll
                        for (int q = 0; q < CO / 2; ++q) wv[q] = f32x2{wp[2 * q], wp[2 * q + 1]};
#pragma unroll
                        for (int q = 0; q < CO / 2; ++q) {
                            acc[b][q] += wv[q] * iv[j + b];             // output 2x + b     (source pixel x:     columns 0, 1 | 1, 2)
                            acc[2 + b][q] += wv[q] * iv[1 + j + b];     // output 2x + 2 + b (source pixel x + 1: columns 1, 2 | 2, 3)
                        }
                    }
            }
        }
    }
    direct_epilogue<CO>(a, acc, n, ox0, oy0, tx, ty, in_lds);
}

template <int CO>
int launch_direct_s2(const ConvArgs &a, hipStream_t st)
{
    dim3 grid(cdiv(a.Wo, DT_W), cdiv(a.Ho, DT_H), a.B);
    const bool prof = prof_enabled();
    if (prof) prof_before(direct_class(a), st);
    hipLaunchKernelGGL((conv_direct_kernel<CO, 3, 2, false, 2>), grid, dim3(256), 0, st, a);
    if (prof) prof_after(direct_class(a), direct_work(a, 9), st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

template <int CO, int KS>
int launch_direct(const ConvArgs &a, hipStream_t st)
{
    dim3 grid(cdiv(a.Wo, DT_W), cdiv(a.Ho, DT_H), a.B);
    const bool prof = prof_enabled();
    if (prof) prof_before(direct_class(a), st);
    if constexpr (KS == 3 && CO >= 8) {
        if (a.sk_w) {
            if (a.sk_planar) hipLaunchKernelGGL((conv_direct_kernel<CO, KS, (CO <= 8 ? 4 : 8), false, 1, 2>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_direct_kernel<CO, KS, (CO <= 8 ? 4 : 8), false, 1, 1>), grid, dim3(256), 0, st, a);
            if (prof) prof_after(direct_class(a), direct_work(a, KS * KS), st);
            IPDM_LAUNCH_CHECK();
            return IPDM_OK;
        }
    }
    if (a.x1_planar) hipLaunchKernelGGL((conv_direct_kernel<CO, KS, (CO <= 8 ? 4 : 8), true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_direct_kernel<CO, KS, (CO <= 8 ? 4 : 8), false>), grid, dim3(256), 0, st, a);
    if (prof) prof_after(direct_class(a), direct_work(a, KS * KS), st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

}  // namespace

namespace ipdm {

bool conv_direct_eligible(const ConvArgs &a)
{
    const int max_cin = opt(OPT_DIRECT_MAX_CIN);
    const int cin = a.C1 + a.C2;
    const bool no_s2 = opt(OPT_DIRECT_NO_S2) != 0;      // A/B: stride 2 on the legacy 4-wave MFMA kernel
    if (a.stride == 2)
        return !no_s2 && a.ksize == 3 && a.Cout <= 16 && cin <= 32 && !a.upsample && !a.x1_planar && a.w_interleave == 0 && a.cout_pad >= 16;
    return (a.ksize == 3 || a.ksize == 1) && a.stride == 1 && a.Cout <= 16 && (cin <= max_cin || (a.Cout <= 4 && cin <= 128)) && a.w_interleave == 0 &&
           a.cout_pad >= 16;
}

// the narrow Upsample convolutions in parity form (w_up2 packed with the plain layout): exact 2x, no prologue, one source
bool conv_direct_up2_eligible(const ConvArgs &a)
{
    const bool off = opt(OPT_CONV_NO_UP2) != 0;      // (read per call, like conv_up2_eligible)
    return !off && a.w_up2 && a.w_interleave == 0 && a.ksize == 3 && a.stride == 1 && a.C2 == 0 && a.act == 0 && !a.res &&
           !a.x1_planar && a.H == 2 * a.Hs && a.W == 2 * a.Ws && a.Ho == a.H && a.Wo == a.W && a.Cout > 4 && a.Cout <= 16 &&
           a.C1 <= 64 && a.cout_pad >= 16;
}

int conv_direct_stats_rows(const ConvArgs &a) { return cdiv(a.Wo, DT_W) * cdiv(a.Ho, DT_H); }

// Can this 3x3 layer (shape fields + the sk_C1 / sk_C2 / sk_cout_pad of the shortcut's input) carry the block's 1x1 shortcut
// as extra K chunks?  Narrow levels only: 8 or 16 couts on the direct kernel, NCHW input, no residual of its own.
bool conv_direct_skip_ok(const ConvArgs &a)
{
    // (round 4: no longer switched off by conv_nm -- that coupling was the whole "paradox" of the 16-cout MFMA kernel: with the
    //  option on, the 16->16 layers ran 1.11-1.16x faster IN the network too, and every narrow block paid for its shortcut as
    //  a launch of its own again; a layer that carries a fused shortcut stays on this kernel either way)
    if (opt(OPT_CONV_NO_DIRECT) || opt(OPT_DIRECT_NO_SKIP_FUSE)) return false;
    return conv_direct_eligible(a) && a.ksize == 3 && a.stride == 1 && !a.upsample && !a.x1_planar && !a.C2 && !a.res && a.Cout > 4 && a.Cout <= 16 &&
           a.sk_C1 > 0 && a.sk_C1 + a.sk_C2 <= opt(OPT_DIRECT_MAX_CIN) && a.sk_cout_pad >= 16 && !conv_direct_up2_eligible(a);
}

int conv2d_direct_launch(const ConvArgs &a, hipStream_t st)
{
    IPDM_REQUIRE(!a.stats || a.stats_rows == conv_direct_stats_rows(a), "conv2d: statistics rows %d != %d", a.stats_rows,
                 conv_direct_stats_rows(a));
    IPDM_REQUIRE(!a.x1_planar || (!a.upsample && !(a.Hs & 1) && !(a.Ws & 1)), "conv2d: parity-planar input of odd size %dx%d", a.Hs, a.Ws);
    IPDM_REQUIRE((long)a.C1 * a.Hs * a.Ws < (1L << 29) && (long)(a.C2 + 1) * a.Hs * a.Ws < (1L << 29),
                 "conv2d: per-sample tensor exceeds the 2 GiB buffer-addressing range");
    IPDM_REQUIRE(!a.sk_w || conv_direct_skip_ok(a), "conv2d: this layer cannot carry a fused shortcut");
    // (with a fused shortcut the block INPUT -- up to direct_max_cin channels -- is the larger source: its descriptors are built
    //  in 32-bit arithmetic like the main input's)
    IPDM_REQUIRE(!a.sk_w || ((long)a.sk_C1 * a.Hs * a.Ws < (1L << 29) && (long)(a.sk_C2 + 1) * a.Hs * a.Ws < (1L << 29)),
                 "conv2d: the fused shortcut's source exceeds the 2 GiB buffer-addressing range");
    IPDM_REQUIRE(!a.sk_w || !a.sk_planar || (!(a.Hs & 1) && !(a.Ws & 1)), "conv2d: parity-planar shortcut source of odd size %dx%d", a.Hs, a.Ws);
    if (!a.sk_w && conv_nm_eligible(a)) return conv2d_nm_launch(a, st);      // same tiles, same statistics rows: interchangeable
    if (a.stride == 2) {
        if (a.Cout <= 4) return launch_direct_s2<4>(a, st);
        if (a.Cout <= 8) return launch_direct_s2<8>(a, st);
        return launch_direct_s2<16>(a, st);
    }
    if (conv_direct_up2_eligible(a)) {
        dim3 grid(cdiv(a.Wo, DT_W), cdiv(a.Ho, DT_H), a.B);
        const bool prof = prof_enabled();
        if (prof) prof_before(4, st);      // (parity form of a narrow Upsample: 16 -> 16, bandwidth-bound)
        if (a.Cout <= 8) hipLaunchKernelGGL((conv_direct_up2_kernel<8, 8>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_direct_up2_kernel<16, 8>), grid, dim3(256), 0, st, a);
        if (prof) prof_after(4, direct_bytes(a), st);
        IPDM_LAUNCH_CHECK();
        return IPDM_OK;
    }
    if (a.ksize == 1) {     // the 1x1 shortcuts of the narrow levels: pure streaming
        if (a.Cout <= 4) return launch_direct<4, 1>(a, st);
        if (a.Cout <= 8) return launch_direct<8, 1>(a, st);
        return launch_direct<16, 1>(a, st);
    }
    if (a.Cout <= 4) return launch_direct<4, 3>(a, st);
    if (a.Cout <= 8) return launch_direct<8, 3>(a, st);
    return launch_direct<16, 3>(a, st);
}

}  // namespace ipdm
