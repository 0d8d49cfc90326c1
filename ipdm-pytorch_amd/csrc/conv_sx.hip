// OPT-IN split-bf16 evaluation of the wide 3x3 stride-1 convolutions on the bf16 matrix pipe (IPDM_CONV_SPLIT=3 or 2
// at weight-packing time; the default path is the exact-f32 kernel of conv_ws.hip).
//
// Every f32 operand is split into NS bf16 pieces, x = x1 + x2 (+ x3) with x1 = bf16(x), x2 = bf16(x - x1), ..., and the
// product is evaluated as the sum of the piece products with i + j <= NS + 1 (NS = 3: 6 terms, every dropped term is
// below 2^-23 of the product: fp32-equivalent; NS = 2: 3 terms, 2^-16).  The piece products are exact in f32 and are
// accumulated in f32 by v_mfma_f32_32x32x16_bf16, smallest terms first.  tools/split_bf16_study.py (CPU) puts the
// end-to-end effect on the smoke pipeline at 5e-8 (6 terms) / 6e-7 (3 terms) relative PSNR against north_star's 1e-4.
//
// Why it can pay: the bf16 MFMA is a real matrix pipe (tools/ubench/coissue_bf16.hip: VALU, LDS and VMEM of both waves
// of a SIMD co-issue beside it), 16x the f32 MFMA rate per clock, at 1.5 GHz instead of 2.3 GHz under load.
//
// Structure (persistent, wave-specialised like conv_ws.hip):
//   * tile = 32 pixels x (8*WR rows) x (32*WM couts), WM*WR = 4 consumer waves; a consumer wave owns 32 couts and 8
//     rows: 8 accumulators.  Its weight pieces (A operands, pre-split and packed in MFMA lane order at create time)
//     come straight from L2 with one 16-byte load per piece and tap, prefetched a tap ahead -- they never touch LDS;
//   * producers stage the haloed input tile of the next 16-channel chunk: buffer loads, GroupNorm(+SiLU) (plain VALU:
//     no burst window is needed beside the bf16 pipe), split, and 16-byte LDS stores in [piece][pixel][16 ch] order,
//     so that a consumer lane's 8 consecutive channels of one pixel are one ds_read_b128;
//   * epilogue as in conv_ws.hip (bias by one f32 MFMA, DPP quad transpose, 16-byte stores).
#include <cstdlib>
#include <vector>
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SX_KC = 16;               // input channels per K chunk = K of one MFMA
constexpr int SX_OOB = 0x7fffffff;

__device__ inline float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

__device__ inline float silu_sx(float v)
{
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * v);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// round-to-nearest-even bf16 of a float, as the high 16 bits of the returned word's f32 image
__device__ inline unsigned bf16_bits(float x)
{
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x);
}

struct SxTile { int n, oy0, ox0, co0; };

template <int TH, int BN>
__device__ inline SxTile sx_decode(const ConvArgs &a, int tile)
{
    SxTile t;
    const int co_t = tile % a.co_tiles;
    int rest = tile / a.co_tiles;
    const int tx = rest % a.tiles_x;
    rest /= a.tiles_x;
    const int ty = rest % a.tiles_y;
    t.n = rest / a.tiles_y;
    t.oy0 = ty * TH;
    t.ox0 = tx * 32;
    t.co0 = co_t * BN;
    return t;
}

// WM: 32-cout blocks per tile (4 or 2); WR = 4 / WM row groups of 8 rows; NS: pieces per operand (3 or 2);
// CW: consumer waves per SIMD (1: four waves of 8 accumulators; 2: eight waves of 4 -- the bf16 matrix pipe takes
// another wave's MFMAs while one waits for its operands, which a single wave per SIMD cannot hide)
template <int WM, int NS, int CW>
__global__ void __launch_bounds__(256 * CW + 256) conv_sx_kernel(ConvArgs a, int ntiles)
{
    constexpr int NCT = 256 * CW;           // consumer threads
    constexpr int NQ = 8 / CW;              // rows (accumulators) per consumer wave
    constexpr int WR = 4 / WM, TH = 8 * WR, BN = 32 * WM;
    constexpr int IN_ROWS = TH + 2, IN_COLS = 34, NPIX = IN_ROWS * IN_COLS;
    constexpr int SP = (NPIX + 255) / 256;
    constexpr int PIECE = NPIX * SX_KC / 2;                 // dwords of one piece of the tile ([pixel][16 bf16])
    constexpr int STAGE = NS * PIECE;                       // dwords per LDS stage
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];

    const int G = gridDim.x, per = G >> 3;
    const int local = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int rounds = (ntiles + G - 1) / G;
    auto tile_of = [&](int k) { return k * G + (local + 5 * k) % G; };
    const int n_my = rounds == 0 ? 0 : (tile_of(rounds - 1) < ntiles ? rounds : rounds - 1);
    const int Ctot = a.C1 + a.C2;
    const int nchunks = (Ctot + SX_KC - 1) / SX_KC;
    const int S = n_my * nchunks;
    const int plane_bytes = a.Hs * a.Ws * 4;

    if (threadIdx.x >= NCT) {
        // =========================================================================== PRODUCERS
        const int tid = threadIdx.x - NCT;
        int in_voff[SP];
        bool in_ok[SP];
        SxTile t = {0, 0, 0, 0};
        for (int s = 0; s < S; ++s) {
            const int k = s / nchunks, ch = s - k * nchunks;
            if (ch == 0) {
                t = sx_decode<TH, BN>(a, tile_of(k));
                const int iy0 = t.oy0 - 1, ix0 = t.ox0 - 1;
#pragma unroll
                for (int j = 0; j < SP; ++j) {
                    const int sp = tid + j * 256;
                    const int r = sp / IN_COLS, c = sp % IN_COLS;
                    const int iy = iy0 + r, ix = ix0 + c;
                    in_ok[j] = sp < NPIX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    int sy = min(max(iy, 0), a.H - 1), sx = min(max(ix, 0), a.W - 1);
                    if (a.upsample) {   // F.interpolate(mode="nearest"): src = min(floor(dst * (in/out) in f32), in-1)
                        sy = min((int)floorf((float)sy * a.scale_y), a.Hs - 1);
                        sx = min((int)floorf((float)sx * a.scale_x), a.Ws - 1);
                    }
                    in_voff[j] = in_ok[j] ? (sy * a.Ws + sx) * 4 : SX_OOB;
                }
            }
            const int c0 = ch * SX_KC;
            const int nvalid = min(SX_KC, Ctot - c0);
            // a chunk never straddles the two concatenated sources (executor: C1 % 16 == 0 when C2 > 0)
            const bool from1 = c0 < a.C1;
            const int csrc = from1 ? a.C1 : a.C2;
            const float *src = from1 ? a.x1 + (size_t)t.n * a.C1 * (plane_bytes / 4) : a.x2 + (size_t)t.n * a.C2 * (plane_bytes / 4);
            const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, csrc * plane_bytes, 0x00020000);
            const int cs0 = from1 ? c0 : c0 - a.C1;
            float raw[SP][SX_KC];
#pragma unroll
            for (int c = 0; c < SX_KC; ++c) {
                const int soff = (cs0 + min(c, nvalid - 1)) * plane_bytes;     // channels beyond Cin meet zero weights
#pragma unroll
                for (int j = 0; j < SP; ++j) raw[j][c] = bload(x_rsrc, in_voff[j], soff);
            }
            if (a.act) {
                const float *gsc = a.gn_scale + (size_t)t.n * Ctot + c0, *gsh = a.gn_shift + (size_t)t.n * Ctot + c0;
#pragma unroll
                for (int c = 0; c < SX_KC; ++c) {
                    const float sc = c < nvalid ? gsc[c] : 0.0f, sh = c < nvalid ? gsh[c] : 0.0f;
#pragma unroll
                    for (int j = 0; j < SP; ++j) {
                        float v = raw[j][c] * sc + sh;
                        if (a.act == 2) v = silu_sx(v);
                        raw[j][c] = in_ok[j] ? v : 0.0f;        // zero padding is re-imposed after the activation
                    }
                }
            }
            // split into NS bf16 pieces and store [piece][pixel][16 ch]: 2 x 16 bytes per pixel and piece
            unsigned *st = lds + (s & 1) * STAGE;
#pragma unroll
            for (int j = 0; j < SP; ++j) {
                const int sp = tid + j * 256;
                if (SP * 256 == NPIX || sp < NPIX) {
                    float r[SX_KC];
#pragma unroll
                    for (int c = 0; c < SX_KC; ++c) r[c] = raw[j][c];
#pragma unroll
                    for (int p = 0; p < NS; ++p) {
                        unsigned w[SX_KC / 2];
#pragma unroll
                        for (int c = 0; c < SX_KC; c += 2) {
                            const unsigned lo = bf16_bits(r[c]), hi = bf16_bits(r[c + 1]);
                            w[c / 2] = lo | (hi << 16);
                            if (p + 1 < NS) {
                                r[c] -= __builtin_bit_cast(float, lo << 16);
                                r[c + 1] -= __builtin_bit_cast(float, hi << 16);
                            }
                        }
                        u32x4 *dst = reinterpret_cast<u32x4 *>(st + p * PIECE + sp * (SX_KC / 2));
                        dst[0] = u32x4{w[0], w[1], w[2], w[3]};
                        dst[1] = u32x4{w[4], w[5], w[6], w[7]};
                    }
                }
            }
            __syncthreads();                       // hand-over: stage (s&1) is complete
        }
        return;
    }

    // =============================================================================== CONSUMERS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 5, l31 = lane & 31;
    const int wm = (wave & 3) % WM, wr = (wave & 3) / WM;      // this wave's cout block and row group
    const int swm = __builtin_amdgcn_readfirstlane(wm);
    const int row0 = __builtin_amdgcn_readfirstlane(wr * 8 + (wave >> 2) * NQ);     // first of this wave's NQ rows
    f32x16 acc[NQ];
    const int out_plane = a.Ho * a.Wo;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
    // packed weights: [cout block][chunk][tap][piece][lane 64][8 bf16] = 1 KB per (block, chunk, tap, piece)
    const u32x4 *wbase = reinterpret_cast<const u32x4 *>(a.w);
    const int cblocks_stride = nchunks * 9 * NS * 64;   // u32x4 units per cout block

    for (int s = 0; s < S; ++s) {
        const int k = s / nchunks, ch = s - k * nchunks;
        const SxTile t = sx_decode<TH, BN>(a, tile_of(k));
        if (ch == 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
        }
        // this wave's weight pieces of tap 0 can be fetched before the hand-over
        // (a ragged last tile may own cout blocks beyond the packed ones: it re-reads the last block, its results are dropped)
        const u32x4 *wp = wbase + (size_t)min(t.co0 / 32 + swm, a.cout_pad / 32 - 1) * cblocks_stride + (size_t)ch * 9 * NS * 64 + lane;
        u32x4 an[NS], ac[NS];
#pragma unroll
        for (int p = 0; p < NS; ++p) an[p] = wp[p * 64];
        __syncthreads();                           // hand-over: stage (s&1) is complete
        const unsigned *st = lds + (s & 1) * STAGE;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int p = 0; p < NS; ++p) ac[p] = an[p];
            if (tap + 1 < 9) {
#pragma unroll
                for (int p = 0; p < NS; ++p) an[p] = wp[((tap + 1) * NS + p) * 64];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int pix = (row0 + q + ky) * IN_COLS + l31 + kx;
                u32x4 b[NS];
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    b[p] = *reinterpret_cast<const u32x4 *>(st + p * PIECE + pix * (SX_KC / 2) + lk * 4);
#define SX_MFMA(i, j) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ac[i]), __builtin_bit_cast(bf16x8, b[j]), acc[q], 0, 0, 0)
                if (NS == 3) { SX_MFMA(0, 2); SX_MFMA(2, 0); SX_MFMA(1, 1); }
                SX_MFMA(0, 1);
                SX_MFMA(1, 0);
                SX_MFMA(0, 0);
#undef SX_MFMA
            }
        }
        if (ch == nchunks - 1) {
            // + bias (one f32 MFMA per accumulator: A = bias of the lane's cout on the k=0 half, B = 1), then the epilogue
            const float bv = bload(b_rsrc, lk ? SX_OOB : l31 * 4, min(t.co0 + swm * 32, a.Cout - 1) * 4);
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, 1.0f, acc[q], 0, 0, 0);
            const size_t sample = (size_t)t.n * a.Cout * out_plane;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + sample), 0, a.Cout * out_plane * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((a.res ? a.res : a.out) + sample), 0, a.Cout * out_plane * 4, 0x00020000);
            const int plane4 = out_plane * 4;
            const int cob = t.co0 + swm * 32;
            if (cob < a.Cout) {                    // uniform; false only on a ragged cout tile
                if ((a.Wo & 3) == 0 && cob + 32 <= a.Cout) {
                    // 16-byte stores after a 4x4 transpose inside lane quads (see conv_ws.hip)
                    const int qi = l31 & 3, qx = l31 >> 2;
                    const int lane_off4 = ((qi + 4 * lk) * out_plane + 4 * qx) * 4;
                    const bool xok = t.ox0 + 4 * qx + 4 <= a.Wo;
                    const bool odd = (l31 & 1) != 0, hi = (l31 & 2) != 0;
#define SX_XCHG(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int oy = t.oy0 + row0 + q;
                        const int voff = (xok && oy < a.Ho) ? lane_off4 : SX_OOB;
                        const int so = cob * plane4 + (min(oy, a.Ho - 1) * a.Wo + t.ox0) * 4;
                        f32x4 rv[4];
                        if (a.res) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                rv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, voff, so + 8 * g * plane4, 0));
                        }
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float r0 = acc[q][4 * g], r1 = acc[q][4 * g + 1], r2 = acc[q][4 * g + 2], r3 = acc[q][4 * g + 3];
                            float x = SX_XCHG(r0, 0xB1), y = SX_XCHG(r1, 0xB1);
                            r0 = odd ? y : r0; r1 = odd ? r1 : x;
                            x = SX_XCHG(r2, 0xB1); y = SX_XCHG(r3, 0xB1);
                            r2 = odd ? y : r2; r3 = odd ? r3 : x;
                            x = SX_XCHG(r0, 0x4E); y = SX_XCHG(r2, 0x4E);
                            r0 = hi ? y : r0; r2 = hi ? r2 : x;
                            x = SX_XCHG(r1, 0x4E); y = SX_XCHG(r3, 0x4E);
                            r1 = hi ? y : r1; r3 = hi ? r3 : x;
                            f32x4 v = {r0, r1, r2, r3};
                            if (a.res) v += rv[g];
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, voff, so + 8 * g * plane4, 0);
                        }
                    }
#undef SX_XCHG
                } else {
                    const int lane_off = (lk * 4 * out_plane + l31) * 4;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int oy = t.oy0 + row0 + q;
                        const int voff = (t.ox0 + l31 < a.Wo && oy < a.Ho) ? lane_off : SX_OOB;
                        int so = cob * plane4 + (min(oy, a.Ho - 1) * a.Wo + t.ox0) * 4;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const bool cok = cob + (r & 3) + 8 * (r >> 2) < a.Cout;     // (+4 lk stays below a multiple of 8 of Cout)
                            float v = acc[q][r];
                            if (cok) {
                                if (a.res) v += bload(r_rsrc, voff, so);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, voff, so, 0);
                            }
                            so += ((r & 3) == 3 ? 5 : 1) * plane4;
                            asm volatile("" : "+s"(so));
                        }
                    }
                }
            }
        }
    }
}

int sx_num_cus() { return device_cu_count(); }

template <int WM, int NS, int CW>
int launch_sx(const ConvArgs &args, hipStream_t st)
{
    constexpr int WR = 4 / WM, TH = 8 * WR, BN = 32 * WM;
    constexpr int NPIX = (TH + 2) * 34;
    constexpr size_t lds = (size_t)2 * NS * NPIX * SX_KC / 2 * sizeof(unsigned);
    static_assert(lds <= 160 * 1024, "conv_sx: LDS stages exceed 160 KiB");
    ConvArgs a = args;
    a.tiles_x = cdiv(a.Wo, 32);
    a.tiles_y = cdiv(a.Ho, TH);
    a.co_tiles = cdiv(a.Cout, BN);
    IPDM_REQUIRE(a.C2 == 0 || a.C1 % SX_KC == 0, "conv2d(split): concat split %d not a multiple of the K chunk %d", a.C1, SX_KC);
    IPDM_REQUIRE((long)a.C1 * a.Hs * a.Ws < (1L << 29) && (long)(a.C2 + 1) * a.Hs * a.Ws < (1L << 29) &&
                     (long)a.Cout * a.Ho * a.Wo < (1L << 29), "conv2d(split): per-sample tensor exceeds the 2 GiB buffer range");
    const long ntiles = (long)a.tiles_x * a.tiles_y * a.co_tiles * a.B;
    IPDM_REQUIRE(ntiles < (1L << 31), "conv2d(split): too many tiles");
    const int cus = sx_num_cus();
    int G = (int)(ntiles < cus ? ntiles : cus);
    G = (G + 7) / 8 * 8;
    if (int rc = ensure_dynamic_lds((const void *)conv_sx_kernel<WM, NS, CW>, lds)) return rc;
    const bool prof = prof_enabled();
    if (prof) prof_before(0, st);
    hipLaunchKernelGGL((conv_sx_kernel<WM, NS, CW>), dim3((unsigned)G), dim3(256 * CW + 256), lds, st, a, (int)ntiles);
    if (prof) prof_after(0, 2.0 * a.B * a.Ho * a.Wo * (double)a.Cout * (a.C1 + a.C2) * 9, st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

}  // namespace

namespace ipdm {

// w_interleave codes of the split layout: 100 + NS
int conv_sx_pieces(int interleave) { return interleave >= 100 ? interleave - 100 : 0; }

// [Cout][Cin][3][3] f32 (host) -> [cout block 32][chunk 16][tap][piece][lane 64][8] bf16 (as uint16), zero padded
static void conv_sx_pack_weights_impl(const float *w, int Cout, int Cin, int ns, std::vector<float> &packed, int &cin_pad, int &cout_pad)
{
    cin_pad = (Cin + SX_KC - 1) / SX_KC * SX_KC;
    cout_pad = (Cout + 31) / 32 * 32;
    const int nchunks = cin_pad / SX_KC, nblk = cout_pad / 32;
    std::vector<unsigned short> out((size_t)nblk * nchunks * 9 * ns * 64 * 8, 0);
    auto to_bf16 = [](float x) -> unsigned short {      // round to nearest even
        unsigned u;
        memcpy(&u, &x, 4);
        const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
        return (unsigned short)(r >> 16);
    };
    for (int blk = 0; blk < nblk; ++blk)
        for (int ch = 0; ch < nchunks; ++ch)
            for (int tap = 0; tap < 9; ++tap)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = blk * 32 + (lane & 31), ci = ch * SX_KC + (lane >> 5) * 8 + j;
                        float r = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.0f;
                        for (int p = 0; p < ns; ++p) {
                            const unsigned short h = to_bf16(r);
                            unsigned u = (unsigned)h << 16;
                            float hf;
                            memcpy(&hf, &u, 4);
                            r -= hf;
                            out[(((((size_t)blk * nchunks + ch) * 9 + tap) * ns + p) * 64 + lane) * 8 + j] = h;
                        }
                    }
    packed.assign((out.size() + 1) / 2, 0.0f);
    memcpy(packed.data(), out.data(), out.size() * 2);
}

static int conv2d_sx_launch_impl(const ConvArgs &a, hipStream_t st)
{
    const int ns = conv_sx_pieces(a.w_interleave);
    IPDM_REQUIRE(a.ksize == 3 && a.stride == 1 && (ns == 2 || ns == 3), "conv2d(split): unsupported configuration");
    const bool wide = a.Cout > 96;
    // A/B switch: two consumer waves per SIMD (eight waves of 4 accumulators).  Measured equal to one (226 vs 228
    // TFLOP/s-equivalent on 128->128 @512^2): at 66 % matrix-pipe occupancy and 1.7 GHz the kernel sits at 86 % of the
    // 1585 TFLOP/s the chip sustains on a pure bf16 MFMA stream -- the limit is the power-managed clock, not issue bubbles.
    const bool cw2 = opt(OPT_CONV_SX_CW2) != 0;
    if (cw2) {
        if (ns == 3) return wide ? launch_sx<4, 3, 2>(a, st) : launch_sx<2, 3, 2>(a, st);
        return wide ? launch_sx<4, 2, 2>(a, st) : launch_sx<2, 2, 2>(a, st);
    }
    if (ns == 3) return wide ? launch_sx<4, 3, 1>(a, st) : launch_sx<2, 3, 1>(a, st);
    return wide ? launch_sx<4, 2, 1>(a, st) : launch_sx<2, 2, 1>(a, st);
}

}  // namespace ipdm

// entry points of libipdm_hip_optin.so (resolved by csrc/optin.hip of the product library when an opt-in mode is switched on)
extern "C" void ipdm_optin_conv_sx_pack_weights(const float *w, int Cout, int Cin, int ns, std::vector<float> *packed, int *cin_pad, int *cout_pad)
{
    ipdm::conv_sx_pack_weights_impl(w, Cout, Cin, ns, *packed, *cin_pad, *cout_pad);
}
extern "C" int ipdm_optin_conv2d_sx_launch(const ipdm::ConvArgs *a, hipStream_t st) { return ipdm::conv2d_sx_launch_impl(*a, st); }
