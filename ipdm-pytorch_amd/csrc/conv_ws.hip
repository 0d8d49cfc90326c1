// Persistent, wave-specialised implicit-GEMM convolution for gfx950 (the dominant kernel of the path).
//
// Same mathematics and fused prologue/epilogue as conv.hip (nearest up-sampling + channel concat as input
// addressing, GroupNorm(+SiLU) applied while staging, bias / time-embedding / residual epilogue, exact-f32
// v_mfma_f32_32x32x2_f32), different execution structure -- chosen from the round-1 counters of conv.hip
// (profiles/r01b_pmc_sq_*: matrix pipe 73 % busy, both co-resident workgroups parked on the same global-load
// latency in lockstep):
//
//   * one 512-thread workgroup per CU, PERSISTENT over a static round-robin of output tiles;
//   * waves 0-3 (one per SIMD) are CONSUMERS: nothing but ds_read + MFMA + the tile epilogue.  Their
//     instruction stream never touches global memory inside the K loop, so the matrix pipe of each SIMD is
//     fed back-to-back by a single wave whose LDS operands are fetched one tap ahead;
//   * waves 4-7 (the SIMD partners of 0-3) are PRODUCERS: they issue the buffer loads of the NEXT K chunk
//     (which may belong to the next tile) with VALU-free addressing and fill the other LDS stage while the
//     consumers compute -- VMEM / LDS / SALU work issues beside the partner's MFMAs for free;
//   * one workgroup barrier per K chunk hands the stage over; the chunk stream runs across tile
//     boundaries, so there is no per-tile prologue bubble and the epilogue stores of tile i drain while
//     tile i+1 is already being multiplied.
//
// Tile = (4*NB rows) x 32 cols of output pixels x (32*MB) couts; consumer wave w owns rows
// [w*NB,(w+1)*NB).  LDS per stage: input halo tile [KC][IN_ROWS][IN_COLS] + weight slab [KC][taps][32][IL].
//
// Issue model measured on MI355X (tools/ubench/coissue.hip): v_mfma_f32_32x32x2_f32 occupies the
// SIMD's vector ALU for its whole 64 cycles.  Every other instruction of the MFMA wave is additive (~9 cycles); a
// partner wave's LDS / VMEM / SALU ops issue freely, but its VALU ops only get the MFMA wave's own stall gaps
// (~280 cycles per op under a saturated stream, at any priority).  Hence:
//   (i)   8 accumulators per consumer wave and ONE wide LDS read per tap for the weights (the slab is stored
//         cout-interleaved, so the MB values of a lane are adjacent: ds_read_b64/b128): 0.3 LDS instructions per MFMA;
//   (ii)  producers do no address arithmetic on the VALU (buffer loads: one per-lane offset per tile + scalar offsets;
//         out-of-range offsets implement the zero padding);
//   (iii) the one unavoidable VALU job, GroupNorm(+SiLU) of the staged tile, runs as a burst in a short window in
//         which the consumers wait at a barrier (~800 cycles per 18.4k-cycle chunk) instead of starving;
//   (iv)  bias is one MFMA per accumulator after the last chunk; the epilogue transposes 4x4 blocks inside lane quads
//         (DPP) so that stores and residual loads move 16 bytes per lane.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

// IL: cout interleave of the packed weights (>= MB).  PBW: the 32 pixels of one MFMA operand are PBW columns x 32/PBW rows
// (32x1 by default; 8x4 for image widths with a large remainder modulo 32: 228, 114)
template <int KS, int STRIDE, int MB, int NB, int KC_, int IL = MB, int PBW_ = 32>
struct WsTile {
    static constexpr int PBW = PBW_;
    static constexpr int KC = KC_;
    static constexpr int TAPS = KS * KS;
    static constexpr int PBH = 32 / PBW;
    static constexpr int TH = 4 * NB * PBH;
    static constexpr int TW = PBW;
    static constexpr int IN_ROWS = (TH - 1) * STRIDE + KS;
    static constexpr int IN_COLS = (TW - 1) * STRIDE + KS;
    // LDS row pitch: the PBH lane rows of an operand read must fall on disjoint banks (pitch = 8 or 24 mod 32)
    static constexpr int PITCH = PBW == 32 ? IN_COLS : (IN_COLS <= 8 ? 8 : (IN_COLS <= 24 ? 24 : 40));
    static constexpr int IN_CH = IN_ROWS * PITCH;
    static constexpr int SP = (IN_CH + 255) / 256;           // staging slots per producer thread per channel
    static constexpr int IN_CHP = SP * 256;                  // LDS channel pitch: every (thread, slot) owns an address
    static constexpr int IN_TILE = KC * IN_CHP;
    static constexpr int BN = 32 * MB;                       // couts per tile
    static constexpr int LW = 32 * IL;                       // couts per weight-slab row (one interleave group)
    static constexpr int W_TILE = KC * TAPS * LW;
    static constexpr int W_VEC = (W_TILE / 4 + 255) / 256;   // 16-byte weight loads per producer thread per chunk
    static constexpr int W_TILEP = W_VEC * 256 * 4;
    static constexpr int BUF = IN_TILE + W_TILEP;
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF * sizeof(float);
};

// extra LDS behind the two stages: 4 x 256 floats, the statistics rows of the 4 consumer waves
template <class T>
constexpr size_t ws_lds_total() { return T::LDS_BYTES + 4 * 256 * 4; }

// In-kernel s_memtime stamps of the consumer / producer phases (tools/bench_conv_dbg.py) are compiled in only with
// -DIPDM_CONV_STAMPS=1 (`make stamps` builds ../libipdm_hip_stamps.so): their ten 64-bit counters cost scalar registers the
// persistent loops need -- spilled SGPRs come back as v_readlane, and a producer's VALU instructions only get the stall
// gaps of the MFMA wave.  Compiled out: 3x3 convolutions 1.2 % faster, 1x1 6 % (tools/ab_lib.py, interleaved).
#ifndef IPDM_CONV_STAMPS
#define IPDM_CONV_STAMPS 0
#endif
struct TileId { int n, oy0, ox0, co0, ks; };      // ks: which slice of the K (input channel) range, ConvArgs::ksplit

template <int TH, int TW, int BN>
__device__ inline TileId decode_tile(const ConvArgs &a, int tile)
{
    TileId t;
    const int co_t = tile % a.co_tiles;
    int rest = tile / a.co_tiles;
    const int tx = rest % a.tiles_x;
    rest /= a.tiles_x;
    const int ty = rest % a.tiles_y;
    rest /= a.tiles_y;
    t.n = rest % a.B;
    t.ks = rest / a.B;
    t.oy0 = ty * TH;
    t.ox0 = tx * TW;
    t.co0 = co_t * BN;
    return t;
}

// Per-lane (VGPR) buffer offset that is out of range: loads return 0 and stores are dropped, no branch.  The hardware
// compares the VGPR offset with (num_records - scalar offset), so the SCALAR offset must always stay inside the buffer:
// invalid elements are always killed through the per-lane offset, never through the scalar one.
constexpr int OOB = 0x7fffffff;

__device__ inline float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// SiLU(x*sc+sh) on a pair: 3 packed VALU + 2 exp + 2 rcp (the producers get ~1 VALU issue per partner MFMA)
__device__ inline f32x2 gn_silu2(f32x2 x, float sc, float sh)
{
    f32x2 z = x * sc + sh;
    f32x2 e = z * -1.4426950408889634f;
    e[0] = __builtin_amdgcn_exp2f(e[0]);
    e[1] = __builtin_amdgcn_exp2f(e[1]);
    e = e + 1.0f;
    e[0] = __builtin_amdgcn_rcpf(e[0]);
    e[1] = __builtin_amdgcn_rcpf(e[1]);
    return z * e;
}

// ---- cross-lane sums for the fused GroupNorm statistics (epilogue).  DPP row rotations inside rows of 16 lanes, then
// ds_swizzle (LDS crossbar, no memory) across the two rows of a 32-lane half; every participating lane ends with the total.
#define IPDM_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))
__device__ inline float sum_lanes_stride4(float x)      // over the 8 lanes {l : l % 4 == lane % 4} of the lane's 32-lane half
{
    x += IPDM_DPP_F(x, 0x124);                           // row_ror:4
    x += IPDM_DPP_F(x, 0x128);                           // row_ror:8
    x += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));   // lane ^ 16
    return x;
}
__device__ inline float sum_lanes_half(float x)         // over all 32 lanes of the lane's half
{
    x += IPDM_DPP_F(x, 0x121);                           // row_ror:1
    x += IPDM_DPP_F(x, 0x122);                           // row_ror:2
    return sum_lanes_stride4(x);
}
#undef IPDM_DPP_F

template <int KS, int STRIDE, int MB, int NB, int KC, int IL, int PBW, bool VEC4, bool PLANAR>
__global__ void __launch_bounds__(512) conv_ws_kernel(ConvArgs a, int ntiles)
{
    using T = WsTile<KS, STRIDE, MB, NB, KC, IL, PBW>;
    constexpr bool UP2 = KS == 2;      // the up-sampling convolution's parity form (below)
    extern __shared__ __attribute__((aligned(16))) float lds[];

    // ---- static tile schedule: at step k the G workgroups cover tiles [kG,(k+1)G); the workgroups of one XCD
    //      (blockIdx % 8) take a contiguous run of them, so the cout tiles / halo neighbours that re-read the
    //      same input share that XCD's L2.
    //      The slot is rotated by 5 every round: with a fixed slot a workgroup would see the same tile column every
    //      round (G is a multiple of tiles_x for the usual sizes) and the ones on the ragged right edge would only
    //      ever get partial tiles.
    const int G = gridDim.x, per = G >> 3;
    const int local = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int rounds = (ntiles + G - 1) / G;
    auto tile_of = [&](int k) { return k * G + (local + 5 * k) % G; };
    const int n_my = rounds == 0 ? 0 : (tile_of(rounds - 1) < ntiles ? rounds : rounds - 1);
    const int Ctot = a.C1 + a.C2;
    // K split (ConvArgs::ksplit > 1: layers with too few tiles to fill the chip): a "tile" of the schedule is then one
    // slice of the channel chunks of an output tile, and its sums go to slice ks of a partial buffer (no bias, residual
    // or statistics here: splitk_combine adds the slices in a fixed order and applies them)
    // up2 (ConvArgs::up2, KS = 2): the convolution of a 2x nearest up-sampled image, evaluated on the SOURCE grid as four
    // 2x2-tap convolutions, one per output parity (a, b) = (row & 1, col & 1): the 3x3 taps that fall on the same source
    // pixel were added up when the weights were packed (4 instead of 9 multiply-adds per output).  The parity is the `ks`
    // digit of the tile; its weights are slab `ks`, its window starts at (oy0 + a - 1, ox0 + b - 1), and its outputs go
    // to plane `ks` of the parity-planar output [n][cout][a][b][Ho][Wo] (readers: ConvArgs::x1_planar).
    const int nchunks = (Ctot + KC - 1) / KC / (UP2 ? 1 : a.ksplit);
    const int S = n_my * nchunks;                  // chunks in this workgroup's stream
    const int plane_bytes = a.Hs * a.Ws * 4;
    if (threadIdx.x >= 256) {
        // =========================================================================== PRODUCERS
        // Addressing is VALU-free: buffer loads take a per-thread byte offset that is constant for a tile (input) or
        // for the whole kernel (weights) plus a scalar offset per channel / chunk; out-of-range offsets read 0.
        const int tid = threadIdx.x - 256;
        int w_voff[T::W_VEC];
#pragma unroll
        for (int e = 0; e < T::W_VEC; ++e) {
            const int v4 = tid + e * 256;
            const int row = v4 / (T::LW / 4), col4 = v4 % (T::LW / 4);
            w_voff[e] = v4 < T::W_TILE / 4 ? (row * a.cout_pad + col4 * 4) * 4 : OOB;
        }
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)a.w, 0, ((Ctot + KC - 1) / KC * KC) * T::TAPS * a.cout_pad * 4 * (UP2 ? 4 : 1), 0x00020000);
        int in_voff[T::SP], in_voffp[T::SP];      // in_voffp: the same elements inside a parity-planar x1
        bool in_ok[T::SP];
        TileId t = {0, 0, 0, 0};
        bool border = false;
        const bool pstamp = IPDM_CONV_STAMPS && (a.dbg & 8) != 0;
        unsigned long long p_issue = 0, p_wait = 0, p_math = 0, p_store = 0, p_t = 0;
        for (int s = 0; s < S; ++s) {
            if (pstamp) p_t = __builtin_amdgcn_s_memtime();
            {
                const int k = s / nchunks, ch = s - k * nchunks;
                if (ch == 0) {      // new tile: spatial descriptors (the same for every channel chunk of the tile)
                    t = decode_tile<T::TH, T::TW, T::BN>(a, tile_of(k));
                    const int iy0 = t.oy0 * STRIDE - (KS == 2 ? 1 - (t.ks >> 1) : KS / 2);
                    const int ix0 = t.ox0 * STRIDE - (KS == 2 ? 1 - (t.ks & 1) : KS / 2);
                    border = iy0 < 0 || ix0 < 0 || iy0 + T::IN_ROWS > a.H || ix0 + T::IN_COLS > a.W;
#pragma unroll
                    for (int j = 0; j < T::SP; ++j) {
                        const int sp = tid + j * 256;
                        const int r = sp / T::PITCH, c = sp % T::PITCH;
                        const int iy = iy0 + r, ix = ix0 + c;
                        in_ok[j] = sp < T::IN_CH && c < T::IN_COLS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                        int sy = min(max(iy, 0), a.H - 1), sx = min(max(ix, 0), a.W - 1);
                        if (a.upsample) {   // F.interpolate(mode="nearest"): src = min(floor(dst * (in/out) in f32), in-1)
                            sy = min((int)floorf((float)sy * a.scale_y), a.Hs - 1);
                            sx = min((int)floorf((float)sx * a.scale_x), a.Ws - 1);
                        }
                        in_voff[j] = in_ok[j] ? (sy * a.Ws + sx) * 4 : OOB;      // padding / idle slots read 0
                        if (PLANAR) in_voffp[j] = in_ok[j] ? ((((sy & 1) * 2 + (sx & 1)) * (a.Hs >> 1) + (sy >> 1)) * (a.Ws >> 1) + (sx >> 1)) * 4 : OOB;
                    }
                }
                const int c0 = ((UP2 ? 0 : t.ks * nchunks) + ch) * KC;
                const int nvalid = min(KC, Ctot - c0);
                float *ib = lds + (s & 1) * T::BUF;
                // a chunk never straddles the two concatenated sources (launcher: C1 % KC == 0 when C2 > 0)
                const bool from1 = c0 < a.C1;
                const int csrc = from1 ? a.C1 : a.C2;
                const float *src = from1 ? a.x1 + (size_t)t.n * a.C1 * (plane_bytes / 4)
                                         : a.x2 + (size_t)t.n * a.C2 * (plane_bytes / 4);
                const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, csrc * plane_bytes, 0x00020000);
                const int cs0 = from1 ? c0 : c0 - a.C1;
                // weights first (they need no transform and go to LDS as soon as they land), then the raw input tile
                f32x4 w_reg[T::W_VEC];
                const int w_row0 = UP2 ? t.ks * (nchunks * KC) : 0;                         // up2: the parity's weight slab
                const int w_soff = ((w_row0 + c0) * T::TAPS * a.cout_pad + t.co0 / T::LW * T::LW) * 4;   // the whole interleave group of the tile
#pragma unroll
                for (int e = 0; e < T::W_VEC; ++e)
                    w_reg[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff[e], w_soff, 0));
                float raw[KC][T::SP];
                auto load_raw = [&](const int (&voff)[T::SP]) __attribute__((always_inline)) {
#pragma unroll
                    for (int c = 0; c < KC; ++c) {
                        // channels beyond Cin re-read the last real channel (in-range scalar offset); their weights are zero
                        const int soff = (cs0 + min(c, nvalid - 1)) * plane_bytes;
#pragma unroll
                        for (int j = 0; j < T::SP; ++j) raw[c][j] = bload(x_rsrc, voff[j], soff);
                    }
                };
                if (PLANAR && from1) load_raw(in_voffp);      // (uniform branch: a select per load would be VALU)
                else load_raw(in_voff);
                // stage (s&1) was last read by chunk s-2, which the consumers finished before the previous hand-over
#pragma unroll
                for (int e = 0; e < T::W_VEC; ++e)
                    *reinterpret_cast<f32x4 *>(ib + T::IN_TILE + (tid + e * 256) * 4) = w_reg[e];
                if (pstamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); p_issue += now - p_t; p_t = now; }
                float *dst = ib + tid;
                if (a.act) {
                    // GroupNorm(+SiLU) of the staged values.  The f32 MFMA executes on the SIMD's vector ALU: while the
                    // partner wave streams MFMAs this wave's VALU only gets the partner's stall gaps (measured ~280
                    // cycles per op, any priority).  So the transform runs as a burst inside a window in which the
                    // consumers wait (barrier A .. hand-over barrier): ~100 VALU ops at full rate instead of a whole
                    // chunk time of starvation.
                    const float *gsc = a.gn_scale + (size_t)t.n * Ctot + c0, *gsh = a.gn_shift + (size_t)t.n * Ctot + c0;
                    float scv[KC], shv[KC];
#pragma unroll
                    for (int c = 0; c < KC; ++c) {
                        scv[c] = gsc[c];           // (no `c < nvalid` selects: channels beyond Cin have zero weights, and the
                        shv[c] = gsh[c];           //  arrays are padded by a K chunk -- 16 starved VALU instructions fewer)
                    }
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    if (pstamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); p_wait += now - p_t; p_t = now; }
                    __syncthreads();                       // A: consumers have finished chunk s-1
                    if (pstamp) p_t = __builtin_amdgcn_s_memtime();
                    // The staged values of the chunk as a flat list of pairs, transformed FOUR PAIRS ABREAST, stage by
                    // stage (multiply-add | scale | exp | +1 | rcp | multiply): one pair at a time is a chain of ten
                    // dependent instructions, four of them transcendental, and the window is as long as that chain times
                    // the number of pairs.  Each group goes to LDS as soon as it is done (the stores, not VALU, issue
                    // beside the next group's math); zero padding is re-imposed AFTER the activation (border tiles only).
                    auto transform = [&](auto silu_tag) __attribute__((always_inline)) {
                        constexpr bool SILU = decltype(silu_tag)::value;
                        constexpr int E = KC * T::SP, NPAIR = (E + 1) / 2, GROUP = 8;
#pragma unroll
                        for (int p0 = 0; p0 < NPAIR; p0 += GROUP) {
                            f32x2 z[GROUP], e[GROUP];
#pragma unroll
                            for (int g = 0; g < GROUP; ++g) {
                                const int e0 = 2 * (p0 + g), e1 = e0 + 1 < E ? e0 + 1 : e0;
                                if (e0 < E)
                                    z[g] = f32x2{raw[e0 / T::SP][e0 % T::SP], raw[e1 / T::SP][e1 % T::SP]} *
                                               f32x2{scv[e0 / T::SP], scv[e1 / T::SP]} + f32x2{shv[e0 / T::SP], shv[e1 / T::SP]};
                            }
                            if (SILU) {
#pragma unroll
                                for (int g = 0; g < GROUP; ++g) if (2 * (p0 + g) < E) e[g] = z[g] * -1.4426950408889634f;
#pragma unroll
                                for (int g = 0; g < GROUP; ++g)
                                    if (2 * (p0 + g) < E) { e[g][0] = __builtin_amdgcn_exp2f(e[g][0]); e[g][1] = __builtin_amdgcn_exp2f(e[g][1]); }
#pragma unroll
                                for (int g = 0; g < GROUP; ++g) if (2 * (p0 + g) < E) e[g] = e[g] + 1.0f;
#pragma unroll
                                for (int g = 0; g < GROUP; ++g)
                                    if (2 * (p0 + g) < E) { e[g][0] = __builtin_amdgcn_rcpf(e[g][0]); e[g][1] = __builtin_amdgcn_rcpf(e[g][1]); }
#pragma unroll
                                for (int g = 0; g < GROUP; ++g) if (2 * (p0 + g) < E) z[g] = z[g] * e[g];
                            }
#pragma unroll
                            for (int g = 0; g < GROUP; ++g) {
                                const int e0 = 2 * (p0 + g), e1 = e0 + 1;
                                if (e0 < E) dst[(e0 / T::SP) * T::IN_CHP + (e0 % T::SP) * 256] = (!border || in_ok[e0 % T::SP]) ? z[g][0] : 0.0f;
                                if (e1 < E) dst[(e1 / T::SP) * T::IN_CHP + (e1 % T::SP) * 256] = (!border || in_ok[e1 % T::SP]) ? z[g][1] : 0.0f;
                            }
                        }
                    };
                    if (a.act == 2) transform(std::true_type{});
                    else transform(std::false_type{});
                    if (pstamp) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                  const unsigned long long now = __builtin_amdgcn_s_memtime(); p_math += now - p_t; p_t = now; }
                } else {
#pragma unroll
                    for (int c = 0; c < KC; ++c)
#pragma unroll
                        for (int j = 0; j < T::SP; ++j) dst[c * T::IN_CHP + j * 256] = raw[c][j];
                    if (pstamp) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                  const unsigned long long now = __builtin_amdgcn_s_memtime(); p_store += now - p_t; p_t = now; }
                }
                __syncthreads();                           // hand-over: stage (s&1) is complete
            }
        }
        if (pstamp && tid == 0) {
            unsigned long long *d = a.dbg_buf + (size_t)blockIdx.x * 8 + 4;
            d[0] = p_issue; d[1] = p_wait; d[2] = p_math; d[3] = p_store;
        }
        return;
    }

    // =============================================================================== CONSUMERS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 5, l31 = lane & 31;
    constexpr int PBH = T::PBH;
    constexpr int ROWS = (NB - 1) * PBH * STRIDE + KS;          // input rows (per lane row ly) one wave's accumulators touch
    const int lx = l31 % PBW, ly = l31 / PBW;                      // the lane's pixel inside its PBW x PBH block
    constexpr int NP = KC / 2;

    f32x16 acc[MB][NB];
    int sub = 0;
    unsigned long long t_mma = 0, t_epi = 0, t_bar = 0, t_last = 0;
    const bool stamp = IPDM_CONV_STAMPS && (a.dbg & 8) != 0;
    const unsigned long long t_begin = stamp ? __builtin_amdgcn_s_memtime() : 0;
    t_last = t_begin;
    // Epilogue addressing is VALU-free: buffer stores take ONE per-lane byte offset (fixed for the whole kernel) and a
    // scalar offset per (cout, row); anything outside the sample's [Cout][Ho][Wo] block is dropped by the range check.
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int out_plane = a.Ho * a.Wo * (UP2 ? 4 : 1);      // channel stride of the output (up2: 4 parity planes)
    const int lane_off = (lk * 4 * out_plane + ly * a.Wo + lx) * 4;      // lane part: cout half, block row, block column
    // bias (conv bias + time-embedding projection) is added by one MFMA per accumulator after the last K chunk; the MB
    // bias values of the NEXT tile are fetched right after, a whole tile ahead of their use.
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
    float nb[MB];
    auto fetch_bias = [&](int k) __attribute__((always_inline)) {
        const TileId t = decode_tile<T::TH, T::TW, T::BN>(a, tile_of(k));
#pragma unroll
        for (int m = 0; m < MB; ++m) nb[m] = bload(b_rsrc, lk ? OOB : l31 * 4, min(t.co0 + m * 32, a.Cout - 1) * 4);
    };
    if (S > 0) fetch_bias(0);
    for (int s = 0; s < S; ++s) {
        if (a.act) __syncthreads();                // A: chunk s-1 done -> the producers' transform burst may use the SIMD
        __syncthreads();                           // hand-over: stage (s&1) is complete
        if (stamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_bar += now - t_last; t_last = now; }
        const int k = s / nchunks, ch = s - k * nchunks;
        if (ch == 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int q = 0; q < NB; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.0f;
        }
        const float *ib = lds + (s & 1) * T::BUF;
        const float *wb = ib + T::IN_TILE;
        if (IL != MB && ch == 0) sub = decode_tile<T::TH, T::TW, T::BN>(a, tile_of(k)).co0 % T::LW / 32;   // m offset inside the group
        {
            // operands: B = the ROWS x KS input values of one channel pair this wave touches (fetched one pair
            // ahead), A = the MB weight values of one tap (fetched one tap ahead); channels beyond Cin are zero
            // in LDS (producer padding), so every chunk runs the full schedule.
            float b_cur[ROWS][KS], b_nxt[ROWS][KS], a_c[MB], a_n[MB];
            auto read_b = [&](int cp, float (&Bv)[ROWS][KS]) __attribute__((always_inline)) {
                const int c = cp * 2 + lk;
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx)
                        Bv[r][kx] = ib[c * T::IN_CHP + ((wave * NB * PBH + ly) * STRIDE + r) * T::PITCH + lx * STRIDE + kx];
            };
            auto read_a = [&](int cp, int t, float (&A)[MB]) __attribute__((always_inline)) {
                const int c = cp * 2 + lk;
                typedef float fvec __attribute__((ext_vector_type(MB)));
                const fvec v = *reinterpret_cast<const fvec *>(wb + ((c * T::TAPS + t) * 32 + l31) * IL + sub);   // [c][tap][l31][m]
#pragma unroll
                for (int m = 0; m < MB; ++m) A[m] = v[m];
            };
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
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[m], b_cur[q * PBH * STRIDE + t / KS][t % KS], acc[m][q], 0, 0, 0);
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
        }
        if (ch == nchunks - 1) {
            // + bias, added LAST as the reference does (conv, then bias): one more MFMA per accumulator with A = the bias
            // of the lane's cout on the k=0 half and B = 1 -- no per-element VALU add
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int q = 0; q < NB; ++q) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(nb[m], 1.0f, acc[m][q], 0, 0, 0);
            if (k + 1 < n_my) fetch_bias(k + 1);
        }
        if (stamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_mma += now - t_last; t_last = now; }
        if (ch == nchunks - 1) {
            // ---- tile epilogue: (+ residual) -> NCHW stores, 32 consecutive pixels per half-wave; the stores drain
            //      while the next tile is multiplied
            const TileId t = decode_tile<T::TH, T::TW, T::BN>(a, tile_of(k));
            const size_t sample = ((size_t)(UP2 ? 0 : t.ks) * a.B + t.n) * a.Cout * out_plane;
            const int par_off = UP2 ? t.ks * a.Ho * a.Wo * 4 : 0, par_row = UP2 ? t.ks * a.Ho : 0;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + sample), 0, a.Cout * out_plane * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((a.res ? a.res : a.out) + sample), 0, a.Cout * out_plane * 4, 0x00020000);
            // per-lane offset: lane part, or out of range for columns beyond Wo / rows beyond Ho (dropped by the range
            // check); scalar offset: pure adds -- cout of register r is co0 + 32m + (r&3) + 8(r>>2) [+ 4 lk in lane_off]
            const int plane4 = out_plane * 4;
            int voffq[NB], rowq[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int oy = t.oy0 + (swave * NB + q) * PBH;          // first row of the block; the lane adds ly
                voffq[q] = (t.ox0 + lx < a.Wo && oy + ly < a.Ho) ? lane_off : OOB;
                rowq[q] = (min(oy, a.Ho - 1) * a.Wo + t.ox0) * 4 + par_off;     // scalar offsets stay in range; the lanes are killed above
            }
            // one row of per-cout partial sums {sum, sum of squares} of block row q: staged [cout][2] in LDS by the lanes
            // that own a cout, read back as contiguous runs (DS operations of one wave execute in order, so the staging
            // row is reused by the next q) and stored with one instruction; row index = pixel row * tile columns + column
            auto store_stats_row = [&](int q, auto partial_tag) __attribute__((always_inline)) {
                constexpr bool PARTIAL = decltype(partial_tag)::value;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const float *sb = lds + 2 * T::BUF + swave * 256;
                const int oy = t.oy0 + (swave * NB + q) * PBH;
                if (oy < a.Ho && lane * 2 < T::BN) {
                    float *dst = a.stats + (((size_t)t.n * a.stats_rows + (size_t)(par_row + oy) * a.tiles_x + t.ox0 / T::TW) * a.Cout + t.co0) * 2;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(sb + lane * 4);       // couts 2 lane, 2 lane + 1
                    if (!PARTIAL) *reinterpret_cast<f32x4 *>(dst + lane * 4) = v;
                    else {
                        if (t.co0 + 2 * lane < a.Cout) *reinterpret_cast<f32x2 *>(dst + lane * 4) = f32x2{v[0], v[1]};
                        if (t.co0 + 2 * lane + 1 < a.Cout) *reinterpret_cast<f32x2 *>(dst + lane * 4 + 2) = f32x2{v[2], v[3]};
                    }
                }
                __builtin_amdgcn_wave_barrier();
            };
            // PARTIAL: the tile's couts run past Cout (Cout % (32*MB) != 0): those registers are skipped by a scalar
            // branch (their scalar offset would leave the buffer); full tiles carry no such test.
            auto epilogue = [&](auto partial_tag) __attribute__((always_inline)) {
                constexpr bool PARTIAL = decltype(partial_tag)::value;
                auto co_ok = [&](int m, int r) { return !PARTIAL || t.co0 + m * 32 + (r & 3) + 8 * (r >> 2) < a.Cout; };
                if (a.res) {
                    // residual first, as its own phase: VMEM loads and stores retire through one in-order counter, so a
                    // load issued behind stores would wait for those stores to reach memory
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int q = 0; q < NB; ++q) {
                            float rv[16];
                            int so = (t.co0 + m * 32) * plane4 + rowq[q];
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                rv[r] = co_ok(m, r) ? bload(r_rsrc, voffq[q], so) : 0.0f;
                                so += ((r & 3) == 3 ? 5 : 1) * plane4;
                                asm volatile("" : "+s"(so));       // keep ONE running scalar offset (no table of 128 SGPRs)
                            }
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[m][q][r] += rv[r];
                        }
                }
                if (a.stats) {
                    // fused GroupNorm statistics of the output, dword-epilogue form: the lane is a pixel, the register a
                    // cout -> 32-lane sums per register (rare shapes: widths that are not multiples of 4); one row of
                    // partial sums per PIXEL ROW, like the 16-byte epilogue
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        float *sb = lds + 2 * T::BUF + swave * 256;
#pragma unroll
                        for (int m = 0; m < MB; ++m)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = voffq[q] != OOB ? acc[m][q][r] : 0.0f;
                                const float s1 = sum_lanes_half(v), s2 = sum_lanes_half(v * v);
                                if (l31 == 0) *reinterpret_cast<f32x2 *>(sb + (m * 32 + 8 * (r >> 2) + (r & 3) + 4 * lk) * 2) = f32x2{s1, s2};
                            }
                        store_stats_row(q, partial_tag);
                    }
                }
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        int so = (t.co0 + m * 32) * plane4 + rowq[q];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[m][q][r];      // (bit_cast straight from the vector element stores element 0)
                            if (co_ok(m, r)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, voffq[q], so, 0);
                            so += ((r & 3) == 3 ? 5 : 1) * plane4;
                            asm volatile("" : "+s"(so));
                        }
                    }
            };
            // Fast path (rows are a multiple of 4 pixels long, full cout tile): 16 bytes per lane and store.  The
            // accumulator holds, per lane (= pixel) and register group g, 4 consecutive COUTS; a 4x4 transpose inside
            // every quad of lanes (DPP quad_perm + select, 8-16 VALU per block) turns that into 4 consecutive PIXELS of one
            // cout per lane, so one buffer_store_dwordx4 replaces four dword stores (the store path retires ~1 instruction
            // per 100 cycles per wave regardless of its width) and a half-wave still writes 4 full 128-byte row segments.
            if constexpr (VEC4) {       // launcher: Cout % (32*MB) == 0
                const int qi = l31 & 3;
                const int qx = lx >> 2;                                    // quad of 4 consecutive columns inside the block row
                const int lane_off4 = ((qi + 4 * lk) * out_plane + ly * a.Wo + 4 * qx) * 4;
                const bool xok = t.ox0 + 4 * qx + 4 <= a.Wo;
                // widths that are not multiples of 4 (250, 125, 63 ... of the projection domain): the run of 4 pixels that
                // straddles the right edge of the image is stored element by element by its lane (ragged: the tile touches
                // the edge, wave-uniform); every other run takes the 16-byte path, at dword alignment
                const bool ragged = t.ox0 + 32 > a.Wo && (a.Wo & 3) != 0;
                const int nval = a.Wo - (t.ox0 + 4 * qx);                   // valid pixels of the lane's run (1..3 when partial)
                int voff4[NB];
                bool part[NB];
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const bool rok = t.oy0 + (swave * NB + q) * PBH + ly < a.Ho;
                    voff4[q] = (xok && rok) ? lane_off4 : OOB;
                    part[q] = ragged && !xok && nval > 0 && rok;
                }
                const bool odd = (l31 & 1) != 0, hi = (l31 & 2) != 0;
#define IPDM_XCHG(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))
                auto block = [&](int m, int q, int g) __attribute__((always_inline)) -> f32x4 {
                    float r0 = acc[m][q][4 * g], r1 = acc[m][q][4 * g + 1], r2 = acc[m][q][4 * g + 2], r3 = acc[m][q][4 * g + 3];
                    // lanes i^1 exchange registers j^1 (quad_perm [1,0,3,2]), then lanes i^2 exchange registers j^2 ([2,3,0,1])
                    float x = IPDM_XCHG(r0, 0xB1), y = IPDM_XCHG(r1, 0xB1);
                    r0 = odd ? y : r0; r1 = odd ? r1 : x;
                    x = IPDM_XCHG(r2, 0xB1); y = IPDM_XCHG(r3, 0xB1);
                    r2 = odd ? y : r2; r3 = odd ? r3 : x;
                    x = IPDM_XCHG(r0, 0x4E); y = IPDM_XCHG(r2, 0x4E);
                    r0 = hi ? y : r0; r2 = hi ? r2 : x;
                    x = IPDM_XCHG(r1, 0x4E); y = IPDM_XCHG(r3, 0x4E);
                    r1 = hi ? y : r1; r3 = hi ? r3 : x;
                    return f32x4{r0, r1, r2, r3};
                };
#undef IPDM_XCHG
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                // fused GroupNorm statistics of the output: after the quad transpose a lane holds 4 pixels of ONE cout
                // (32m + 8g + (l31 & 3) + 4 lk), so the per-cout sums are formed in-lane over the 4 pixels and only 8 lanes
                // (l31 >> 2) remain to be combined.  One row of partial sums per PIXEL ROW and tile column -- the same
                // rows in the same order whatever tile shape the launcher picked (it depends on the batch size), so the
                // statistics, and with them every later value, do not depend on how slices are batched.
                // residual: 16-byte loads, one (row block, cout block) AHEAD of its use.  VMEM operations retire through one
                // in-order counter: a load issued behind the previous block's stores would wait for those stores to reach
                // memory, and its own latency would be exposed once per block (8 times per tile).
                f32x4 rv[2][4];
                auto load_res = [&](int q, int m, f32x4 (&dst)[4]) __attribute__((always_inline)) {
                    const int so = (t.co0 + m * 32) * plane4 + rowq[q];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        dst[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, voff4[q], so + 8 * g * plane4, 0));
                };
                if (a.res) load_res(0, 0, rv[0]);
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    float st1[MB][4], st2[MB][4];
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const int i = q * MB + m;
                        const int so = (t.co0 + m * 32) * plane4 + rowq[q];
                        if (a.res && i + 1 < NB * MB) load_res((i + 1) / MB, (i + 1) % MB, rv[(i + 1) & 1]);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 v = block(m, q, g);
                            if (a.res) v += rv[i & 1][g];
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, voff4[q], so + 8 * g * plane4, 0);
                            if (ragged) {            // wave-uniform; the partial run's residual was not loaded above (its offset is OOB)
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const int vo = (part[q] && j < nval) ? lane_off4 + 4 * j : OOB;
                                    float e = v[j];
                                    if (a.res) e += bload(r_rsrc, vo, so + 8 * g * plane4);
                                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e), o_rsrc, vo, so + 8 * g * plane4, 0);
                                    if (part[q]) v[j] = j < nval ? e : 0.0f;
                                }
                            }
                            if (a.stats) {
                                const bool ok = voff4[q] != OOB || (ragged && part[q]);      // (a partial run has its tail zeroed above)
                                st1[m][g] = ok ? (v[0] + v[1]) + (v[2] + v[3]) : 0.0f;
                                st2[m][g] = ok ? fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0]))) : 0.0f;
                            }
                        }
                    }
                    if (a.stats) {
                        float *sb = lds + 2 * T::BUF + swave * 256;
#pragma unroll
                        for (int m = 0; m < MB; ++m)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const float s1 = sum_lanes_stride4(st1[m][g]), s2 = sum_lanes_stride4(st2[m][g]);
                                if ((l31 >> 2) == 0) *reinterpret_cast<f32x2 *>(sb + (m * 32 + 8 * g + qi + 4 * lk) * 2) = f32x2{s1, s2};
                            }
                        store_stats_row(q, std::false_type{});
                    }
                }
            } else {
                if (t.co0 + T::BN <= a.Cout) epilogue(std::false_type{});
                else epilogue(std::true_type{});
            }
        }
        if (stamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_epi += now - t_last; t_last = now; }
    }
    if (stamp && tid == 0) {
        unsigned long long *d = a.dbg_buf + (size_t)blockIdx.x * 8;
        d[0] = t_mma; d[1] = t_epi; d[2] = t_bar; d[3] = __builtin_amdgcn_s_memtime() - t_begin;
    }
}

int num_cus() { return device_cu_count(); }

template <int KS, int STRIDE, int MB, int NB, int KC, int IL, int PBW, bool VEC4, bool PLANAR = false>
int launch_ws_v(const ConvArgs &args, hipStream_t st, int prof_cls)
{
    using T = WsTile<KS, STRIDE, MB, NB, KC, IL, PBW>;
    constexpr size_t LDS_TOTAL = ws_lds_total<T>();      // stages + statistics rows
    static_assert(LDS_TOTAL <= 160 * 1024, "conv_ws: LDS stages exceed 160 KiB");
    ConvArgs a = args;
    // IPDM_CONV_DBG=8: in-kernel s_memtime stamps per phase (tools/bench_conv_dbg.py; needs a.dbg_buf, bench entry only)
    const int dbg = a.dbg_buf ? opt(OPT_CONV_DBG) : 0;
    a.dbg = a.dbg_buf ? (dbg & 8) : 0;
    a.tiles_x = cdiv(a.Wo, T::TW);
    a.tiles_y = cdiv(a.Ho, T::TH);
    a.co_tiles = cdiv(a.Cout, T::BN);
    IPDM_REQUIRE(a.C2 == 0 || a.C1 % KC == 0, "conv2d: concat split %d not a multiple of the K chunk %d", a.C1, KC);
    IPDM_REQUIRE((KS == 2) == (a.up2 != 0), "conv2d: 2x2 taps are the up-sampling convolution's");
    IPDM_REQUIRE(PLANAR == (a.x1_planar != 0), "conv2d: this kernel variant does not read parity-planar inputs");
    IPDM_REQUIRE(!a.x1_planar || (!a.upsample && !(a.Hs & 1) && !(a.Ws & 1)), "conv2d: parity-planar input of odd size %dx%d", a.Hs, a.Ws);
    IPDM_REQUIRE((long)a.C1 * a.Hs * a.Ws < (1L << 29) && (long)(a.C2 + 1) * a.Hs * a.Ws < (1L << 29) &&
                     (long)a.Cout * a.Ho * a.Wo * (a.up2 ? 4 : 1) < (1L << 29) &&
                     (long)((a.C1 + a.C2 + KC - 1) / KC * KC) * T::TAPS * a.cout_pad * (a.up2 ? 4 : 1) < (1L << 29),
                 "conv2d: per-sample tensor exceeds the 2 GiB buffer-addressing range");
    if (a.ksplit < 1) a.ksplit = 1;
    if (a.up2) {
        IPDM_REQUIRE(!a.res && !a.C2 && !a.upsample && a.H == a.Ho && a.W == a.Wo, "conv2d: bad up-sampling convolution");
        a.ksplit = 4;       // the parity is the tile's `ks` digit
    } else
        IPDM_REQUIRE(((a.C1 + a.C2 + KC - 1) / KC) % a.ksplit == 0 && (a.ksplit == 1 || (!a.bias && !a.res && !a.stats)),
                     "conv2d: bad K split %d", a.ksplit);
    const long ntiles = (long)a.tiles_x * a.tiles_y * a.co_tiles * a.B * a.ksplit;
    IPDM_REQUIRE(ntiles < (1L << 31), "conv2d: too many tiles");
    const int cus = num_cus();
    int G = (int)(ntiles < cus ? ntiles : cus);
    G = (G + 7) / 8 * 8;
    IPDM_REQUIRE(!a.stats || a.stats_rows == a.tiles_x * a.Ho * (a.up2 ? 4 : 1), "conv2d: statistics rows %d != %d", a.stats_rows,
                 a.tiles_x * a.Ho * (a.up2 ? 4 : 1));
    if (int rc = ensure_dynamic_lds((const void *)conv_ws_kernel<KS, STRIDE, MB, NB, KC, IL, PBW, VEC4, PLANAR>, LDS_TOTAL)) return rc;
    const bool prof = prof_enabled();
    if (prof) prof_before(prof_cls, st);
    hipLaunchKernelGGL((conv_ws_kernel<KS, STRIDE, MB, NB, KC, IL, PBW, VEC4, PLANAR>), dim3((unsigned)G), dim3(512), LDS_TOTAL, st, a, (int)ntiles);
    // (up2: the four parity convolutions executed, 16 multiply-adds per source pixel -- not the 36 of the 3x3 form)
    if (prof) prof_after(prof_cls, 2.0 * a.B * a.Ho * a.Wo * (double)a.Cout * (a.C1 + a.C2) * KS * KS * (a.up2 ? 4 : 1), st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

// the 16-byte-store epilogue needs whole cout tiles (a ragged right edge is handled inside it); everything else takes the
// dword epilogue (with its ragged-cout variant).  Two kernels instead of one keep either epilogue out of the other's
// register allocation.
template <int KS, int STRIDE, int MB, int NB, int KC, int IL = MB, int PBW = 32>
int launch_ws(const ConvArgs &a, hipStream_t st, int prof_cls)
{
    const bool strict = opt(OPT_CONV_VEC4_STRICT) != 0;     // A/B: rows of whole 4-pixel runs only, as in round 1
    if ((!strict || (a.Wo & 3) == 0) && a.Cout % (32 * MB) == 0) {
        if constexpr (STRIDE == 1 && KS != 2) {       // readers of an up2 convolution's parity-planar output (conv_ws_planar_ok)
            if (a.x1_planar) return launch_ws_v<KS, STRIDE, MB, NB, KC, IL, PBW, true, true>(a, st, prof_cls);
        }
        return launch_ws_v<KS, STRIDE, MB, NB, KC, IL, PBW, true>(a, st, prof_cls);
    }
    return launch_ws_v<KS, STRIDE, MB, NB, KC, IL, PBW, false>(a, st, prof_cls);
}

}  // namespace

namespace ipdm {

// ---- K split: which layers, how many slices
// The number of slices depends on the layer alone (never on the batch size): a slice of a batch must stay bit-equal to
// the same slice sampled alone or in another shard, and a different K split is a different summation order.  So only
// layers that cannot fill the chip even at 8 slices per GPU are split: at most 16 of the 8x32-pixel x 128-cout tiles per
// SAMPLE (256 channels at 32x32, 63x29, 32x15: 8-16 tiles).
int conv_ws_split(const ConvArgs &a)
{
    const bool off = opt(OPT_CONV_NO_SPLITK) != 0;
    if (off || !a.w_interleave || a.w_interleave > 4 || conv_up2_eligible(a)) return 1;
    const long per_sample = (long)cdiv(a.Wo, 32) * cdiv(a.Ho, 8) * cdiv(a.Cout, 128);
    if (per_sample > 16) return 1;
    const int nch = cdiv(a.C1 + a.C2, a.ksize == 1 ? 32 : 8);
    const int want = per_sample <= 8 ? 8 : 4;
    int best = 1;
    for (int S = 2; S <= want; ++S)
        if (nch % S == 0 && nch / S >= 2) best = S;
    return best;
}

// which launches of this file can read x1 parity-planar: the stride-1 kernels with whole cout tiles (launch_ws)
bool conv_ws_planar_ok(const ConvArgs &a)
{
    const bool strict = opt(OPT_CONV_VEC4_STRICT) != 0;
    return (a.w_interleave == 2 || a.w_interleave == 4) && a.stride == 1 && (a.ksize == 1 || a.ksize == 3) &&
           a.Cout % (32 * a.w_interleave) == 0 && (!strict || (a.Wo & 3) == 0) && !conv_up2_eligible(a);
}

// the up-sampling convolution as four parity convolutions (ConvArgs::w_up2): exact 2x nearest, wide layers, no prologue
bool conv_up2_eligible(const ConvArgs &a)
{
    const bool off = opt(OPT_CONV_NO_UP2) != 0;      // (read per call: bench.py times both forms in one process)
    return !off && a.w_up2 && (a.w_interleave == 2 || a.w_interleave == 4) && a.ksize == 3 && a.stride == 1 && a.C2 == 0 &&
           a.act == 0 && !a.res && a.H == 2 * a.Hs && a.W == 2 * a.Ws && a.Ho == a.H && a.Wo == a.W;
}

int conv_ws_stats_rows(const ConvArgs &a)
{
    if (conv_up2_eligible(a)) return 4 * a.Hs * cdiv(a.Ws, 32);
    if (a.split_ws && conv_split(a) > 1) return cdiv((long)a.Ho * a.Wo, SPLIT_PIX);      // the combine pass writes them (either kernel)
    return a.Ho * cdiv(a.Wo, 32);      // one row per pixel row and 32-pixel tile column: independent of the tile variant
}

namespace {
// out = sum over the K slices (ascending: a fixed order) + bias (+ residual); the GroupNorm statistics of the result as
// one row of per-channel partial sums per SPLIT_PIX pixels.  Block = (pixel chunk, cout, sample).
__global__ void __launch_bounds__(256) splitk_combine_kernel(const float *__restrict__ ws, int S, const float *__restrict__ bias,
                                                             const float *__restrict__ res, float *__restrict__ out,
                                                             float *__restrict__ stats, int rows, int B, int Cout, int HW)
{
    const int p = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
    const size_t plane = ((size_t)n * Cout + c) * HW, slice = (size_t)B * Cout * HW;
    const float b = bias ? bias[c] : 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
    const int end = min(HW, (p + 1) * SPLIT_PIX);
    for (int i = p * SPLIT_PIX + threadIdx.x; i < end; i += 256) {
        float v = ws[plane + i];
        for (int s = 1; s < S; ++s) v += ws[(size_t)s * slice + plane + i];
        v += b;                                     // conv, then bias, then the residual: the order of the unsplit kernels
        if (res) v += res[plane + i];
        out[plane + i] = v;
        s1 += v;
        s2 = fmaf(v, v, s2);
    }
    if (!stats) return;
    __shared__ float r1[4], r2[4];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float *d = stats + (((size_t)n * rows + p) * Cout + c) * 2;
        d[0] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
        d[1] = (r2[0] + r2[1]) + (r2[2] + r2[3]);
    }
}
}  // namespace

static int conv2d_ws_dispatch(const ConvArgs &a, hipStream_t st);

// 3x3 stride-1 convolutions with more than 32 output channels (weights packed cout-interleaved, see
// conv_weight_interleave / conv_pack_weights).
int conv2d_ws_launch(const ConvArgs &a, hipStream_t st)
{
    if (conv_up2_eligible(a)) {
        if (conv_wup2_eligible(a)) return conv2d_wup2_launch(a, st, 7);      // the F(2x2,2x2) form of the same four convolutions
        ConvArgs k = a;
        k.up2 = 1; k.w = a.w_up2; k.ksize = 2; k.upsample = 0; k.H = k.Ho = a.Hs; k.W = k.Wo = a.Ws; k.split_ws = nullptr; k.ksplit = 1;
        // (16-channel chunks -- 16k instead of 8k cycles of MFMA per hand-over -- measured 0.7 % slower per forward)
        if (a.w_interleave == 4) return launch_ws<2, 1, 4, 2, 8>(k, st, 1);
        return launch_ws<2, 1, 2, 4, 8>(k, st, 1);
    }
    const bool wino = conv_wino_eligible(a);                                 // the Winograd-domain form (conv_wino.hip / conv_wino2.hip)
    const int S = a.split_ws ? conv_split(a) : 1;
    if (S == 1) return wino ? conv2d_wino_launch(a, st) : conv2d_ws_dispatch(a, st);
    ConvArgs k = a;
    k.out = a.split_ws; k.bias = nullptr; k.res = nullptr; k.stats = nullptr; k.stats_rows = 0; k.ksplit = S;
    if (int rc = wino ? conv2d_wino_launch(k, st) : conv2d_ws_dispatch(k, st)) return rc;
    const int HW = a.Ho * a.Wo, rows = cdiv(HW, SPLIT_PIX);
    IPDM_REQUIRE(!a.stats || a.stats_rows == rows, "conv2d: statistics rows %d != %d (K split)", a.stats_rows, rows);
    hipLaunchKernelGGL(splitk_combine_kernel, dim3(rows, a.Cout, a.B), dim3(256), 0, st, a.split_ws, S, a.bias, a.res, a.out,
                       a.stats, rows, a.B, a.Cout, HW);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

static int conv2d_ws_dispatch(const ConvArgs &a, hipStream_t st)
{
    if (a.ksize == 3 && a.stride == 1 && a.w_interleave == 4) {
        // layers whose 8x32x128 tiling gives fewer tiles than CUs (32x32 and 63x29 at 256 channels) use 4x32x64 tiles
        // over the same packed weights: 4x the workgroups, each reading its half of the 128-cout interleave group
        const long tiles = (long)cdiv(a.Wo, 32) * cdiv(a.Ho, 8) * cdiv(a.Cout, 128) * a.B;
        // (a K-split launch has ksplit times as many schedule entries: at 8 slices per GPU the 32x32 / 63x29 layers fill the
        //  chip with full-size tiles, 10 % faster than on quarter tiles; the choice does not change any result)
        if (tiles * (a.ksplit > 1 ? a.ksplit : 1) < 160) return launch_ws<3, 1, 2, 1, 8, 4>(a, st, 0);
        // (8x4-pixel MFMA blocks -- PBW = 8, tile 32 rows x 8 cols -- pad the 228/114-wide layers 8 % less but measured
        // 3-9 % SLOWER: 32-byte row pieces in every load and store, 18 instead of 12 operand reads per channel pair)
        return launch_ws<3, 1, 4, 2, 8>(a, st, 0);
    }
    if (a.ksize == 3 && a.stride == 1 && a.w_interleave == 2) return launch_ws<3, 1, 2, 4, 8>(a, st, 0);
    // Downsample (3x3 stride 2): same kernel, input tile (2*TH+1) x 65 per channel
    if (a.ksize == 3 && a.stride == 2 && a.w_interleave == 4) {
        // (under-filled launches on quarter tiles, as for the 1x1 and stride-1 kernels: one slice alone at 250x114 -> 125x57)
        const long tiles = (long)cdiv(a.Wo, 32) * cdiv(a.Ho, 8) * cdiv(a.Cout, 128) * a.B;
        if (tiles < 128 && !opt(OPT_CONV1X1_NO_QUARTER)) return launch_ws<3, 2, 2, 1, 8, 4>(a, st, 1);
        return launch_ws<3, 2, 4, 2, 8>(a, st, 1);
    }
    if (a.ksize == 3 && a.stride == 2 && a.w_interleave == 2) return launch_ws<3, 2, 2, 2, 8>(a, st, 1);
    // 1x1: a plain GEMM over channels; 32-channel chunks give the producers 8k cycles of MFMA per hand-over
    if (a.ksize == 1 && a.stride == 1) {
        // launches that leave most of the chip idle (one slice alone on the low-resolution levels: 64 tiles of 8x32x128 for
        // 256 channels at 125x57) run on quarter tiles (4x32 pixels x 64 couts) over the same packed weights: 4x the
        // workgroups, the same K order per output -- the choice looks at the batch and does not change a bit, like the 3x3
        // variant above.  B = 1: 256->256 @125x57 0.059 -> see NOTEBOOK.md (B = 1 latency)
        const int rows = a.w_interleave == 4 ? 8 : 16;
        const long tiles = (long)cdiv(a.Wo, 32) * cdiv(a.Ho, rows) * cdiv(a.Cout, 32 * a.w_interleave) * a.B * (a.ksplit > 1 ? a.ksplit : 1);
        if (tiles < 128 && !opt(OPT_CONV1X1_NO_QUARTER)) {
            if (a.w_interleave == 4) return launch_ws<1, 1, 2, 1, 32, 4>(a, st, 1);
            return launch_ws<1, 1, 2, 1, 32, 2>(a, st, 1);
        }
        if (a.w_interleave == 4) return launch_ws<1, 1, 4, 2, 32>(a, st, 1);
        return launch_ws<1, 1, 2, 4, 32>(a, st, 1);
    }
    set_error("conv2d_ws: unsupported ksize=%d stride=%d interleave=%d", a.ksize, a.stride, a.w_interleave);
    return IPDM_ERR_UNSUPPORTED;
}

}  // namespace ipdm
