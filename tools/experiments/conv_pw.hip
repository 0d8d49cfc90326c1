// EXPERIMENT, NOT BUILT (round 4, shelved): kept for the record of DESIGN.md section 6 "negative results".
//   * timing (tools/pw_check.py bench at the time, B = 8): 0.74-1.11x conv_ws.hip's 1x1 path -- no gain.  Knock-outs: the bare
//     MFMA + operand-read loop 0.73 of the f32 peak (672 items on 256 workgroups: 3 rounds for 2.6 rounds of work), pixel
//     staging -16 %, stores -8 %, the chunk barrier -4...8 %.
//   * parity: a run-dependent handful of wrong outputs (exact zeros in accumulator register 0 of lanes 12-15 / 28-31 of a
//     wave) that neither wait states nor a full vmcnt(0) at the end of the item removed: unresolved, so it never shipped.
// 1x1 convolutions (pointwise: qkv / proj_out of the AttentionBlocks, the ResidualBlocks' channel-changing shortcuts;
// Model/model.py:116-119,142-155) as a plain GEMM over channels on the exact-f32 MFMA (gfx950) -- round 4.
//
//     out[n][co][p] = sum_c W[co][c] * act(x[n][c][p]) + bias[co] (+ res[n][co][p]),   p = the H*W pixels of a plane, flattened
//
// Structure: the one conv_wino2.hip arrived at (v_mfma_f32_32x32x2_f32 shares its SIMD's vector ALU, so nothing may sit in
// a wave that only stages, and the staging work per MFMA has to be small), which for a pointwise operator becomes simple:
//
//   * one 512-thread workgroup per CU, persistent; tile = 64 RPW FLAT pixels x 128 couts (RPW = 8: 512 pixels; 4 / 2 for
//     launches that would leave the chip under-filled -- the accumulation order of an output and the statistics rows do not
//     depend on RPW, so that choice may look at the batch).  A pointwise operator does not care about image rows: no ragged
//     right edges, no padding of 57- or 228-pixel rows to multiples of 32 (the 2-D tiles of conv_ws.hip waste 8-12 % there);
//   * all eight waves multiply: wave w owns RPW rows of 32 pixels (row half w & 1) x cout quarter (w >> 1): RPW accumulators
//     of 32x32.  The operand roles are SWAPPED against conv_ws.hip -- pixels on M, couts on N -- so four consecutive
//     accumulator registers are four consecutive pixels of one cout: 16-byte stores and residual loads with no transposes,
//     and the per-cout statistics are in-lane sums plus ONE exchange with lane ^ 32;
//   * all eight waves stage: wave w loads channels 8 (w >> 1) + 2 kp + (w & 1), kp = 0..3, of each 32-channel chunk for all
//     pixels of the tile -- one dword per (pixel, channel), a wave-load is 256 contiguous bytes, addressing is a scalar
//     offset -- applies GroupNorm(+SiLU) where the layer has one (the qkv projection: scale/shift as SCALAR operands), and
//     writes the four channels of a pixel as one 16-byte LDS store: the image [row][c8][lk][pixel 32][kp 4] the pixel operand
//     is read from with one ds_read_b128 per four k steps.  Loads are issued a whole chunk ahead;
//   * the weights never touch LDS: packed [c8][cout 32-tile][lk][cout 32][kp 4] (conv_pack_weights_pw), a lane's four k
//     steps are 16 bytes, loaded from L2 into the MFMA's B registers one chunk ahead, reloaded in place;
//   * one barrier per 32-channel chunk (16 RPW MFMAs per wave), LDS stage double-buffered.
// Fused GroupNorm statistics of the output: one row of per-cout {sum, sum of squares} per 32 flat pixels.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef IPDM_PW_KO
#define IPDM_PW_KO 0                // compile-time timing knock-outs (tools/build_variants.sh; WRONG results): 1 no pixel loads / staging after
#endif                              // the first chunk, 2 no weight loads, 4 no stores / residual, 8 no barrier

namespace {

constexpr int KC = 32;                                 // channels per staged chunk: four groups of 8 (4 MFMA k steps of 2)
constexpr int BN = 128;                                // couts per tile
constexpr int ROW_FLOATS = 4 * 2 * 32 * 4;             // one row of 32 pixels, 32 channels: [c8 4][lk 2][pixel 32][kp 4]
constexpr int W_BLOCK_FLOATS = 2 * 32 * 4;             // packed weights of one (8-channel group, 32-cout tile)
constexpr int OOB = 0x7fffffff;

template <int RPW> constexpr size_t lds_bytes() { return (size_t)2 * (2 * RPW) * ROW_FLOATS * sizeof(float); }

struct Item { int n, p0, co0; };

template <int RPW>
__device__ inline Item decode_item(const ConvArgs &a, int item)
{
    Item t;
    const int co_t = item % a.co_tiles;                // the cout tiles of one pixel tile are neighbours: they share its input in L2
    const int rest = item / a.co_tiles;
    t.p0 = (rest % a.tiles_x) * (64 * RPW);
    t.n = rest / a.tiles_x;
    t.co0 = co_t * BN;
    return t;
}

template <int RPW>
__global__ void __launch_bounds__(512) conv_pw_kernel(ConvArgs a, int nitems)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int STAGE = 2 * RPW * ROW_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 5, l31 = lane & 31;

    // static schedule (conv_ws.hip): the workgroups of one XCD take a contiguous run of items, slot rotated per round
    const int G = gridDim.x, per = G >> 3;
    const int local = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int rounds = (nitems + G - 1) / G;
    auto item_of = [&](int k) { return k * G + (local + 5 * k) % G; };
    const int n_my = rounds == 0 ? 0 : (item_of(rounds - 1) < nitems ? rounds : rounds - 1);
    const int Ctot = a.C1 + a.C2;
    const int nchunks = (Ctot + KC - 1) / KC;            // (weights zero-padded to whole chunks; launcher: C1 % KC == 0 with a concat)
    const int S = n_my * nchunks;
    const int HW = a.Ho * a.Wo;
    if (S == 0) return;

    // ---------------------------------------------------------------- staging role
    const int c8w = swave >> 1, lkw = swave & 1;          // the wave's four channels of a chunk: 8 c8w + 2 kp + lkw
    float xs[RPW][4];                                    // [pixel lane + 64 j][kp]: in flight for one chunk
    int g_n = 0, g_p0 = 0, g_co = 0;                     // the item whose chunks are being LOADED
    auto describe = [&](int k) __attribute__((always_inline)) {
        const Item t = decode_item<RPW>(a, item_of(k));
        g_n = t.n; g_p0 = t.p0; g_co = t.co0;
    };
    float s_sc[4], s_sh[4];                              // GroupNorm scale / shift of the chunk in flight
    // (no prologue: descriptors with zero records -- the loads are dropped by the range check, still counted, unconditional)
    const __amdgpu_buffer_rsrc_t gsc_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.act ? a.gn_scale : a.out), 0, a.act ? a.B * Ctot * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t gsh_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.act ? a.gn_shift : a.out), 0, a.act ? a.B * Ctot * 4 : 0, 0x00020000);
    auto issue_x = [&](int ch) __attribute__((always_inline)) {
        const int c0 = ch * KC;
        const bool from1 = c0 < a.C1;
        const float *src = from1 ? a.x1 + (size_t)g_n * a.C1 * HW : a.x2 + (size_t)g_n * a.C2 * HW;
        const int Cs = from1 ? a.C1 : a.C2;
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, Cs * HW * 4, 0x00020000);
        const int cl = (from1 ? c0 : c0 - a.C1) + 8 * c8w + lkw;      // + 2 kp: channel inside its source
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) {
            // (channels past the source -- the zero-padded tail of the last chunk -- are killed through the per-lane offset:
            //  the scalar offset must stay inside the buffer)
            const bool cok = cl + 2 * kp < Cs;
            const int so = ((cok ? cl + 2 * kp : 0) * HW + g_p0) * 4;
#pragma unroll
            for (int j = 0; j < RPW; ++j)
                xs[j][kp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, cok ? lane * 4 : OOB, so + j * 256, 0));
            // (GroupNorm scale / shift of the channel: the same dword for every lane; loaded like the pixels so that they ride
            //  the same in-order counter -- as a scalar load the compiler went through the vector path plus v_readfirstlane
            //  behind a full vmcnt(0) wait at the head of every chunk)
            const int gso = (g_n * Ctot + min(c0 + 8 * c8w + lkw + 2 * kp, Ctot - 1)) * 4;
            s_sc[kp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gsc_rsrc, 0, gso, 0));
            s_sh[kp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gsh_rsrc, 0, gso, 0));
        }
    };
    // (pixels past the plane -- the tail of its last tile -- read the next channel's pixels or, at the very end of the
    //  tensor, zeros: finite values that only reach accumulator rows the epilogue never stores or counts)
    auto stage_x = [&](int par) __attribute__((always_inline)) {
        float *dst = lds + par * STAGE + (c8w * 2 + lkw) * 128 + l31 * 4 + lk * ROW_FLOATS;      // pixel lane + 64 j: row 2 j + lk
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            f32x4 v = {xs[j][0], xs[j][1], xs[j][2], xs[j][3]};
            if (a.act) {
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) v[kp] = fmaf(v[kp], s_sc[kp], s_sh[kp]);
                if (a.act == 2) {
#pragma unroll
                    for (int kp = 0; kp < 4; ++kp) v[kp] = v[kp] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[kp] * -1.4426950408889634f));
                }
            }
            *reinterpret_cast<f32x4 *>(dst + j * 2 * ROW_FLOATS) = v;
        }
    };

    // ---------------------------------------------------------------- multiplying role
    const int rh = swave & 1, hq = swave >> 1;            // which RPW rows, which cout quarter
    f32x16 acc[RPW];
    f32x4 ub[4];                                          // the weights (B operand) of the chunk: four k steps per 8-channel group
    const int co_tiles32 = a.Cout / 32;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, nchunks * 4 * co_tiles32 * W_BLOCK_FLOATS * 4, 0x00020000);
    const int u_voff = (lk * 32 + l31) * 16;
    int w_co = 0;
    auto issue_w = [&](int c8, int ch) __attribute__((always_inline)) {
        const int soff = ((ch * 4 + c8) * co_tiles32 + (w_co >> 5) + hq) * (W_BLOCK_FLOATS * 4);
        ub[c8] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, u_voff, soff, 0));
    };
    const int a_off = rh * RPW * ROW_FLOATS + (lk * 32 + l31) * 4;      // + r * ROW_FLOATS + c8 * 256 (+ stage)
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
    float nb = 0.0f;
    auto fetch_bias = [&](int co0) __attribute__((always_inline)) {
        nb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, lk ? OOB : l31 * 4, (co0 + hq * 32) * 4, 0));
    };

    // ---------------------------------------------------------------- prologue: item 0, chunk 0 staged, chunk 1 in flight
    describe(0);
    w_co = g_co;
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) issue_w(c8, 0);
    issue_x(0);
    fetch_bias(g_co);
    stage_x(0);
    issue_x(1);                                            // (launcher: at least two chunks)
    __syncthreads();

    int s = 0, k = 0;
    Item cur = {g_n, g_p0, g_co};
    // One chunk: multiply chunk s from stage s & 1, then stage chunk s + 1 (its loads were issued a chunk ago) into the other
    // stage and issue the loads of chunk s + 2; the first chunk of an item STARTS its accumulators.
    // (one flat chunk loop, the accumulators cleared at an item's first chunk: with that chunk peeled off -- C = 0 in its
    //  first MFMAs, as conv_wino2.hip does -- this compiler kept two copies of the accumulators and moved all of them at the
    //  head of every chunk)
    int ch = 0;
    for (; k < n_my;) {
        const int ch1 = ch + 1 == nchunks ? 0 : ch + 1, ch2 = ch1 + 1 == nchunks ? 0 : ch1 + 1;
        const float *stage = lds + (s & 1) * STAGE + a_off;
        if (ch == 0) {
            asm volatile("" ::: "memory");                  // (a real branch: if-converted, this was 16 RPW selects in EVERY chunk)
#pragma unroll
            for (int r = 0; r < RPW; ++r) acc[r] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (ch1 == 0) w_co = g_co;                          // the weights loaded from here on belong to the item described last
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) {
            f32x4 a_c = *reinterpret_cast<const f32x4 *>(stage + c8 * 256), a_n;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                if (r + 1 < RPW) a_n = *reinterpret_cast<const f32x4 *>(stage + (r + 1) * ROW_FLOATS + c8 * 256);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[q], ub[c8][q], acc[r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (r + 1 < RPW) a_c = a_n;
            }
            if (!(IPDM_PW_KO & 2)) issue_w(c8, ch1);        // this group's weights of the NEXT chunk, into the registers just read
        }
        if (!(IPDM_PW_KO & 1) && s + 1 < S) stage_x((s + 1) & 1);      // chunk s + 1 (that stage was last read by chunk s - 1)
        // the loads of chunk s + 2: its item is described two chunks before its first MFMA
        if (ch == nchunks - 2 && k + 1 < n_my) describe(k + 1);
        if (!(IPDM_PW_KO & 1)) issue_x(ch2);
        if (!(IPDM_PW_KO & 8)) __syncthreads();            // stage (s + 1) & 1 complete; every wave is done with stage s & 1
        ++s;
        ch = ch1;
        if (ch != 0) continue;
        // ---------------------------------------------------------------- item epilogue
        // + bias: one MFMA per accumulator (pixel operand 1 on k step 0, weight operand = the bias there)
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(lk ? 0.0f : 1.0f, nb, acc[r], 0, 0, 0);
#ifdef PW_NOP_AFTER_BIAS                // (pw_repro.hip bisect: 64 wait states between the bias MFMAs and the first read of their results)
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#endif
        const Item t = cur;
        cur = Item{g_n, g_p0, g_co};                         // (describe(k + 1) ran two chunks ago)
        if (k + 1 < n_my) fetch_bias(cur.co0);
        const size_t sample = (size_t)t.n * a.Cout * HW;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + sample), 0, a.Cout * HW * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((a.res ? a.res : a.out) + sample), 0,
                                                                                   a.res ? a.Cout * HW * 4 : 0, 0x00020000);
        // lane: cout co0 + 32 hq + l31; register 4 q + e of accumulator r: pixel p0 + 32 (rh RPW + r) + 8 q + 4 lk + e
        const int lane_off = (l31 * HW + 4 * lk) * 4;
        const int so0 = ((t.co0 + hq * 32) * HW + t.p0 + rh * RPW * 32) * 4;
        const int pw0 = t.p0 + rh * RPW * 32 + 4 * lk;     // the lane's first pixel of row 0, run 0
        const bool tail = t.p0 + 64 * RPW > HW;            // (uniform) the plane ends inside this tile
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        // residual: 16-byte loads one ROW ahead of their use, each register set reloaded in place right after its add
        f32x4 rv[4];
        auto load_res = [&](int r, int q) __attribute__((always_inline)) {
            const int pq = pw0 + 32 * r + 8 * q;
            rv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, (!tail || pq + 3 < HW) ? lane_off : OOB, so0 + (32 * r + 8 * q) * 4, 0));
        };
        if (a.res) {
#pragma unroll
            for (int q = 0; q < 4; ++q) load_res(0, q);
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[r][4 * q], acc[r][4 * q + 1], acc[r][4 * q + 2], acc[r][4 * q + 3]};
#ifdef PW_OPAQUE_COPY                   // (pw_repro.hip bisect: the stored values pass through an opaque VALU-visible copy)
                asm volatile("" : "+v"(v));
#endif
                const int pq = pw0 + 32 * r + 8 * q;
                const int so = so0 + (32 * r + 8 * q) * 4;
                if (a.res) {
                    v += rv[q];
                    if (r + 1 < RPW) load_res(r + 1, q);
                }
                if (!tail) {
                    if (!(IPDM_PW_KO & 4) || pq == 12345) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, lane_off, so, 0);
#ifdef PW_DRAIN_AFTER_STORE             // (pw_repro.hip bisect: every store has completed before anything else happens)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifdef PW_NOP_AFTER_STORE               // (pw_repro.hip bisect: PW_NOP_AFTER_STORE + 1 wait states between a 16-byte store and whatever writes its
                    asm volatile("s_nop %0" :: "n"(PW_NOP_AFTER_STORE) : "memory");      //  data registers next: -DPW_NOP_AFTER_STORE=0 is ONE wait state)
#endif
                } else {                                     // (uniform) the last tile of a plane: whole runs, then the partial one by element
                    const int nval = HW - pq;                // valid pixels of the lane's run (>= 4: whole)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, nval >= 4 ? lane_off : OOB, so, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool part = nval < 4 && e < nval;
                        float x = v[e];
                        if (a.res && part) x += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, lane_off + 4 * e, so, 0));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), o_rsrc, part ? lane_off + 4 * e : OOB, so, 0);
                        v[e] = (nval >= 4 || part) ? x : 0.0f;
                    }
                }
                if (a.stats) {
                    s1 += (v[0] + v[1]) + (v[2] + v[3]);
                    s2 += fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
                }
            }
            if (a.stats) {
                // the other 16 pixels of the row sit in lane ^ 32; one row of {sum, sum of squares} per 32 flat pixels
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                const int row = (t.p0 >> 5) + rh * RPW + r;
                if (lk == 0 && row * 32 < HW) {
                    float *dst = a.stats + (((size_t)t.n * a.stats_rows + row) * a.Cout + t.co0 + hq * 32 + l31) * 2;
                    *reinterpret_cast<f32x2 *>(dst) = f32x2{s1, s2};
                }
            }
        }
        // The last 16-byte store of the item sits at the end of the loop body, and the VALU write that recycles its data
        // registers at the head of the next chunk is across the loop's back edge, where the compiler's hazard recogniser does
        // not look: without these wait states the store read zeros (the next stage offset) for lanes 12-15 of every row of
        // 16 (found by tools/pw_check.py: a few wrong pixels, run-dependent).
#ifndef PW_NO_TAIL_WAIT                 // (pw_repro.hip: -DPW_NO_TAIL_WAIT builds the kernel WITHOUT these wait states)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7" ::: "memory");
#endif
        ++k;
    }
}

}  // namespace

namespace ipdm {

bool conv_pw_shape_ok(int Cout, int Cin, int ks, int stride, int interleave)
{
    return ks == 1 && stride == 1 && (interleave == 2 || interleave == 4) && Cout % BN == 0 && Cin > KC;      // (two chunks at least)
}

// [8-channel group][32-cout tile][lk][cout 32][kp 4], channel = 8 group + 2 kp + lk; channels padded with zeros to whole chunks
void conv_pack_weights_pw(const float *w, int Cout, int Cin, std::vector<float> &packed)
{
    const int ng = (Cin + KC - 1) / KC * 4, nt = Cout / 32;
    packed.assign((size_t)ng * nt * W_BLOCK_FLOATS, 0.0f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const int g = ci >> 3, kp = (ci & 7) >> 1, lk = ci & 1;
            packed[(((size_t)g * nt + co / 32) * 2 + lk) * 128 + (co % 32) * 4 + kp] = w[(size_t)co * Cin + ci];
        }
}

// Which 1x1 convolutions run here: whole 128-cout tiles, a concat that splits at a chunk boundary, an NCHW x1 (readers of
// a parity-planar source stay on conv_ws.hip) and not one of the layers conv_ws.hip splits along K (too few tiles for any
// tiling).  A rule of the layer alone: this kernel and conv_ws.hip write different statistics rows.
bool conv_pw_eligible(const ConvArgs &a)
{
    if (opt(OPT_CONV_NO_PW) || !a.w_wino || a.ksize != 1 || a.stride != 1 || a.upsample || a.x1_planar) return false;
    if (a.H != a.Ho || a.W != a.Wo || a.Hs != a.H || a.Ws != a.W) return false;
    if (!conv_pw_shape_ok(a.Cout, a.C1 + a.C2, a.ksize, a.stride, a.w_interleave)) return false;
    if (a.C2 && a.C1 % KC) return false;
    return conv_ws_split(a) == 1;
}

int conv_pw_stats_rows(const ConvArgs &a) { return cdiv((long)a.Ho * a.Wo, 32); }

template <int RPW>
static int launch_pw(ConvArgs a, hipStream_t st)
{
    a.tiles_x = cdiv((long)a.Ho * a.Wo, 64 * RPW);
    a.tiles_y = 1;
    const long nitems = (long)a.tiles_x * a.co_tiles * a.B;
    const int cus = device_cu_count();
    long G = nitems < cus ? nitems : cus;
    G = (G + 7) / 8 * 8;
    if (int rc = ensure_dynamic_lds((const void *)conv_pw_kernel<RPW>, lds_bytes<RPW>())) return rc;
    hipLaunchKernelGGL((conv_pw_kernel<RPW>), dim3((unsigned)G), dim3(512), lds_bytes<RPW>(), st, a, (int)nitems);
    return IPDM_OK;
}

int conv2d_pw_launch(const ConvArgs &args, hipStream_t st)
{
    ConvArgs a = args;
    IPDM_REQUIRE(conv_pw_eligible(args), "conv2d_pw: layer not eligible");
    a.w = args.w_wino;                                   // (for a 1x1 layer this field carries conv_pack_weights_pw's image)
    a.co_tiles = a.Cout / BN;
    a.ksplit = 1;
    const long HW = (long)a.Ho * a.Wo;
    IPDM_REQUIRE((long)a.C1 * HW < (1L << 29) && (long)(a.C2 + 1) * HW < (1L << 29) && (long)a.Cout * HW < (1L << 29),
                 "conv2d_pw: per-sample tensor exceeds the 2 GiB buffer-addressing range");
    IPDM_REQUIRE(!a.stats || a.stats_rows == conv_pw_stats_rows(a), "conv2d_pw: statistics rows %d != %d", a.stats_rows, conv_pw_stats_rows(a));
    // tile height by the fill of the launch (the accumulation order of an output and the statistics rows do not depend on
    // it, so this may look at the batch): the largest tile that still gives every CU most of a round
    const long per_px_row = (long)a.co_tiles * a.B;
    const bool prof = prof_enabled();
    if (prof) prof_before(1, st);
    int rc;
    if (cdiv(HW, 512) * per_px_row >= 192) rc = launch_pw<8>(a, st);
    else if (cdiv(HW, 256) * per_px_row >= 192) rc = launch_pw<4>(a, st);
    else rc = launch_pw<2>(a, st);
    if (rc) return rc;
    if (prof) prof_after(1, 2.0 * a.B * HW * (double)a.Cout * (a.C1 + a.C2), st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

}  // namespace ipdm
