// 1x1 convolutions (pointwise: qkv / proj_out of the AttentionBlocks, the channel-changing shortcuts of the ResidualBlocks;
// Model/model.py:116-119,142-155) as a plain GEMM over channels on the exact-f32 MFMA of gfx950 -- round 4.
//
//     out[n][co][p] = sum_c W[co][c] * gn(x[n][c][p]) + bias[co] (+ res[n][co][p]),   p = the H*W pixels of a plane, FLAT
//
// Why a kernel of its own.  conv_ws.hip runs these layers through its producer/consumer structure (LDS stage, one hand-over
// barrier per 32-channel chunk) at 0.4-0.6 of the f32 MFMA peak, and the first attempt at a pointwise kernel
// (tools/experiments/conv_pw.hip: the conv_wino2 recipe, LDS-staged pixels, one barrier per chunk) did not beat it: what
// those structures lose is the hand-over, not issue slots (NOTEBOOK.md).  A pointwise operator needs no hand-over at
// all -- both MFMA operands can be loaded from memory in exactly the lane layout the instruction wants:
//
//   * a WAVE is the unit of work, not a workgroup: item = 32 flat pixels x 128 couts (four 32x32 accumulators), walked over
//     all input channels.  No LDS on the data path, NO BARRIER anywhere: the eight waves of a workgroup (one per CU, two per
//     SIMD) never wait for each other, their phases drift apart, and whatever one wave does outside its MFMA stream (item
//     epilogue, address arithmetic, a late load) is covered by the other wave of its SIMD;
//   * operand roles swapped against conv_ws.hip -- pixels on M, couts on N -- so that four consecutive accumulator
//     registers are four consecutive pixels of one cout (16-byte stores and residual loads, in-lane statistics);
//   * A operand (pixels): lane (pixel l & 31, channel parity l >> 5) loads ONE dword x[c0 + 2 j + (l >> 5)][p0 + (l & 31)]: a
//     wave-load is two runs of 128 contiguous bytes, addressing is a per-item lane offset plus a scalar channel offset;
//   * B operand (weights): the slab conv_pack_weights already builds for these layers ([Cin][128-cout group][cout & 31][cout
//     >> 5], interleave 4): the four floats a lane needs for the four cout blocks of one k step are adjacent -- one
//     16-byte load (two runs of 512 contiguous bytes per wave-load), straight from L2 into the B registers;
//   * both are loaded a whole 32-channel chunk (16 k steps) ahead into a register ring, reloaded in place right after the
//     MFMAs that read them; the ring runs across item boundaries (the issue side is one chunk ahead of the multiplying side);
//   * GroupNorm of the input (the qkv projection; no layer applies SiLU in front of a 1x1): one fma on the A register with
//     the channel's {scale, shift} from a WAVE-PRIVATE table in LDS (rewritten when the wave's sample changes: no barrier);
//   * parity-planar x1 (the output of an up2 convolution, conv_ws.hip): only the lane offsets differ.
// Fused GroupNorm statistics of the output: one row of per-cout {sum, sum of squares} per 32 flat pixels.
// Schedule: static, wave w of the launch takes items w, w + #waves, ...; items are ordered cout tile fastest, so the waves
// of a CU that share a pixel tile read it from the same L1 / L2.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef IPDM_PW_SAFE_WAIT
#define IPDM_PW_SAFE_WAIT 0         // 1: every wait for a ring slot drains the whole queue (s_waitcnt vmcnt(0)) instead of the hand-counted
#endif                              // vmcnt(N): `make pwsafe` builds libipdm_hip_pwsafe.so with it, and a GPU test holds the shipped kernel's bits
                                    // to that build's (tests/test_gpu_parity.py::test_pointwise_ring_waits_equal_a_full_drain)
#ifndef IPDM_PW_KO
#define IPDM_PW_KO 0                // compile-time timing knock-out (tools/build_variants.sh; WRONG results): 1 = no operand loads after the
#endif                              // prologue.  (Skipping the epilogue is not a valid knock-out: the MFMAs become dead code.)

namespace {

constexpr int BN = 128;                                // couts per item
constexpr int NW = 8;                                  // waves per workgroup
constexpr int OOB = 0x7fffffff;
constexpr int KC_MIN = 32;                             // channel granularity the launcher asks for (both ring shapes divide it)

__device__ inline float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ inline f32x4 bload4(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// The register ring is loaded and waited for BY HAND.  With builtin loads the compiler's wait-count pass cannot follow loads
// that stay in flight across the loop's back edge: it drains the queue (s_waitcnt vmcnt(0)) at the top of every chunk, which
// turns the ring into load-everything-then-multiply.  Invisible to that pass, the loads below are waited for with the exact
// count: (NPB + 1) (DEPTH - 1) ring loads are issued between a slot's loads and its use, loads return in order, and anything
// else the wave has in flight by then (bias / residual loads, stores) is younger and only makes the wait stricter.  The
// compiler's own waits (for ITS loads) do not know about the ring either and are likewise stricter than necessary, never
// weaker.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ inline i32x4 make_rsrc(const void *p, int bytes)
{
    const unsigned long long u = (unsigned long long)p;
    return i32x4{(int)(unsigned)u, (int)((unsigned)(u >> 32) & 0xffffu), bytes, 0x00020000};
}
__device__ inline void ring_load_x(float &x, int x_v, i32x4 x_rsrc, int x_s)
{
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=&v"(x) : "v"(x_v), "s"(x_rsrc), "s"(x_s));
}
__device__ inline void ring_load_w(f32x4 &w, int w_v, i32x4 w_rsrc, int w_s)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(w) : "v"(w_v), "s"(w_rsrc), "s"(w_s));
}
// (the slot's registers are in / out operands of the wait: nothing that reads them can be scheduled above it)
template <int N>
__device__ inline void ring_wait(float (&x)[1], f32x4 &w)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(x[0]), "+v"(w) : "n"(IPDM_PW_SAFE_WAIT ? 0 : N));
}
template <int N>
__device__ inline void ring_wait(float (&x)[2], f32x4 &w)
{
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(x[0]), "+v"(x[1]), "+v"(w) : "n"(IPDM_PW_SAFE_WAIT ? 0 : N));
}

struct Item { int n, p0, co0; };

// NPB: 32-pixel blocks per item (1: 32 x 128 outputs, a 16-step ring of 5 registers; 2: 64 x 128 outputs, an 8-step ring of 6
// -- half the weight loads per MFMA, twice the work per item).  Both accumulate an output in the same order.
template <bool ACT, bool PLANAR, int NPB>
__global__ void __launch_bounds__(512) conv_pw_kernel(ConvArgs a, int nitems)
{
    constexpr int DEPTH = 16 / NPB;                        // k steps in the ring
    constexpr int KC = 2 * DEPTH;                          // channels per chunk
    constexpr int BP = 32 * NPB;                           // flat pixels per item
    constexpr int WAITN = (NPB + 1) * (DEPTH - 1);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 5, l31 = lane & 31;
    // the workgroups of one XCD (blockIdx & 7) hold a contiguous run of wave indices: neighbouring items -- the cout tiles of
    // one pixel tile, then the next pixel tile -- meet in the same L2
    const int G = gridDim.x, per = G >> 3;
    const int local = ((blockIdx.x & 7) * per + (blockIdx.x >> 3)) * NW + swave;
    const int WT = G * NW;
    if (local >= nitems) return;
    const int n_my = (nitems - 1 - local) / WT + 1;
    const int Ctot = a.C1 + a.C2, nch = Ctot / KC, HW = a.Ho * a.Wo;
    const int plane_bytes = HW * 4;
    auto decode = [&](int k) {
        const int item = k * WT + local;
        Item t;
        t.co0 = (item % a.co_tiles) * BN;
        const int rest = item / a.co_tiles;
        t.p0 = (rest % a.tiles_x) * BP;
        t.n = rest / a.tiles_x;
        return t;
    };

    // ---------------------------------------------------------------- issue side: the chunk whose operands are being LOADED
    Item nx = decode(0);
    int i_ch = 0, i_k = 0;
    int vx[NPB], vxp[PLANAR ? NPB : 1];                    // the lane's byte offsets inside a channel pair: NCHW / parity-planar x1
    auto lane_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            const int p = nx.p0 + 32 * pb + l31;
            // (past the end of the plane -- the ragged last item -- the loads read the next channel or, past the tensor, 0:
            //  those pixel ROWS of the accumulators are never stored, and a pixel's row depends on its own operand row only)
            vx[pb] = (lk * HW + p) * 4;
            if (PLANAR) {
                const int pc = p < HW ? p : HW - 1;
                const int y = pc / a.Wo, x = pc - y * a.Wo;
                vxp[PLANAR ? pb : 0] = (lk * HW + ((y & 1) * 2 + (x & 1)) * (HW >> 2) + (y >> 1) * (a.Wo >> 1) + (x >> 1)) * 4;
            }
        }
    };
    const int wv = (lk * a.cout_pad + l31 * 4) * 4;        // the lane's byte offset inside a (channel pair, 128-cout group)
    const i32x4 w_rsrc = make_rsrc(a.w, Ctot * a.cout_pad * 4);
    float xs[DEPTH][NPB];
    f32x4 ws[DEPTH];

    // ---------------------------------------------------------------- multiplying side
    f32x16 acc[NPB][4];
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float *const tab = lds + swave * (ACT ? Ctot * 2 : 0);                // the wave's {scale, shift} table
    int tab_n = -1;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
    float bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float one_k0 = lk == 0 ? 1.0f : 0.0f;
    const int vo = (l31 * HW + 4 * lk) * 4;                // output / residual: cout l31 of a block, the lane's first pixel of a run of 8

    // One chunk: the k steps of the ring (4 NPB MFMAs each); every slot is reloaded with the issue-side chunk's operands as
    // soon as its MFMAs are out.
    auto chunk = [&](int c_ch) __attribute__((always_inline)) {
        const int c0 = i_ch * KC;
        const bool from1 = c0 < a.C1;
        const i32x4 x_rsrc = make_rsrc(from1 ? a.x1 + (size_t)nx.n * a.C1 * HW : a.x2 + (size_t)nx.n * a.C2 * HW, (from1 ? a.C1 : a.C2) * plane_bytes);
        const int x_s = (from1 ? c0 : c0 - a.C1) * plane_bytes;
        const int w_s = (c0 * a.cout_pad + nx.co0) * 4;
        const float *const tc = tab + (c_ch * KC + lk) * 2;
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            ring_wait<WAITN>(xs[j], ws[j]);
            float xv[NPB];
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) xv[pb] = xs[j][pb];
            if (ACT) {
                const f32x2 st = *reinterpret_cast<const f32x2 *>(tc + j * 4);
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) xv[pb] = __builtin_fmaf(xv[pb], st[0], st[1]);
            }
            const f32x4 wq = ws[j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[pb], wq[b], acc[pb][b], 0, 0, 0);
            if (!(IPDM_PW_KO & 1)) {
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb)
                    ring_load_x(xs[j][pb], (PLANAR && from1) ? vxp[PLANAR ? pb : 0] : vx[pb], x_rsrc, x_s + j * 2 * plane_bytes);
                ring_load_w(ws[j], wv, w_rsrc, w_s + j * 2 * a.cout_pad * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the issue side moves on: next chunk of its item, or the first chunk of the wave's next item (past the last item it
        // keeps re-reading that item's chunks: valid addresses, results unused)
        if (++i_ch == nch) {
            i_ch = 0;
            if (++i_k < n_my) { nx = decode(i_k); lane_offsets(); }
        }
    };

    // ---------------------------------------------------------------- prologue: chunk 0 of item 0 in flight
    lane_offsets();
    Item cur = nx;
    {
        const i32x4 x_rsrc = make_rsrc(a.x1 + (size_t)nx.n * a.C1 * HW, a.C1 * plane_bytes);
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) ring_load_x(xs[j][pb], PLANAR ? vxp[PLANAR ? pb : 0] : vx[pb], x_rsrc, j * 2 * plane_bytes);
            ring_load_w(ws[j], wv, w_rsrc, (j * 2 * a.cout_pad + nx.co0) * 4);
        }
        i_ch = 1;
    }

    for (int k = 0; k < n_my; ++k) {
        // ---------------------------------------------------------------- item start
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[pb][b] = zero16;
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = bload(b_rsrc, l31 * 4, (cur.co0 + 32 * b) * 4);
        if (ACT && cur.n != tab_n) {
            for (int c = lane; c < Ctot; c += 64)
                *reinterpret_cast<f32x2 *>(tab + c * 2) = f32x2{a.gn_scale[(size_t)cur.n * Ctot + c], a.gn_shift[(size_t)cur.n * Ctot + c]};
            tab_n = cur.n;
        }
        for (int c_ch = 0; c_ch < nch; ++c_ch) chunk(c_ch);

        // ---------------------------------------------------------------- item epilogue
        // + bias: one more k step with A = 1 (k lane 0), B = the cout's bias
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[pb][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(one_k0, bv[b], acc[pb][b], 0, 0, 0);
        const size_t sample = (size_t)cur.n * a.Cout * HW;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + sample), 0, a.Cout * plane_bytes, 0x00020000);
        // (no residual: zero records -- the loads return 0 and the add stays unconditional)
        // (the GroupNorm instantiations -- the qkv projections -- have no residual path: no layer of the reference normalises the
        //  input of a 1x1 AND adds a residual, conv_pw_layer_ok refuses such a layer, and the 32 registers of the residual
        //  prefetch were what made the ACT / 64-pixel instantiations spill next to the register ring: ADVICE r04.  The Makefile
        //  refuses a build in which any instantiation of this kernel spills a vector register)
        const bool has_res = !ACT && a.res;
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((has_res ? a.res : a.out) + sample), 0,
                                                                                   has_res ? a.Cout * plane_bytes : 0, 0x00020000);
        // a 32-pixel block x 32-cout block of the item: + residual, 16-byte stores, the block's statistics row
        auto block = [&](int pb, int b, const f32x4 (&r)[4]) __attribute__((always_inline)) {
            const int so = ((cur.co0 + 32 * b) * HW + cur.p0 + 32 * pb) * 4;
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = f32x4{acc[pb][b][4 * q], acc[pb][b][4 * q + 1], acc[pb][b][4 * q + 2], acc[pb][b][4 * q + 3]} + r[q];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), o_rsrc, vo, so + q * 32, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1 += v[e]; s2 = __builtin_fmaf(v[e], v[e], s2); }
            }
            if (a.stats) {
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lk == 0)
                    *reinterpret_cast<f32x2 *>(a.stats + (((size_t)cur.n * a.stats_rows + cur.p0 / 32 + pb) * a.Cout + cur.co0 + 32 * b + l31) * 2) = f32x2{s1, s2};
            }
        };
        // ... of the ragged end of the plane (nv < 32 valid pixels): dword by dword
        auto ragged_block = [&](int pb, int b, int nv) __attribute__((always_inline)) {
            const int so = ((cur.co0 + 32 * b) * HW + cur.p0 + 32 * pb) * 4;
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int px = 8 * (i >> 2) + 4 * lk + (i & 3);
                const int off = px < nv ? vo + (i >> 2) * 32 + (i & 3) * 4 : OOB;
                const float v = ACT ? acc[pb][b][i] : acc[pb][b][i] + bload(r_rsrc, off, so);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, off, so, 0);
                const float vm = px < nv ? v : 0.0f;
                s1 += vm; s2 = __builtin_fmaf(vm, vm, s2);
            }
            if (a.stats) {
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lk == 0)
                    *reinterpret_cast<f32x2 *>(a.stats + (((size_t)cur.n * a.stats_rows + cur.p0 / 32 + pb) * a.Cout + cur.co0 + 32 * b + l31) * 2) = f32x2{s1, s2};
            }
        };
        if (HW - cur.p0 >= BP) {
            // (a wait for a residual load also waits for every store issued before it: the loads run one block ahead of the
            //  stores, and a layer without a residual issues none)
            if (has_res) {
                f32x4 r[2][4];
                auto fetch = [&](int t) __attribute__((always_inline)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        r[t & 1][q] = bload4(r_rsrc, vo, ((cur.co0 + 32 * (t & 3)) * HW + cur.p0 + 32 * (t >> 2)) * 4 + q * 32);
                };
                fetch(0);
#pragma unroll
                for (int t = 0; t < 4 * NPB; ++t) {
                    if (t + 1 < 4 * NPB) fetch(t + 1);
                    block(t >> 2, t & 3, r[t & 1]);
                }
            } else {
                const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
                const f32x4 r0[4] = {z4, z4, z4, z4};
#pragma unroll
                for (int t = 0; t < 4 * NPB; ++t) block(t >> 2, t & 3, r0);
            }
        } else {
            for (int pb = 0; pb < NPB; ++pb) {
                const int nv = HW - cur.p0 - 32 * pb;
                if (nv <= 0) break;
                if (nv >= 32) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        f32x4 r[4] = {};
                        if (!ACT) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) r[q] = bload4(r_rsrc, vo, ((cur.co0 + 32 * b) * HW + cur.p0 + 32 * pb) * 4 + q * 32);
                        }
                        if (pb == 0) block(0, b, r); else block(NPB - 1, b, r);
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) { if (pb == 0) ragged_block(0, b, nv); else ragged_block(NPB - 1, b, nv); }
                }
            }
        }
        cur = nx;
    }
}

}  // namespace

namespace ipdm {

// The layers this kernel CAN take: every wide 1x1 with whole 128-cout groups and whole 32-channel chunks (all of the reference
// architectures' qkv / proj_out / shortcut layers from 128 couts up), except the few low-resolution ones conv_ws.hip splits
// along K (conv_ws_split: a rule of the layer alone).
bool conv_pw_layer_ok(const ConvArgs &a)
{
    if (opt(OPT_CONV_NO_PW)) return false;
    const int Ctot = a.C1 + a.C2;
    if (a.ksize != 1 || a.stride != 1 || a.w_interleave != 4 || a.Cout % BN || a.cout_pad != a.Cout) return false;
    if (a.upsample || a.H != a.Hs || a.W != a.Ws || a.Ho != a.H || a.Wo != a.W) return false;
    if (Ctot % KC_MIN || Ctot < 2 * KC_MIN || (a.C2 && a.C1 % KC_MIN) || a.act == 2 || a.sk_w || (a.act && a.res)) return false;
    if ((long)(a.C1 > a.C2 ? a.C1 : a.C2) * a.Ho * a.Wo >= (1L << 29) || (long)a.Cout * a.Ho * a.Wo >= (1L << 29)) return false;
    return conv_ws_split(a) == 1;
}

// ... and the ones it DOES take.  An item is one wave walking all input channels: a launch with fewer items than waves runs
// at the latency of that walk (27 us at 256 channels), where conv_ws.hip puts a whole workgroup on a quarter tile -- a lone
// slice's qkv / proj_out launches were up to 2x slower here (profiles/r04d_layer_sweep_b1.txt).  Outputs are the same bits
// either way, so a layer WITHOUT fused statistics chooses by the fill of THIS launch (>= 1.3 rounds of 32-pixel items);
// the statistics rows of the two kernels differ in geometry (sums over different pixel sets round differently), so a layer
// WITH fused statistics must not look at the batch: it comes here only if one sample alone brings >= 1024 items.
bool conv_pw_stats_layer(const ConvArgs &a)
{
    if (!conv_pw_layer_ok(a)) return false;
    return opt(OPT_PW_FORCE) || (long)cdiv((long)a.Ho * a.Wo, 32) * (a.Cout / BN) >= 1024;      // (pw_force: tests drive small shapes through the kernel)
}
bool conv_pw_eligible(const ConvArgs &a)
{
    if (!conv_pw_layer_ok(a)) return false;
    if (a.stats) return conv_pw_stats_layer(a);
    if (opt(OPT_PW_FORCE)) return true;
    const long items = (long)cdiv((long)a.Ho * a.Wo, 32) * (a.Cout / BN) * a.B;
    return 10 * items >= 13L * device_cu_count() * NW;
}

int conv_pw_stats_rows(const ConvArgs &a) { return cdiv((long)a.Ho * a.Wo, 32); }

namespace {
template <int NPB>
void launch_pw(const ConvArgs &a, long G, size_t lds_bytes, long nitems, hipStream_t st)
{
    if (a.act) {
        if (a.x1_planar) hipLaunchKernelGGL((conv_pw_kernel<true, true, NPB>), dim3((unsigned)G), dim3(512), lds_bytes, st, a, (int)nitems);
        else hipLaunchKernelGGL((conv_pw_kernel<true, false, NPB>), dim3((unsigned)G), dim3(512), lds_bytes, st, a, (int)nitems);
    } else {
        if (a.x1_planar) hipLaunchKernelGGL((conv_pw_kernel<false, true, NPB>), dim3((unsigned)G), dim3(512), lds_bytes, st, a, (int)nitems);
        else hipLaunchKernelGGL((conv_pw_kernel<false, false, NPB>), dim3((unsigned)G), dim3(512), lds_bytes, st, a, (int)nitems);
    }
}
}  // namespace

int conv2d_pw_launch(const ConvArgs &args, hipStream_t st)
{
    ConvArgs a = args;
    IPDM_REQUIRE(conv_pw_eligible(a), "conv2d_pw: not a layer of this kernel");
    IPDM_REQUIRE(!a.x1_planar || (!(a.Ho & 1) && !(a.Wo & 1)), "conv2d_pw: parity-planar source of odd size %dx%d", a.Ho, a.Wo);
    const int HW = a.Ho * a.Wo, Ctot = a.C1 + a.C2;
    IPDM_REQUIRE(!a.stats || a.stats_rows == conv_pw_stats_rows(a), "conv2d_pw: statistics rows %d != %d", a.stats_rows, conv_pw_stats_rows(a));
    const int cus = device_cu_count();
    // Item shape: 64-pixel items load half the weights per MFMA and run ~7 % faster per pixel (tools/pw_check.py), but the
    // static schedule's makespan is whole items per wave: take them when 2 x 0.93 x their rounds does not exceed the rounds of
    // the 32-pixel items.  Both shapes produce the same bits, so the choice may look at the batch (pw_item: 1 / 2 force one).
    a.co_tiles = a.Cout / BN;
    const long waves = (long)cus * NW;
    const long r32 = cdiv((long)cdiv(HW, 32) * a.co_tiles * a.B, waves), r64 = cdiv((long)cdiv(HW, 64) * a.co_tiles * a.B, waves);
    const int forced = opt(OPT_PW_ITEM);
    const int npb = forced ? forced : (1.86 * r64 <= (double)r32 ? 2 : 1);
    a.tiles_x = cdiv(HW, 32 * npb);
    a.tiles_y = 1;
    const long nitems = (long)a.tiles_x * a.co_tiles * a.B;
    IPDM_REQUIRE(nitems < (1L << 31), "conv2d_pw: too many items");
    long G = cdiv(nitems, NW) < cus ? cdiv(nitems, NW) : cus;
    G = (G + 7) / 8 * 8;
    const size_t lds_bytes = a.act ? (size_t)NW * Ctot * 2 * sizeof(float) : 0;
    IPDM_REQUIRE(lds_bytes <= 64 * 1024, "conv2d_pw: %d input channels exceed the scale/shift tables", Ctot);
    const bool prof = prof_enabled();
    if (prof) prof_before(1, st);
    if (npb == 2) launch_pw<2>(a, G, lds_bytes, nitems, st);
    else launch_pw<1>(a, G, lds_bytes, nitems, st);
    if (prof) prof_after(1, 2.0 * a.B * HW * (double)a.Cout * Ctot, st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

}  // namespace ipdm
