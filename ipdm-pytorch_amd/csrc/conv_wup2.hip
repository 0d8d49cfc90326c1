// The wide Upsample convolutions (nearest 2x + 3x3, Model/model.py Upsample) in the Winograd F(2x2, 2x2) domain of their
// parity form, on the exact-f32 MFMA (gfx950) -- round 5.
//
// The parity form (conv_ws.hip, conv_pack_weights_up2) evaluates the layer on the SOURCE grid as four 2x2-tap convolutions, one
// per output parity (a, b): 4 multiply-adds per output and (cin, cout) instead of 9.  A 2-tap filter has a minimal algorithm
// too -- F(2, 2): outputs y0 = g0 d0 + g1 d1, y1 = g0 d1 + g1 d2 from three products
//
//     m0 = (d0 - d1) g0,   m1 = d1 (g0 + g1),   m2 = (d2 - d1) g1;      y0 = m0 + m1,   y1 = m1 + m2
//
// (all transform entries 0 / +-1: better conditioned than F(2,3); tools/wino_accuracy.py: 0.70x the float32 error of the 3x3
// form on the up-sampled image, the 2x2-tap parity form has 0.66x).  In two dimensions: 9 products per 2x2 outputs of one
// parity instead of 16 -- 2.25 multiply-adds per output where the 3x3 form has 9.  The four parities of a 2x2 block of source
// pixels read the same 4x4 source patch as a tile of conv_wino2.hip does, and this kernel is that kernel's structure:
//
//   * an item = (sample, 4 x 32 source pixels, 128 couts, ROW parity a): 32 patches, for each the 2 x 4 outputs of rows 2y + a;
//     window = 5 source rows from oy0 - 1 + a (a patch row-transforms as R0 - R1, R1, R2 - R1 whatever a is);
//   * wave w: COLUMN parity b = w & 1, cout quarter w >> 1; nine positions (p, q) of the 3x3 transform domain = nine 32x32
//     accumulators; the column values of both parities are five per patch row -- X0 - X1, X1, X2 - X1, X2, X3 - X2 -- parity 0
//     reads the first three, parity 1 the last three, with the sign of its first one folded into the packed weights;
//   * staging as in conv_wino2 (every wave stages two channels of a 16-channel chunk: 16-byte loads -> wave-private scratch ->
//     patch -> 15 transform values into the shared V stage), without a prologue: an Upsample has no GroupNorm in front;
//   * the A operands (U) go from L2 straight into registers, one position ahead, image
//     [8-channel chunk][128-cout tile][a][b][cout quarter][position 9][k parity][cout 32][k step 4];
//   * the output transform is in-lane (a wave owns all nine positions of its outputs: no exchange, no second barrier), the
//     lanes of two neighbouring patches swap halves for 16-byte stores into plane (a, b) of the parity-planar output
//     [n][cout][a][b][H][W]; statistics rows as conv_ws's parity form writes them.
// One barrier per 16-channel chunk (72 MFMAs per wave).  Bit-identical across batch sizes (the rule looks at the layer only).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef IPDM_WUP2_KO
#define IPDM_WUP2_KO 0              // compile-time timing knock-outs (results are WRONG): 2 no input transform, 4 no output transform / stores,
#endif                              // 8 no window loads, 16 no U loads, 32 output transform kept but no stores / statistics, 64 no statistics

namespace {

constexpr int KC = 16;                                 // channels per staged chunk: two MFMA sub-chunks of 8 (4 k-steps of 2)
constexpr int TH = 4, TW = 32, BN = 128;               // source pixels / couts of an item
constexpr int NW = 8;                                  // waves
constexpr int NV = 15;                                 // transform values per patch and channel: 3 rows x 5 column values
constexpr int VH_FLOATS = NV * 2 * 32 * 4;             // V of 8 channels [value 15][lk 2][patch 32][kp 4]: 15 KB
constexpr int V_FLOATS = 2 * VH_FLOATS;                // a stage: both sub-chunks
constexpr int U_WAVE_BYTES = 9 * 2 * 32 * 4 * 4;       // a wave's nine positions of one 8-channel chunk: 9 KB
constexpr int U_ITEM_BYTES = 8 * U_WAVE_BYTES;         // one (8-channel chunk, 128-cout tile, row parity): [b 2][cout quarter 4]
constexpr int XP = 40;                                 // scratch row pitch (34 window columns, 36 loaded)
constexpr int XWAVE = 10 * XP + 64 * 4 + 8;            // per wave: 10 row segments (2 channels x 5 window rows) + a dump slot per lane
constexpr size_t LDS_BYTES = (size_t)(2 * V_FLOATS + NW * XWAVE) * sizeof(float);
static_assert(LDS_BYTES <= 160 * 1024, "conv_wup2: LDS budget exceeded");
static_assert(256 <= XWAVE, "conv_wup2: the statistics staging aliases the wave's scratch");

struct ItemId { int n, oy0, ox0, co0, a; };

__device__ inline ItemId decode_item(const ConvArgs &a, int item)
{
    ItemId t;
    t.a = item & 1;                                    // the two row parities of a tile are neighbours in the schedule (same window but one row)
    const int tile = item >> 1;
    const int co_t = tile % a.co_tiles;
    int rest = tile / a.co_tiles;
    const int tx = rest % a.tiles_x;
    rest /= a.tiles_x;
    const int ty = rest % a.tiles_y;
    t.n = rest / a.tiles_y;
    t.oy0 = ty * TH;
    t.ox0 = tx * TW;
    t.co0 = co_t * BN;
    return t;
}

constexpr int OOB = 0x7fffffff;                        // per-lane buffer offset out of range: loads return 0, stores are dropped

__device__ inline float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// a.H x a.W: the SOURCE grid (= the size of one parity plane of the output); a.w: the image of conv_pack_weights_wup2
__global__ void __launch_bounds__(512) conv_wup2_kernel(ConvArgs a, int nitems)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *const xw = lds + 2 * V_FLOATS + swave * XWAVE;      // the wave's scratch (statistics staging in the epilogue)

    // static schedule: the workgroups of one XCD take a contiguous run of items, slot rotated per round (conv_wino2.hip)
    const int G = gridDim.x, per = G >> 3;
    const int local = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int rounds = (nitems + G - 1) / G;
    auto item_of = [&](int k) { return k * G + (local + 5 * k) % G; };
    const int n_my = rounds == 0 ? 0 : (item_of(rounds - 1) < nitems ? rounds : rounds - 1);
    const int C = a.C1;
    const int nchunks = C / KC;                          // launcher: C % KC == 0, nchunks >= 2
    const int S = n_my * nchunks;
    const int HW = a.H * a.W;
    const int plane_bytes = HW * 4;
    if (S == 0) return;

    // =============================================================================== staging role
    // wave w: channels 2 w, 2 w + 1 of the 16-channel chunk, all five window rows (34 columns, nine 16-byte parts) -- 90 of the
    // wave's 128 load slots
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, (C / 8) * a.co_tiles * 2 * U_ITEM_BYTES, 0x00020000);
    int lconst[2], xoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = lane + 64 * j, cc = u / 45, r = (u - cc * 45) / 9, part = u - cc * 45 - r * 9;
        const bool v = u < 90;
        lconst[j] = v ? (r * a.W + 4 * part) * 4 + cc * plane_bytes : OOB;
        xoff[j] = v ? (cc * 5 + r) * XP + 4 * part : 10 * XP + lane * 4;
    }
    // item descriptors: issue side (g_*: the item whose chunks are being LOADED), parking side (a_*: one chunk behind)
    int g_n = 0, g_co = 0, g_oy = 0, g_ox = 0, g_a = 0;
    const float *g_src = a.x1;
    bool g_bord = false, a_bord = false;
    int vo[2] = {lconst[0], lconst[1]}, g_so = 0;
    unsigned g_vm = 0xffu, g_lsh = 0, a_vm = 0xffu, a_lsh = 0;
    auto describe = [&](int k) __attribute__((always_inline)) {
        const ItemId tl = decode_item(a, item_of(k));
        const int iy0 = tl.oy0 - 1 + tl.a, ix0 = tl.ox0 - 1;
        g_n = tl.n; g_co = tl.co0 / BN; g_oy = tl.oy0; g_ox = tl.ox0; g_a = tl.a;
        g_src = a.x1 + (size_t)tl.n * C * HW;
        g_bord = iy0 < 0 || ix0 < 0 || iy0 + 5 > a.H || tl.ox0 + TW + 1 > a.W;
        const int g_base = (iy0 * a.W + ix0) * 4;
        g_so = g_bord ? 0 : g_base;
#pragma unroll
        for (int j = 0; j < 2; ++j) vo[j] = lconst[j];
        if (g_bord) {
            g_vm = 0; g_lsh = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int u = lane + 64 * j, cc = u / 45, r = (u - cc * 45) / 9, part = u - cc * 45 - r * 9;
                const bool rowok = u < 90 && iy0 + r >= 0 && iy0 + r < a.H;
                // the 16 bytes of the leftmost part of an image row start one pixel before the row: shifted by one pixel and
                // rotated back after the load (at the very first row they would start before the buffer)
                const bool lsh = rowok && ix0 + 4 * part < 0;
                g_lsh |= lsh ? 1u << j : 0u;
                vo[j] = rowok ? lconst[j] + g_base + (lsh ? 4 : 0) : OOB;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ix = ix0 + 4 * part + e;
                    g_vm |= (rowok && ix >= 0 && ix < a.W && 4 * part + e < 34) ? 1u << (4 * j + e) : 0u;
                }
                // (a load of a border item may straddle the end of an image row or of the tensor: those elements are masked by
                //  g_vm, and a raw buffer load range-checks dword by dword -- conv_wino2.hip)
            }
        }
    };
    f32x4 raw[2];
    // every iteration issues the same loads, needed or not (past the end of the stream they re-read chunks of the last item)
    auto issue_raw = [&](int ch) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g_src, 0, C * plane_bytes, 0x00020000);
        const int cb = (ch * KC + 2 * swave) * plane_bytes;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            raw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, vo[j], cb + g_so, 0));
    };
    // zero what lies outside the image, park the 8 landed values in the wave's scratch
    auto park = [&]() __attribute__((always_inline)) {
        f32x4 d[2] = {raw[0], raw[1]};
        if (a_bord) {            // (uniform)
#pragma unroll
            for (int j = 0; j < 2; ++j)      // undo the left-edge shift: {x0, x1, x2, x3} loaded from one pixel further right
                if (a_lsh >> j & 1) d[j] = f32x4{0.0f, d[j][0], d[j][1], d[j][2]};
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e >> 2][e & 3] = (a_vm >> e & 1) ? d[e >> 2][e & 3] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4 *>(xw + xoff[j]) = d[j];
    };

    // =============================================================================== multiplying role
    const int lk = lane >> 5, l31 = lane & 31;
    const int bq = swave & 1, hq = swave >> 1;             // column parity, cout quarter
    // the lane's patch inside the item: row ty, column 2 txh + odd.  The two patches of a column pair sit 16 lanes apart (DPP
    // rows r, r + 1), so that the epilogue's exchange of halves is one v_permlane16_swap per register pair
    const int odd = l31 >> 4, ty = l31 & 1, txh = (l31 & 15) >> 1;
    f32x16 acc[9];
    f32x4 ua[9];                                           // the A operands (U) of the wave's 9 positions: four k steps each
    const int out_plane = 4 * HW;                          // channel stride of the parity-planar output
    const int plane4 = out_plane * 4;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
    float nb = 0.0f;
    auto fetch_bias = [&](int co0) __attribute__((always_inline)) {
        nb = bload(b_rsrc, lk ? OOB : l31 * 4, (co0 + hq * 32) * 4);
    };
    // input transform: the wave transforms the (patch, channel) pairs of its own scratch -- lane map 16 patch columns x 2 patch
    // rows x 2 channels (conv_wino2.hip: the LDS stores of a value are at most 2-way conflicted)
    const int w_tx = lane & 15, w_ty = (lane >> 4) & 1;
    const float *const xr = xw + (lk * 5 + 2 * w_ty) * XP + 2 * w_tx;
    const int v_slot = (w_tx & 1) * 16 + (w_tx >> 1) * 2 + w_ty;                    // MFMA lane of patch (row w_ty, column w_tx)
    const int v_lane = (swave >> 2) * VH_FLOATS + (lk * 32 + v_slot) * 4 + (swave & 3);      // + value * 256 (+ stage)
    f32x2 pp[3][2];
    auto read_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            pp[r][0] = *reinterpret_cast<const f32x2 *>(xr + r * XP);
            pp[r][1] = *reinterpret_cast<const f32x2 *>(xr + r * XP + 2);
        }
    };
    // rows (R0 - R1, R1, R2 - R1), then per row the five column values, into V stage `par`
    auto transform_patch = [&](int par) __attribute__((always_inline)) {
        f32x2 T[3][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            T[0][h] = pp[0][h] - pp[1][h];
            T[1][h] = pp[1][h];
            T[2][h] = pp[2][h] - pp[1][h];
        }
        float *vd = lds + par * V_FLOATS + v_lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float t0 = T[i][0][0], t1 = T[i][0][1], t2 = T[i][1][0], t3 = T[i][1][1];
            vd[(i * 5 + 0) * 256] = t0 - t1;
            vd[(i * 5 + 1) * 256] = t1;
            vd[(i * 5 + 2) * 256] = t2 - t1;
            vd[(i * 5 + 3) * 256] = t2;
            vd[(i * 5 + 4) * 256] = t3 - t2;
        }
    };
    const int u_voff = lane * 16;                                            // bytes; + position * 1024 + the wave's slab
    const int b_off = 2 * bq * 256 + (lk * 32 + l31) * 4;                    // floats; + (p * 5 + q) * 256 (+ stage)
    int w_co = 0, w_a = 0;                                 // cout tile / row parity of the item whose weights are being loaded
    // q8 = 8-channel chunk (2 * chunk + sub-chunk)
    auto issue_u = [&](int e, int q8) __attribute__((always_inline)) {
        const int w_soff = ((q8 * a.co_tiles + w_co) * 2 + w_a) * U_ITEM_BYTES + (bq * 4 + hq) * U_WAVE_BYTES + e * 1024;
        ua[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, u_voff, w_soff, 0));
    };

    // ---------------------------------------------------------------- prologue: item 0, chunk 0 staged, chunk 1 in flight
    describe(0);
    a_vm = g_vm; a_lsh = g_lsh; a_bord = g_bord;
    w_co = g_co; w_a = g_a;
#pragma unroll
    for (int e = 0; e < 9; ++e) issue_u(e, 0);
    issue_raw(0);
    fetch_bias(g_co * BN);
    park();
    issue_raw(1);
    read_patch();
    transform_patch(0);
    __syncthreads();

    int s = 0;                                             // running chunk index (V stage parity)
    int k = 0;
    ItemId cur = {g_n, g_oy, g_ox, g_co * BN, g_a};
    // One chunk: park chunk s + 1 (-> scratch -> patch), multiply chunk s, transform chunk s + 1 into the other V stage, barrier.
    // The first chunk of an item STARTS its accumulators (C = 0 in the first MFMA of each).
    auto chunk = [&](auto first, int ch) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value;
        const bool more1 = s + 1 < S;
        const int ch1 = ch + 1 == nchunks ? 0 : ch + 1, ch2 = ch1 + 1 == nchunks ? 0 : ch1 + 1;
        if (more1) {
            if (ch == nchunks - 1) { a_vm = g_vm; a_lsh = g_lsh; a_bord = g_bord; }      // chunk s + 1 opens the item described last
            park();                                        // raw(s + 1) -> scratch
        }
        if (ch == nchunks - 2 && k + 1 < n_my) describe(k + 1);      // before the first loads of the next item
        if (!(IPDM_WUP2_KO & 8)) issue_raw(ch2);           // raw(s + 2), consumed one iteration from now
        const float *stage = lds + (s & 1) * V_FLOATS;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {                   // the two 8-channel sub-chunks of the staged chunk
            const float *vh = stage + kc * VH_FLOATS;
            f32x4 b_c = *reinterpret_cast<const f32x4 *>(vh + b_off), b_n;
            if (kc == 1) {
                if (!(IPDM_WUP2_KO & 2)) read_patch();     // patch(s + 1) (behind the first operand: the LDS returns in order)
                if (ch1 == 0) { w_co = g_co; w_a = g_a; }  // from here on the weights loaded belong to the item described last
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                // the next position's operand BEFORE this position's MFMAs (pinned, conv_wino2.hip)
                if (e + 1 < 9) b_n = *reinterpret_cast<const f32x4 *>(vh + b_off + (((e + 1) / 3) * 5 + (e + 1) % 3) * 256);
                __builtin_amdgcn_sched_barrier(0);
                if (FIRST && kc == 0) {
                    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[e][0], b_c[0], zero, 0, 0, 0);
                } else {
                    acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[e][0], b_c[0], acc[e], 0, 0, 0);
                }
#pragma unroll
                for (int qq = 1; qq < 4; ++qq) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[e][qq], b_c[qq], acc[e], 0, 0, 0);
                // the U of this position for the NEXT sub-chunk, into the registers just read
                if (!(IPDM_WUP2_KO & 16)) issue_u(e, kc == 0 ? 2 * ch + 1 : 2 * ch1);
                __builtin_amdgcn_sched_barrier(0);
                if (e + 1 < 9) b_c = b_n;
            }
        }
        if (!(IPDM_WUP2_KO & 2)) transform_patch((s + 1) & 1);      // V(s + 1); that stage was last read by chunk s - 1
        __syncthreads();                                   // V(s + 1) complete; every wave is done with V(s)
        ++s;
    };
    for (; k < n_my; ++k) {
        chunk(std::true_type{}, 0);
        for (int ch = 1; ch < nchunks; ++ch) chunk(std::false_type{}, ch);
        // ---------------------------------------------------------------- item epilogue
        if (IPDM_WUP2_KO & 4) {      // (timing only: one dword per lane keeps the accumulators alive)
            float ks = 0.0f;
#pragma unroll
            for (int e = 0; e < 9; ++e)
#pragma unroll
                for (int r = 0; r < 16; ++r) ks += acc[e][r];
            a.out[(size_t)blockIdx.x * 512 + tid] = ks;
            cur = ItemId{g_n, g_oy, g_ox, g_co * BN, g_a};
            continue;
        }
        // + bias through position (1, 1), whose products go to all four outputs with coefficient 1
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(nb, 1.0f, acc[4], 0, 0, 0);
        if (k + 1 < n_my) fetch_bias(g_co * BN);            // (describe(k + 1) ran two chunks ago)
        const ItemId t = cur;
        cur = ItemId{g_n, g_oy, g_ox, g_co * BN, g_a};
        const int par = t.a * 2 + bq;
        const size_t sample = (size_t)t.n * a.Cout * out_plane;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + sample), 0, a.Cout * out_plane * 4, 0x00020000);
        // The lanes l and l + 16 (patch columns 2m, 2m + 1) hold 4 consecutive pixels of a plane row between them.  Registers are
        // taken in pairs (couts c, c + 1): v_permlane16_swap exchanges one half each, after which the lane of the even column
        // owns the 4 pixels of cout c and the other one those of cout c + 1 -- 16-byte stores.
        const int px4 = t.ox0 + 4 * txh;
        const bool ragged = t.ox0 + TW > a.W && (a.W & 3) != 0;                  // (wave-uniform) a run straddles the right edge
        const bool clipped = t.oy0 + TH > a.H || t.ox0 + TW > a.W;               // (wave-uniform) some lanes own no pixels
        const int nval = a.W - px4;                                               // ... then it has 1..3 pixels
        const int so0 = ((t.co0 + hq * 32) * out_plane + par * HW + t.oy0 * a.W + t.ox0) * 4;
        bool rok[2], part[2];
        int lane_off4[2], voff4[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            rok[u] = t.oy0 + 2 * ty + u < a.H;
            lane_off4[u] = ((lk * 4 + odd) * out_plane + (2 * ty + u) * a.W + 4 * txh) * 4;
            voff4[u] = (rok[u] && px4 + 3 < a.W) ? lane_off4[u] : OOB;           // all four pixels of the lane's run
            part[u] = ragged && rok[u] && nval > 0 && nval < 4;
        }
        float *sb = xw;                                      // statistics staging: the wave's scratch is idle here
        f32x4 ko_sum = {0.0f, 0.0f, 0.0f, 0.0f};
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 2 * i, g = i >> 1, uu = 2 * (i & 1);
#define IPDM_M(e) f32x2{acc[e][r], acc[e][r + 1]}
            // Y[u][v] = sum of M[p][q] over p in {u, u + 1}, q in {v, v + 1}: columns first
            const f32x2 T00 = IPDM_M(0) + IPDM_M(1), T01 = IPDM_M(1) + IPDM_M(2);
            const f32x2 T10 = IPDM_M(3) + IPDM_M(4), T11 = IPDM_M(4) + IPDM_M(5);
            const f32x2 T20 = IPDM_M(6) + IPDM_M(7), T21 = IPDM_M(7) + IPDM_M(8);
#undef IPDM_M
            const f32x2 Y[2][2] = {{T00 + T10, T01 + T11}, {T10 + T20, T11 + T21}};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // rows r (even patch column) and r + 1 (odd) of the DPP row pair: the even one gives its cout c + 1 and takes the
                // odd one's cout c  (inline asm, s_nop: conv_wino2.hip)
                float ya0 = Y[u][0][0], yb0 = Y[u][0][1], ya1 = Y[u][1][0], yb1 = Y[u][1][1];
                asm("s_nop 1\n\t"
                    "v_permlane16_swap_b32 %0, %1\n\t"
                    "v_permlane16_swap_b32 %2, %3"
                    : "+v"(ya0), "+v"(yb0), "+v"(ya1), "+v"(yb1));
                // (four scalars, not a vector indexed element by element: this compiler stored component 0 of such a vector for
                //  every element of the ragged run below -- tools/experiments/dbg_up2.py, NOTEBOOK.md round 5)
                float v[4] = {ya0, ya1, yb0, yb1};
                const int so = so0 + (8 * g + uu) * plane4;
                if (IPDM_WUP2_KO & 32) {      // (timing only: ONE store per item, of the sum of all results, keeps the transform alive)
                    ko_sum += f32x4{v[0], v[1], v[2], v[3]};
                    if (i == 7 && u == 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ko_sum), o_rsrc, voff4[u], so, 0);
                    continue;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), o_rsrc, voff4[u], so, 0);
                if (ragged) {            // (wave-uniform) the run that straddles the edge: element by element, by its lane
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int vo1 = (part[u] && e < nval) ? lane_off4[u] + 4 * e : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), o_rsrc, vo1, so, 0);
                        if (part[u]) v[e] = e < nval ? v[e] : 0.0f;
                    }
                }
                if (a.stats && !(IPDM_WUP2_KO & 64)) {
                    // fused GroupNorm statistics of the output: one row of per-cout {sum, sum of squares} per PLANE ROW and 32-pixel
                    // column block, as conv_ws.hip's parity form writes them; the 8 lanes of one patch row in a DPP row share a cout
                    float s1 = (v[0] + v[1]) + (v[2] + v[3]);
                    float s2 = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
                    if (clipped) {
                        const bool ok = voff4[u] != OOB || part[u];
                        s1 = ok ? s1 : 0.0f; s2 = ok ? s2 : 0.0f;
                    }
                    asm("s_nop 1\n\t"
                        "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                        "v_add_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 0\n\t"
                        "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                        "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 0\n\t"
                        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf"
                        : "+v"(s1), "+v"(s2));
                    if (txh == 0) *reinterpret_cast<f32x2 *>(sb + ((u * 2 + ty) * 32 + 8 * g + uu + odd + 4 * lk) * 2) = f32x2{s1, s2};
                }
            }
        }
        if (a.stats) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row_y = t.oy0 + 2 * lk + u;       // lanes 0-31: patch row 0, lanes 32-63: patch row 1; cout = l31
                if (row_y < a.H) {
                    float *dst = a.stats + (((size_t)t.n * a.stats_rows + (size_t)(par * a.H + row_y) * a.tiles_x + t.ox0 / TW) * a.Cout + t.co0 + hq * 32 + l31) * 2;
                    *reinterpret_cast<f32x2 *>(dst) = *reinterpret_cast<const f32x2 *>(sb + (u * 64 + lane) * 2);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // (one wait state between the item's last 16-byte stores and whatever writes their data registers next: gfx950 needs it for
        //  a buffer store with an SGPR soffset too -- NOTEBOOK.md round 5; tools/check_store_hazard.py scans for the pair)
        asm volatile("s_nop 7");
    }
}

}  // namespace

namespace ipdm {

bool conv_wup2_shape_ok(int Cout, int Cin)
{
    return Cout % BN == 0 && Cin % KC == 0 && Cin >= 2 * KC;
}

// a rule of the layer alone (another summation order than the 2x2-tap parity form: the choice must not look at the batch)
bool conv_wup2_eligible(const ConvArgs &a)
{
    return conv_up2_eligible(a) && a.w_wup2 && !opt(OPT_CONV_NO_WUP2) && conv_wup2_shape_ok(a.Cout, a.C1);
}

// [Cin/8][Cout/128][a][b][cout quarter][position (p, q)][k parity][cout 32][k step]: U = G g_ab G^T of the parity's 2x2 filter
// g_ab (conv_pack_weights_up2's sums), G = [[1, 0], [1, 1], [0, 1]], in double, rounded once; column parity 1 reads the shared
// value X2 - X1 where its algorithm has X1 - X2: the sign lives here (q = 0 of b = 1 negated)
void conv_pack_weights_wup2(const float *w, int Cout, int Cin, std::vector<float> &packed)
{
    packed.assign((size_t)Cin * Cout * 36, 0.0f);
    const int co_tiles = Cout / BN;
    static const double Gm[3][2] = {{1, 0}, {1, 1}, {0, 1}};
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float *wk = w + ((size_t)co * Cin + ci) * 9;
            for (int pa = 0; pa < 2; ++pa)
                for (int pb = 0; pb < 2; ++pb) {
                    double g[2][2];
                    for (int i = 0; i < 2; ++i)
                        for (int j = 0; j < 2; ++j) {
                            const int ky0 = pa == 0 ? (i == 0 ? 0 : 1) : (i == 0 ? 0 : 2), ky1 = pa == 0 ? (i == 0 ? 0 : 2) : (i == 0 ? 1 : 2);
                            const int kx0 = pb == 0 ? (j == 0 ? 0 : 1) : (j == 0 ? 0 : 2), kx1 = pb == 0 ? (j == 0 ? 0 : 2) : (j == 0 ? 1 : 2);
                            double acc = 0.0;
                            for (int ky = ky0; ky <= ky1; ++ky)
                                for (int kx = kx0; kx <= kx1; ++kx) acc += (double)wk[ky * 3 + kx];
                            g[i][j] = acc;
                        }
                    for (int p = 0; p < 3; ++p)
                        for (int q = 0; q < 3; ++q) {
                            double u = 0.0;
                            for (int i = 0; i < 2; ++i)
                                for (int j = 0; j < 2; ++j) u += Gm[p][i] * g[i][j] * Gm[q][j];
                            if (pb == 1 && q == 0) u = -u;
                            const int q8 = ci / 8, c8 = ci % 8, kp = c8 >> 1, lk = c8 & 1;
                            const int co_t = co / BN, hq = (co % BN) / 32, c32 = co % 32;
                            const size_t idx = ((((((((size_t)q8 * co_tiles + co_t) * 2 + pa) * 2 + pb) * 4 + hq) * 9 + (p * 3 + q)) * 2 + lk) * 32 + c32) * 4 + kp;
                            packed[idx] = (float)u;
                        }
                }
        }
}

// `orig`: the Upsample layer's arguments as the executor passes them (H x W = the up-sampled size), conv_wup2_eligible(orig)
int conv2d_wup2_launch(const ConvArgs &orig, hipStream_t st, int prof_cls)
{
    ConvArgs a = orig;
    a.w = orig.w_wup2; a.up2 = 1; a.upsample = 0; a.H = a.Ho = orig.Hs; a.W = a.Wo = orig.Ws; a.split_ws = nullptr; a.ksplit = 1;
    a.tiles_x = cdiv(a.W, TW); a.tiles_y = cdiv(a.H, TH); a.co_tiles = a.Cout / BN;
    IPDM_REQUIRE(conv_wup2_shape_ok(a.Cout, a.C1) && !a.C2 && !a.res && a.act == 0, "conv2d_wup2: not an eligible Upsample layer");
    IPDM_REQUIRE((long)a.Cout * 4 * a.H * a.W < (1L << 29) && (long)a.C1 * a.H * a.W < (1L << 29) && (long)a.C1 * a.Cout * 36 < (1L << 29),
                 "conv2d_wup2: per-sample tensor exceeds the 2 GiB buffer-addressing range");
    const long nitems = (long)a.tiles_x * a.tiles_y * a.co_tiles * a.B * 2;
    IPDM_REQUIRE(nitems < (1L << 31), "conv2d_wup2: too many items");
    IPDM_REQUIRE(!a.stats || a.stats_rows == 4 * a.H * a.tiles_x, "conv2d_wup2: statistics rows %d != %d", a.stats_rows, 4 * a.H * a.tiles_x);
    const int cus = device_cu_count();
    long G = nitems < cus ? nitems : cus;
    G = (G + 7) / 8 * 8;
    if (int rc = ensure_dynamic_lds((const void *)conv_wup2_kernel, LDS_BYTES)) return rc;
    const bool prof = prof_enabled();
    if (prof) prof_before(prof_cls, st);
    hipLaunchKernelGGL(conv_wup2_kernel, dim3((unsigned)G), dim3(512), LDS_BYTES, st, a, (int)nitems);
    // (executed flops: nine products per 2x2 outputs of each of the four parities = 9 multiply-adds per source pixel)
    if (prof) prof_after(prof_cls, 2.0 * a.B * a.H * a.W * (double)a.Cout * a.C1 * 9, st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

}  // namespace ipdm
