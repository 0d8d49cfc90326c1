// conv_wino3 (round 6, OPT-IN: option conv_bf16x3) -- conv_wino2.hip's kernel with the channel contraction of the sixteen
// Winograd-domain GEMMs moved from the f32 MFMA (v_mfma_f32_32x32x2_f32, which shares the SIMD's vector ALU) to the bf16 matrix pipe
// through an ERROR-FREE THREE-WAY SPLIT of both operands:
//
//     u = u1 + u2 + u3,  v = v1 + v2 + v3   (each term a bf16: the top 8 significant bits of what is left; 3 x 8 = the 24 bits of a float)
//     u v  ~=  u1 v1 + u1 v2 + u2 v1 + u2 v2 + u1 v3 + u3 v1          (the dropped products are below 2^-24 |u v|)
//
// six v_mfma_f32_32x32x16_bf16 per position and 16-channel chunk instead of eight v_mfma_f32_32x32x2_f32: the products are exact in
// float32, the accumulation is the matrix unit's float32.  NOT bit-identical to the f32 kernels (another summation order, another
// set of roundings): an alternative evaluation judged by the parity gates (tests: test_wino3_*; bench.py alt_modes), never the default.
//   * U is split when the weights are packed (conv_pack_weights_wino appends the image: [16-ch chunk][128-cout tile][xi 16][term 3]
//     [cout quarter 4][k half 2][cout 32][8 x bf16]): a lane's operand of one (position, term) is 16 contiguous bytes, L2 -> registers;
//   * V is split by the wave that transforms it (three truncations and two subtractions per value), stage image
//     [xi 16][term 3][k half 2][tile 32][8 x bf16] = 48 KB, double-buffered; the epilogue's exchange buffer aliases the stage the
//     tile's last chunk has just consumed (+ 16 KB between the two stages);
//   * everything else -- staging, GroupNorm+SiLU prologue, tile schedule, output transform, fused statistics -- is conv_wino2's.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifndef IPDM_WINO2_KO
#define IPDM_WINO2_KO 0             // compile-time timing knock-outs (tools/build_variants.sh; results are WRONG under a knock-out):
#endif                              // 1 no activation, 2 no input transform, 4 no output transform / stores, 8 no window loads, 16 no U loads,
                                    // 32 output transform and exchange kept, but NO residual loads, stores or statistics (the part of the epilogue
                                    // that could move under the next tile's first chunk: an upper bound on what deferring it can buy),
                                    // 64 no exchange (LDS round trip + barrier E; the in-lane form of the 16x16x4 idea would remove exactly this)
#ifndef IPDM_WINO2_STAGGER
#define IPDM_WINO2_STAGGER 0
#endif
#ifndef IPDM_WINO2_SGN
#define IPDM_WINO2_SGN 1            // 1: the GroupNorm scale / shift of the wave's two channels through SCALAR loads (two s_load_dwordx2 per chunk
#endif                              //    instead of four per-lane dword buffer loads: 4 of the 22 vector-memory instructions of a chunk) -- round 5 experiment
#ifndef IPDM_WINO2_NTSTORE
#define IPDM_WINO2_NTSTORE 0        // 1: the output runs with non-temporal stores -- round 5 experiment
#endif
#ifndef IPDM_WINO2_PKT
#define IPDM_WINO2_PKT 1            // 1: the input transform of the NCHW instantiations as sixteen packed-f32 adds (0: the scalar form; same bits)
#endif
#ifndef IPDM_WINO2_DEFER
#define IPDM_WINO2_DEFER 0          // 1: the eight 16-byte stores of an interior tile are issued under the NEXT tile's first MFMAs (round 5 experiment)
#endif
#ifndef IPDM_WINO2_SEL
#define IPDM_WINO2_SEL 1            // 1 (round 6): which of the wave's two channels a staging slot belongs to is a CONSTANT lane mask (slot u = lane + 64 j,
#endif                              //    channel u / 54: j = 1 always the second, j = 0 the second from lane 54 on; planar: 60) -- the select takes it as an
                                    //    immediate SGPR pair instead of a compare result the compiler keeps alive (and spills) for the kernel's lifetime
#ifndef IPDM_WINO2_BMASK
#define IPDM_WINO2_BMASK 1          // 1 (round 6): the border tiles' per-element zeroing as v_bfe_i32 + v_and_b32 on the lane's bit mask instead of eight
#endif                              //    v_cndmask on eight 64-bit SGPR masks recomputed per tile (sixteen SGPRs held through the chunk loop -> spills)
#ifndef IPDM_WINO2_PRIO
#define IPDM_WINO2_PRIO 0           // round 6 experiment: 1 = the wave's issue priority raised (s_setprio 2) over its 64 MFMAs of a chunk, 2 = over its
#endif                              //    staging / transform instead (the two waves of a SIMD share the vector ALU: who wins the issue slot when both are ready)
#ifndef IPDM_WINO3_DRAIN
#define IPDM_WINO3_DRAIN 0          // s_nop 15 statements (16 cycles each) between the last MFMA of a chunk and the transform: the last accumulate chain has left the matrix unit
#endif
#ifndef IPDM_WINO3_STAGGER
#define IPDM_WINO3_STAGGER 0        // 0 (shipped): every wave stages / transforms chunk s + 1 first and multiplies chunk s after.  1: waves 4-7 multiply first
                                    //    (x1.03 ... 1.05 faster; 3: every wave multiplies first; 4: the halves swapped).  With ANY wave in the multiply-first order and
                                    //    the input transform's packed add (IPDM_WINO3_PKFIX 0, below) the FIRST forward of a process came out wrong (one Winograd
                                    //    position of sixteen tiles of one workgroup tile, all couts, 1e-2 relative) in 30 ... 50 % of fresh processes on three of the
                                    //    boxes seen; never in later forwards, never with blocking launches, never in this order (0 of 60 processes).  The instruction
                                    //    is identified and replaced (PKFIX 2: 0 of 32 in the multiply-first order); this order stays for this round (every gate was run on it).
                                    //    NOTEBOOK.md round 6, tools/experiments/dbg_bf16x3_fwd.py, analyze_trace3.py.
#endif
#ifndef IPDM_WINO3_PKFIX
#define IPDM_WINO3_PKFIX 2          // how columns j = 2, 3 of the input transform are formed (transform_patch).  0: conv_wino2's packed add
#endif                              //    `v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]`, whose LOW result takes the HIGH half of b: the
                                    //    instruction behind the first-forward defect (IPDM_WINO3_STAGGER above) -- in the wrong forwards the low result of ONE such
                                    //    instruction is `a.lo + 0` instead of `a.lo - b.hi` in lanes 48-63 (tools/experiments/analyze_trace3.py recovers T2 to 1e-7 from
                                    //    the two outputs: profiles/r06v_analysis.log).  1: the same with a destination that is not a source: still wrong (5 of 10 fresh
                                    //    processes).  2 (shipped): two plain subtractions -- the same bits; multiply-first order with it: 0 of 32 beside 3/10 and 5/10
                                    //    (profiles/r06w_pkfix.log, r06y_stag_sub.log).  Why the packed form fails there is not known (it needs the wave's own MFMAs in flight, a process's
                                    //    first forward, asynchronous launches); conv_wino2.hip uses it in the stage-first order, where it has never been seen to fail.
#ifndef IPDM_WINO3_BBUF
#define IPDM_WINO3_BBUF 2
#endif
#ifndef IPDM_WINO3_PACK
#define IPDM_WINO3_PACK 0           // 1: V stored as dwords (the lane pair's two halfwords brought together by v_permlane32_swap) instead of 48 two-byte stores per lane
#endif
#ifndef IPDM_WINO3_DBG
#define IPDM_WINO3_DBG 0            // debugging arms: 1 = V stored as whole dwords (the lane pair's two halfwords merged through a shuffle), 16 = a barrier at the top of
#endif                              //    every chunk and in front of the epilogue, 32 = every chunk and the epilogue start with all loads landed; 512 ... 8192: below
#ifndef IPDM_CONV_STAMPS
#define IPDM_CONV_STAMPS 0          // `make stamps`: in-kernel s_memtime stamps of the phases (a stamped build changes what it measures)
#endif

namespace {

constexpr int KC = 16;                                 // channels per staged chunk: two MFMA sub-chunks of 8 (4 k-steps of 2)
constexpr int TH = 4, TW = 32, BN = 128;               // output pixels / couts of a workgroup tile
constexpr int NW = 8;                                  // waves
constexpr int U_CHUNK_FLOATS = 16 * 2 * 2 * 32 * 4;    // packed f32 weights of one (8-channel chunk, 64-cout tile): the image in front of ours
constexpr int U3_BLOCK_BYTES = 16 * 3 * 4096;          // bf16 x 3 weights of one (16-channel chunk, 128-cout tile): [xi][term][quarter 4][k half 2][cout 32][8 bf16]
constexpr int V3_STAGE_BYTES = 16 * 3 * 1024;          // a V stage: [xi 16][term 3][k half 2][tile 32][8 bf16] = 48 KB
constexpr int XPAD_BYTES = 16 * 1024;                  // between the two stages: the exchange buffer (64 KB) = the consumed stage + this
constexpr int XCH_FLOATS = NW * 8 * 64 * 4;            // per wave: 8 x (64 lanes x 16 bytes)
constexpr int SCR_OFF_FLOATS = (2 * V3_STAGE_BYTES + XPAD_BYTES) / 4;
static_assert(XCH_FLOATS * 4 == V3_STAGE_BYTES + XPAD_BYTES, "conv_wino3: the exchange buffer is one stage + the pad");
constexpr int XP = 40;                                 // scratch row pitch (34 window columns; planar: columns up to 39)
constexpr int XWAVE = 12 * XP + 64 * 4 + 8;            // per wave: 12 row segments (2 channels x 6 window rows) + a dump slot per lane
constexpr size_t LDS_BYTES = (size_t)(SCR_OFF_FLOATS + NW * XWAVE) * sizeof(float);
static_assert(LDS_BYTES <= 160 * 1024, "conv_wino3: LDS budget exceeded");
static_assert(128 <= XWAVE, "conv_wino3: the statistics staging aliases the wave's scratch");
__device__ __host__ constexpr int stage_off(int p) { return p ? V3_STAGE_BYTES + XPAD_BYTES : 0; }      // byte offset of V stage p

struct TileId { int n, oy0, ox0, co0, ks; };      // ks: slice of the K (input channel) range, ConvArgs::ksplit

__host__ __device__ constexpr int row_slot(int i) { return i ^ (i >> 1); }      // rows of the transform domain in the order 0, 1, 3, 2

__device__ inline TileId decode_tile(const ConvArgs &a, int item)
{
    TileId t;
    t.ks = item % a.ksplit;                            // the K slices of one tile are neighbours in the schedule
    const int tile = item / a.ksplit;
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

#ifndef IPDM_WINO2_RES_T
#define IPDM_WINO2_RES_T 1          // 1: layers WITHOUT a residual run an instantiation that issues no residual loads (0: the round-4 form, eight
#endif                              //    range-checked-away loads per tile and wave: a vector-memory instruction costs its issue slot whatever it fetches)
// RES: the layer adds a residual (conv2 of a ResidualBlock with an identity or launched shortcut)
template <bool PLANAR, bool RES>
__global__ void __launch_bounds__(512) conv_wino3_kernel(ConvArgs a, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *const ldsb = reinterpret_cast<char *>(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *const xw = lds + SCR_OFF_FLOATS + swave * XWAVE;      // the wave's scratch (statistics staging in the epilogue)

    // static tile schedule: the workgroups of one XCD take a contiguous run of tiles, slot rotated per round
    const int G = gridDim.x, per = G >> 3;
    const int local = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int rounds = (ntiles + G - 1) / G;
    auto tile_of = [&](int k) { return k * G + (local + 5 * k) % G; };
    const int n_my = rounds == 0 ? 0 : (tile_of(rounds - 1) < ntiles ? rounds : rounds - 1);
    const int Ctot = a.C1 + a.C2;
    // chunks per schedule item: the whole channel range, or one of a.ksplit equal slices of it (K split: the item's partial
    // sums go to slice ks of a.out = the split workspace, no bias / residual / statistics; conv_ws.hip's combine pass folds
    // the slices in ascending order)
    const int nchunks = Ctot / KC / a.ksplit;            // launcher: Ctot % (KC * ksplit) == 0, C1 % KC == 0, nchunks >= 2
    const int S = n_my * nchunks;
    const int plane_bytes = a.Hs * a.Ws * 4;
    if (S == 0) return;

    // =============================================================================== staging role
    // wave w: channels 2 w, 2 w + 1 of the 16-channel chunk, ALL six window rows of the tile (both tile rows: 6 x 34 values per
    // channel) -- every window element is loaded and activated once per workgroup (as (tile row, 4 channels) per wave, rows 2
    // and 3 of the window were staged twice: 3 loads and 12 activations per lane and chunk instead of 2 and 8)
    // (the bf16 x 3 image follows the f32 one of conv_pack_weights_wino; a.co_tiles counts 128-cout tiles)
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.w + (size_t)2 * (Ctot / KC) * 2 * a.co_tiles * U_CHUNK_FLOATS), 0, (Ctot / KC) * a.co_tiles * U3_BLOCK_BYTES, 0x00020000);
    // slot u = lane + 64 j  ->  (channel cc, row r, 4-float part); 108 of the 128 slots exist (planar x1: 120)
    int lconst[2], xoff[2], gnoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = lane + 64 * j, cc = u / 54, r = (u - cc * 54) / 9, part = u - cc * 54 - r * 9;
        const bool v = u < 108;
        lconst[j] = v ? (r * a.Ws + 4 * part) * 4 + cc * plane_bytes : OOB;
        xoff[j] = v ? (cc * 6 + r) * XP + (PLANAR ? 2 : 4) * part : 12 * XP + lane * (PLANAR ? 2 : 4);
        gnoff[j] = (v ? cc : 0) * 4;
    }
    // parity-planar x1 ([ch][row & 1][col & 1][H/2][W/2]): 10 lanes per window row (5 x 16 bytes per plane), the scratch row
    // de-interleaved [even window columns: 20][odd: 20] (conv_wino.hip has the reasoning)
    int lconstp[PLANAR ? 2 : 1], xoffp[PLANAR ? 2 : 1], gnoffp[PLANAR ? 2 : 1];
    const int h2 = a.Hs >> 1, w2 = a.Ws >> 1;
    if (PLANAR) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int u = lane + 64 * j, cc = u / 60, r = (u - cc * 60) / 10, part = u - cc * 60 - r * 10;
            const bool v = u < 120;
            const int px = part < 5 ? 1 : 0, py = (r + 1) & 1, yo = (r + 1) / 2 - 1;      // window row r = image row oy0 - 1 + r, oy0 even
            const int xo = px ? 4 * part - 1 : 4 * (part - 5);
            lconstp[PLANAR ? j : 0] = v ? (((py * 2 + px) * h2 + yo) * w2 + xo) * 4 + cc * plane_bytes : OOB;
            xoffp[PLANAR ? j : 0] = v ? (cc * 6 + r) * XP + (px ? 4 * part : 20 + 4 * (part - 5)) : 12 * XP + lane * 4;
            gnoffp[PLANAR ? j : 0] = (v ? cc : 0) * 4;
        }
    }
    // tile descriptors: issue side (g_*: the tile whose chunks are being LOADED), activation side (a_*: one chunk behind)
    int g_n = 0, g_co = 0, g_oy = 0, g_ox = 0, g_ks = 0;
    const float *g_src1 = a.x1, *g_src2 = a.x2 ? a.x2 : a.x1;
    bool g_bord = false;
    int vo[2] = {lconst[0], lconst[1]}, g_so = 0;                        // NCHW offsets of the tile being loaded
    int va[2] = {lconst[0], lconst[1]}, g_sa = 0;                        // the offsets the loads use (planar x1 / NCHW)
    int vop[PLANAR ? 2 : 1] = {}, g_sop = 0;
    unsigned g_vmp = 0xffu, a_vmp = 0xffu, g_vm = 0xffu, g_lsh = 0, a_vm = 0xffu, a_lsh = 0;
    bool a_bord = false;
    auto describe = [&](int k) __attribute__((always_inline)) {
        const TileId tl = decode_tile(a, tile_of(k));
        const int iy0 = tl.oy0 - 1, ix0 = tl.ox0 - 1;
        g_n = tl.n; g_co = tl.co0 / BN; g_oy = tl.oy0; g_ox = tl.ox0; g_ks = tl.ks;
        g_src1 = a.x1 + (size_t)tl.n * a.C1 * (plane_bytes / 4);
        g_src2 = a.x2 ? a.x2 + (size_t)tl.n * a.C2 * (plane_bytes / 4) : g_src1;
        g_bord = tl.oy0 - 1 < 0 || tl.ox0 - 1 < 0 || tl.oy0 + TH + 1 > a.H || tl.ox0 + TW + 1 > a.W;
        const int g_base = (iy0 * a.Ws + ix0) * 4;
        g_so = g_bord ? 0 : g_base;
#pragma unroll
        for (int j = 0; j < 2; ++j) vo[j] = lconst[j];
        if (PLANAR) {
            const int basep = ((tl.oy0 >> 1) * w2 + (tl.ox0 >> 1)) * 4;
            g_sop = g_bord ? 0 : basep;
            g_vmp = 0xffu;
#pragma unroll
            for (int j = 0; j < 2; ++j) vop[PLANAR ? j : 0] = lconstp[PLANAR ? j : 0];
            if (g_bord) {
                g_vmp = 0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int u = lane + 64 * j, cc = u / 60, r = (u - cc * 60) / 10, part = u - cc * 60 - r * 10;
                    const int c0 = part < 5 ? 8 * part : 8 * (part - 5) + 1;
                    const bool rowok = u < 120 && iy0 + r >= 0 && iy0 + r < a.H;
                    vop[PLANAR ? j : 0] = rowok ? lconstp[PLANAR ? j : 0] + basep : OOB;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = c0 + 2 * e, ix = ix0 + c;
                        g_vmp |= (rowok && ix >= 0 && ix < a.W && c < 34) ? 1u << (4 * j + e) : 0u;
                    }
                }
            }
        }
        if (g_bord) {
            g_vm = 0; g_lsh = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int u = lane + 64 * j, cc = u / 54, r = (u - cc * 54) / 9, part = u - cc * 54 - r * 9;
                const bool rowok = u < 108 && iy0 + r >= 0 && iy0 + r < a.H;
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
                // (a load of a border tile may straddle the end of an image row: the next row's pixels, or -- at the last row
                //  of the tensor -- dwords past num_records, which a raw buffer load range-checks one by one and returns as
                //  0: tools/ubench/oob_probe.hip; either way those elements are masked by g_vm)
            }
        }
        const bool starts_in_x1 = g_ks * nchunks * KC < a.C1;      // (a K slice may begin in the skip half of a concat: NCHW)
#pragma unroll
        for (int j = 0; j < 2; ++j) va[j] = (PLANAR && starts_in_x1) ? vop[PLANAR ? j : 0] : vo[j];
        g_sa = (PLANAR && starts_in_x1) ? g_sop : g_so;
    };
    struct Raw { f32x4 v[2]; float sc[2], sh[2]; bool planar; float ssc[2], ssh[2]; };      // (ssc / ssh: wave-uniform, IPDM_WINO2_SGN)
    Raw raw;
    const __amdgpu_buffer_rsrc_t gsc_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.act ? a.gn_scale : a.out), 0, a.act ? (a.B * Ctot + 64) * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t gsh_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.act ? a.gn_shift : a.out), 0, a.act ? (a.B * Ctot + 64) * 4 : 0, 0x00020000);
    // every iteration issues the same loads, needed or not (past the end of the stream they re-read chunks of the last tile:
    // valid addresses, results unused): with conditional issue the waitcnt pass has to assume the shortest path
    auto issue_raw = [&](int ch) __attribute__((always_inline)) {
        const int c0 = (g_ks * nchunks + ch) * KC;           // (g_*: the item whose chunks are being loaded)
        const bool from1 = c0 < a.C1;
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(from1 ? g_src1 : g_src2), 0, (from1 ? a.C1 : a.C2) * plane_bytes, 0x00020000);
        const int cb = ((from1 ? c0 : c0 - a.C1) + 2 * swave) * plane_bytes;
        raw.planar = PLANAR && from1;
        if (PLANAR && c0 == a.C1) {  // (uniform, once per tile) only x1 is stored parity-planar; the skip half of a concat is NCHW
#pragma unroll
            for (int j = 0; j < 2; ++j) va[j] = vo[j];
            g_sa = g_so;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            raw.v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, va[j], cb + g_sa, 0));
        const int gso = (g_n * Ctot + c0 + 2 * swave) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {                        // (the slot's channel: a per-lane offset; the planar slot map differs)
            const int go = (PLANAR && raw.planar) ? gnoffp[PLANAR ? j : 0] : gnoff[j];
            if (!IPDM_WINO2_SGN) {
                raw.sc[j] = bload(gsc_rsrc, go, gso);
                raw.sh[j] = bload(gsh_rsrc, go, gso);
            }
        }
        if (IPDM_WINO2_SGN && a.act) {          // (uniform address: s_load_dwordx2; the arrays carry a chunk of read-ahead past [B, Ctot])
            const int gi = __builtin_amdgcn_readfirstlane(g_n * Ctot + c0 + 2 * swave);
            const float *__restrict__ ps = a.gn_scale + gi, *__restrict__ ph = a.gn_shift + gi;
            raw.ssc[0] = ps[0]; raw.ssc[1] = ps[1];
            raw.ssh[0] = ph[0]; raw.ssh[1] = ph[1];
        }
    };
    // activate the 8 landed values, zero what lies outside the image, park them in the wave's scratch
    auto activate = [&]() __attribute__((always_inline)) {
        f32x2 d[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) { d[2 * j] = f32x2{raw.v[j][0], raw.v[j][1]}; d[2 * j + 1] = f32x2{raw.v[j][2], raw.v[j][3]}; }
        const bool pl = PLANAR && raw.planar;
        if (a_bord && !pl) {  // (uniform) undo the left-edge shift: {x0, x1, x2, x3} loaded from one pixel further right
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (a_lsh >> j & 1) { d[2 * j + 1] = f32x2{d[2 * j][1], d[2 * j + 1][0]}; d[2 * j] = f32x2{0.0f, d[2 * j][0]}; }
        }
        if (a.act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (IPDM_WINO2_SGN && IPDM_WINO2_SEL && !(e & 1)) {
                    if (e == 0) {      // slot j = 0: lanes 0..53 (planar x1: 0..59) hold the wave's first channel, the rest its second
                        const unsigned long long m = (unsigned long long)((PLANAR && raw.planar) ? 0xf0000000u : 0xffc00000u) << 32;
                        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(raw.sc[0]) : "v"(raw.ssc[0]), "v"(raw.ssc[1]), "s"(m));
                        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(raw.sh[0]) : "v"(raw.ssh[0]), "v"(raw.ssh[1]), "s"(m));
                    } else {           // slot j = 1: the second channel (the lanes without a slot park what they compute in their dump slot)
                        raw.sc[1] = raw.ssc[1];
                        raw.sh[1] = raw.ssh[1];
                    }
                } else if (IPDM_WINO2_SGN && !(e & 1)) {      // (the slot's channel: the first or the second of the wave's two)
                    const bool second = ((PLANAR && raw.planar) ? gnoffp[PLANAR ? (e >> 1) : 0] : gnoff[e >> 1]) != 0;
                    raw.sc[e >> 1] = second ? raw.ssc[1] : raw.ssc[0];
                    raw.sh[e >> 1] = second ? raw.ssh[1] : raw.ssh[0];
                }
                const f32x2 sc2 = {raw.sc[e >> 1], raw.sc[e >> 1]}, sh2 = {raw.sh[e >> 1], raw.sh[e >> 1]};
                d[e] = __builtin_elementwise_fma(d[e], sc2, sh2);
            }
            if (a.act == 2) {
                f32x2 ex[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) ex[e] = d[e] * -1.4426950408889634f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { ex[e][0] = __builtin_amdgcn_exp2f(ex[e][0]); ex[e][1] = __builtin_amdgcn_exp2f(ex[e][1]); }
#pragma unroll
                for (int e = 0; e < 4; ++e) ex[e] = ex[e] + 1.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { ex[e][0] = __builtin_amdgcn_rcpf(ex[e][0]); ex[e][1] = __builtin_amdgcn_rcpf(ex[e][1]); }
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = d[e] * ex[e];
            }
        }
        if (a_bord) {
            const unsigned vm = pl ? a_vmp : a_vm;
            if (IPDM_WINO2_BMASK) {
                // bit e of the lane's mask, sign-extended: 0 / ~0 (volatile: not to be hoisted into eight live registers).  On SCALARS: a
                // bit_cast of `d[e >> 1][e & 1]` in this unrolled loop was compiled as component 0 of the UPDATED vector for the odd
                // elements (round 6: 46 GPU tests red; the same compiler behaviour as NOTEBOOK.md round 5, `tools/experiments/dbg_up2.py`)
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = d[e >> 1][e & 1];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    int keep;
                    asm volatile("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(vm), "n"(e));
                    f[e] = __builtin_bit_cast(float, __builtin_bit_cast(int, f[e]) & keep);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = f32x2{f[2 * e], f[2 * e + 1]};
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e >> 1][e & 1] = (vm >> e & 1) ? d[e >> 1][e & 1] : 0.0f;
            }
        }
        if (pl) {            // every other window column: the de-interleaved half of the scratch row
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4 *>(xw + xoffp[PLANAR ? j : 0]) = f32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]};
        } else if (PLANAR) { // four consecutive columns into the de-interleaved row (one scratch format per instantiation)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float *dst = xw + xoff[j];
                dst[0] = d[2 * j][0]; dst[1] = d[2 * j + 1][0];
                dst[20] = d[2 * j][1]; dst[21] = d[2 * j + 1][1];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4 *>(xw + xoff[j]) = f32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]};
        }
    };

    // =============================================================================== multiplying role
    const int lk = lane >> 5, l31 = lane & 31;
    const int ih = swave & 1, hq = swave >> 1;             // which two rows of the transform domain, which cout quarter
    // the lane's tile inside the workgroup tile: row ty, column 2 txh + odd.  The two tiles of a column pair sit 16 lanes
    // apart (DPP rows r, r + 1), so that the epilogue's exchange of halves is one v_permlane16_swap per register pair
    const int odd = l31 >> 4, ty = l31 & 1, txh = (l31 & 15) >> 1;
    f32x16 acc[8];
    f32x4 ua[4][3];                                        // the A operands (the three bf16 terms of U) of FOUR positions: ring, reloaded three positions ahead
    constexpr int NB = IPDM_WINO3_BBUF;                    // V operand buffers: 2 = position e + 1 goes into the registers of e - 1, 3 = of e - 2 (two accumulate chains behind them)
    f32x4 bb[NB == 3 ? 4 : 2][3] = {};                     // the B operands (the three bf16 terms of V); with three buffers slot = position % 4 over 0, 1, 2, 3 -> ring of FOUR (8 % 4 == 0: static indices)
    const int out_plane = a.Ho * a.Wo;
    const int plane4 = out_plane * 4;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
    float nb = 0.0f;
    auto fetch_bias = [&](int co0) __attribute__((always_inline)) {
        nb = bload(b_rsrc, lk ? OOB : l31 * 4, (co0 + hq * 32) * 4);
    };
    const float sgn = ih == 0 ? 1.0f : -1.0f;
    // input transform: the wave transforms the (tile, channel) pairs of its own scratch -- lane map 16 tiles x 2 k-steps x 2
    // channel parities, so that the 32 lanes of an LDS store group write a 64-float span of the [tile][k-step] image at most
    // 2-way conflicted (free)
    // (lane = tile column w_tx, tile row w_ty, channel lk of the wave's two: channel 2 w + lk of the chunk = k step w & 3, k
    //  parity lk of sub-chunk w >> 2)
    const int w_tx = lane & 15, w_ty = (lane >> 4) & 1;
    const float *const xr = xw + (lk * 6 + 2 * w_ty) * XP + 2 * w_tx;
    const int v_slot = (w_tx & 1) * 16 + (w_tx >> 1) * 2 + w_ty;                    // MFMA lane of tile (row w_ty, column w_tx)
    // the lane's values: channel 2 w + lk of the chunk = k half w >> 2, element 2 (w & 3) + lk of the tile's eight: 2 bytes each
    const int v_lane_b = (swave >> 2) * 512 + v_slot * 16 + (2 * (swave & 3) + lk) * 2;      // bytes; + (xi * 3 + term) * 1024 (+ stage)
    float patch[16];
    f32x2 pp[4][2];                                                                  // the same patch as aligned column pairs (IPDM_WINO2_PKT)
    constexpr int pcol[4] = {0, PLANAR ? 2 : 1, PLANAR ? 1 : 2, 3};                  // register position of patch column c
    auto read_patch = [&]() __attribute__((always_inline)) {
        const float *const xrp = xr - w_tx;                                         // column pairs (tx, tx + 1) of both halves
        if (PLANAR) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                patch[4 * r] = xrp[r * XP]; patch[4 * r + 1] = xrp[r * XP + 1];
                patch[4 * r + 2] = xrp[r * XP + 20]; patch[4 * r + 3] = xrp[r * XP + 21];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 lo = *reinterpret_cast<const f32x2 *>(xr + r * XP), hi = *reinterpret_cast<const f32x2 *>(xr + r * XP + 2);
                if (IPDM_WINO2_PKT) { pp[r][0] = lo; pp[r][1] = hi; }
                else { patch[4 * r] = lo[0]; patch[4 * r + 1] = lo[1]; patch[4 * r + 2] = hi[0]; patch[4 * r + 3] = hi[1]; }
            }
        }
    };
    // B^T d B of the lane's 4x4 patch, each of the sixteen values split into its three bf16 terms (truncation: the top 8 significant
    // bits, then the top 8 of what is left, then the rest -- exact), into V stage `par`
    auto transform_patch = [&](int par) __attribute__((always_inline)) {
        float o[16];                                      // [row slot rs][j]
        if (!PLANAR) {
            f32x2 T[4][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(T[0][h]) : "v"(pp[0][h]), "v"(pp[2][h]));
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(T[1][h]) : "v"(pp[1][h]), "v"(pp[2][h]));
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(T[2][h]) : "v"(pp[2][h]), "v"(pp[1][h]));
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(T[3][h]) : "v"(pp[1][h]), "v"(pp[3][h]));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rs = row_slot(i);
                f32x2 o01, o23;
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(o01) : "v"(T[i][0]), "v"(T[i][1]));
#if IPDM_WINO3_PKFIX == 1      // the destination may not be one of the sources
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=&v"(o23) : "v"(T[i][1]), "v"(T[i][0]));
#elif IPDM_WINO3_PKFIX == 2    // two plain subtractions instead of the packed add that takes the HIGH half of its second source for its LOW result
                o23 = f32x2{T[i][1][0] - T[i][0][1], T[i][0][1] - T[i][1][1]};
#else
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(o23) : "v"(T[i][1]), "v"(T[i][0]));
#endif
                o[rs * 4 + 0] = o01[0]; o[rs * 4 + 1] = o01[1]; o[rs * 4 + 2] = o23[0]; o[rs * 4 + 3] = o23[1];
            }
        } else {
            float tt[16];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                tt[0 + x] = patch[0 + x] - patch[8 + x];
                tt[4 + x] = patch[4 + x] + patch[8 + x];
                tt[8 + x] = patch[8 + x] - patch[4 + x];
                tt[12 + x] = patch[4 + x] - patch[12 + x];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rs = row_slot(i);
                o[rs * 4 + 0] = tt[i * 4 + pcol[0]] - tt[i * 4 + pcol[2]];
                o[rs * 4 + 1] = tt[i * 4 + pcol[1]] + tt[i * 4 + pcol[2]];
                o[rs * 4 + 2] = tt[i * 4 + pcol[2]] - tt[i * 4 + pcol[1]];
                o[rs * 4 + 3] = tt[i * 4 + pcol[1]] - tt[i * 4 + pcol[3]];
            }
        }
        if (IPDM_WINO3_PACK) {
            // Two positions at a time: the lanes l (channel 2 w) and l + 32 (channel 2 w + 1) hold the two halfwords of one dword of the tile's
            // operand cell.  v_permlane32_swap puts both halfwords of position x into the lower lane and both of position x + 1 into the upper
            // one; v_perm_b32 packs them: 24 dword stores per lane and chunk instead of 48 halfword ones.
            char *vdp = ldsb + stage_off(par) + (swave >> 2) * 512 + v_slot * 16 + (swave & 3) * 4 + (lk ? 3 * 1024 : 0);
#pragma unroll
            for (int x = 0; x < 16; x += 2) {
                unsigned ta[3], tb[3];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float v = o[x + q];
                    const unsigned b1 = __builtin_bit_cast(unsigned, v) & 0xffff0000u;
                    const float r1 = v - __builtin_bit_cast(float, b1);
                    const unsigned b2 = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
                    const float r2 = r1 - __builtin_bit_cast(float, b2);
                    unsigned *t = q ? tb : ta;
                    t[0] = b1; t[1] = b2; t[2] = __builtin_bit_cast(unsigned, r2);
                }
                // (s_nop 1: the swap reads need two wait states after the VALU writes; after it ta = the even channel's term, tb = the odd one's,
                //  of position x in lanes 0-31 and of position x + 1 in lanes 32-63)
                asm("s_nop 1\n\t"
                    "v_permlane32_swap_b32 %0, %3\n\t"
                    "v_permlane32_swap_b32 %1, %4\n\t"
                    "v_permlane32_swap_b32 %2, %5"
                    : "+v"(ta[0]), "+v"(ta[1]), "+v"(ta[2]), "+v"(tb[0]), "+v"(tb[1]), "+v"(tb[2]));
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    *reinterpret_cast<unsigned *>(vdp + (x * 3 + t) * 1024) = __builtin_amdgcn_perm(tb[t], ta[t], 0x07060302u);
            }
            return;
        }
        char *vd = ldsb + stage_off(par) + v_lane_b;
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            const unsigned b1 = __builtin_bit_cast(unsigned, o[x]) & 0xffff0000u;
            const float r1 = o[x] - __builtin_bit_cast(float, b1);
            const unsigned b2 = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
            const float r2 = r1 - __builtin_bit_cast(float, b2);
            const unsigned b3 = __builtin_bit_cast(unsigned, r2);
            if (IPDM_WINO3_DBG & 1) {      // lanes l (even channel) and l + 32 (odd channel) hold the two halves of one dword: the lower lane stores it
                const unsigned t[3] = {b1, b2, b3};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const unsigned other = (unsigned)__shfl_xor((int)t[q], 32, 64);
                    if (lk == 0) *reinterpret_cast<unsigned *>(vd + (x * 3 + q) * 1024) = (t[q] >> 16) | (other & 0xffff0000u);
                }
            } else {
                *reinterpret_cast<unsigned short *>(vd + (x * 3 + 0) * 1024) = (unsigned short)(b1 >> 16);
                *reinterpret_cast<unsigned short *>(vd + (x * 3 + 1) * 1024) = (unsigned short)(b2 >> 16);
                *reinterpret_cast<unsigned short *>(vd + (x * 3 + 2) * 1024) = (unsigned short)(b3 >> 16);
            }
        }
    };
    const int u_voff = (8 * ih) * 3 * 4096 + hq * 1024 + lk * 512 + l31 * 16;         // bytes; + (e * 3 + term) * 4096 + (16-channel chunk, 128-cout tile) block
    const int b_off_b = (8 * ih) * 3 * 1024 + lk * 512 + l31 * 16;                    // bytes; + (e * 3 + term) * 1024 (+ stage)
    int w_co = 0, w_q0 = 0;                                // cout tile / first 8-channel chunk of the item whose weights are being loaded
    // the three terms of position e of 16-channel chunk q16 of the item (w_co, w_q0), into ring slot e & 3
    auto issue_u = [&](int e, int q16) __attribute__((always_inline)) {
        const int w_soff = ((w_q0 + q16) * a.co_tiles + w_co) * U3_BLOCK_BYTES + e * 3 * 4096;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            ua[e & 3][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, u_voff, w_soff + t * 4096, 0));
    };

    // ---------------------------------------------------------------- prologue: tile 0, chunk 0 staged, chunk 1 in flight
    describe(0);
    a_vm = g_vm; a_vmp = g_vmp; a_lsh = g_lsh; a_bord = g_bord;
    issue_raw(0);
    if (ih == 0) fetch_bias(g_co * BN);
    activate();
    issue_raw(1);
    read_patch();
    transform_patch(0);
    __syncthreads();

    int s = 0;                                             // running chunk index (V stage parity)
    int k = 0;
    // deferred stores (IPDM_WINO2_DEFER): the finished 4-pixel runs of the previous tile, stored two per position under the first
    // sixteen MFMAs of this tile -- their registers are the ones the accumulators started last will take
    // (the stores are issued UNCONDITIONALLY, with an out-of-range offset when there is nothing to store: a conditionally issued
    //  memory operation makes the wait-count pass assume the shorter path, and every wait behind it comes out too strict)
    int d_voff = OOB;
    f32x4 d_v[8];
    size_t d_sample = 0;
    int d_so0 = 0;
    const int d_lane_off4 = ((lk * 4 + odd) * out_plane + (2 * ty + ih) * a.Wo + 4 * txh) * 4;
    typedef unsigned du32x4 __attribute__((ext_vector_type(4)));
    auto deferred_store = [&](int i) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + d_sample), 0, a.Cout * out_plane * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(du32x4, d_v[i]), rs, d_voff, d_so0 + (8 * (i >> 1) + 2 * (i & 1)) * plane4, 0);
    };
    const bool stamp = IPDM_CONV_STAMPS && (a.dbg & 8) != 0;
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0;      // stage / multiply / transform / barrier / epilogue
    const unsigned long long st_begin = stamp ? __builtin_amdgcn_s_memtime() : 0;
#define IPDM_STAMP(slot) if (stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[slot] += now_ - st_t; st_t = now_; __builtin_amdgcn_sched_barrier(0); }
    if (stamp) st_t = st_begin;
    TileId cur = {g_n, g_oy, g_ox, g_co * BN, g_ks};
    // One chunk: stage chunk s + 1 (activation -> scratch -> patch), multiply chunk s, transform chunk s + 1 into the other V
    // stage, barrier.  The first chunk of a tile STARTS its accumulators (C = 0 in the first MFMA of each), so that they are
    // dead from the output transform to the next tile.
    // Waves w and w + 4 share a SIMD.  Both running the same program in lockstep would stage together (the matrix pipe idle)
    // and then compete for it; waves 4-7 therefore do their staging BETWEEN the two sub-chunks, under their partners' MFMAs,
    // and multiply while the partners stage at the chunk boundary (IPDM_WINO2_STAGGER, NOTEBOOK.md).
    const bool late = (IPDM_WINO2_STAGGER == 1 && swave >= 4) || IPDM_WINO2_STAGGER == 2;
    auto chunk = [&](auto first, int ch) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value;
        const bool more1 = s + 1 < S;
        const int ch1 = ch + 1 == nchunks ? 0 : ch + 1, ch2 = ch1 + 1 == nchunks ? 0 : ch1 + 1;
        auto stage_next = [&]() __attribute__((always_inline)) {
            if (more1) {
                if (ch == nchunks - 1) { a_vm = g_vm; a_vmp = g_vmp; a_lsh = g_lsh; a_bord = g_bord; }      // chunk s + 1 opens the tile described last
                if (!(IPDM_WINO2_KO & 1)) activate();          // raw(s + 1) -> scratch
            }
            if (ch == nchunks - 2 && k + 1 < n_my) describe(k + 1);      // before the first loads of the next tile
            if (!(IPDM_WINO2_KO & 8)) issue_raw(ch2);          // raw(s + 2), consumed one iteration from now
        };
        // Every wave stages and transforms chunk s + 1 first and multiplies chunk s after (V(s + 1) goes into the stage chunk s - 1 read).
        // IPDM_WINO3_STAGGER 1 lets the two waves of a SIMD (w and w + 4) take the chunk in OPPOSITE order -- the bf16 matrix pipe is a unit
        // of its own, so one wave's MFMAs run under the other's activation / transform / split instructions: x1.03 ... 1.05 -- and is NOT
        // shipped yet: with any wave multiplying first, a process's first forward came out wrong on some boxes through one packed add of the
        // input transform (the flags' comments above; that instruction is gone from this kernel, the order waits for more runs).
        const bool early = IPDM_WINO3_STAGGER == 3 ? false : IPDM_WINO3_STAGGER == 4 ? swave >= 4 : (!IPDM_WINO3_STAGGER || swave < 4);      // (3: every wave multiplies first; 4: the halves swapped -- experiments)
        auto stage_all = [&]() __attribute__((always_inline)) {
            stage_next();
            if (!(IPDM_WINO2_KO & 2)) { read_patch(); transform_patch((s + 1) & 1); }
        };
        if (FIRST) {
            // the U terms of the tile's first three positions: NOT prefetched across the epilogue of the tile before (the ring's 36 registers
            // are free there: with them held, the instantiations that add a residual spilled 52 vector registers and lost the kernel's gain);
            // waves 0-3 cover the latency with their staging, waves 4-7 pay it once per tile
            w_co = cur.co0 / BN; w_q0 = cur.ks * nchunks;
#pragma unroll
            for (int e = 0; e < ((IPDM_WINO3_DBG & 2048) ? 4 : 3); ++e) issue_u(e, 0);
        }
        if (IPDM_WINO3_DBG & 32) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (debugging arm: every chunk starts with all loads landed)
        if (IPDM_WINO3_DBG & 16) __syncthreads();                                                   // (debugging arm: a barrier at the top of every chunk ...)
        constexpr bool OLD = (IPDM_WINO3_DBG & 4096) != 0;      // (debugging arm: the FIRST version's structure -- stage at the start, patch read at e = 4, transform at the end)
        if (OLD) stage_next();
        else if (early) stage_all();
        IPDM_STAMP(0)
        const char *stage = ldsb + stage_off(s & 1);
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        // ORDER OF THE LOOP (round 6; tools/experiments/dbg_wino3.py, NOTEBOOK.md).  With conv_wino2's order -- the next position's V terms read
        // from LDS in front of a position's MFMAs, U reloaded in place right behind them -- this kernel produced run-dependent garbage in one
        // output row of a tile in ~1 of 100 tiles; worse the longer a position's MFMA group took, gone with every accumulate chain drained
        // behind its position.  The suspected mechanism (a queued v_mfma_f32_32x32x16_bf16 reading its sources after a later write to them) does
        // NOT reproduce in isolation (tools/ubench/mfma_src_window.hip), and the old orders switched back on here (IPDM_WINO3_DBG 512 / 2048 / 4096)
        // do not bring it back: the cause is neither identified nor isolated.  The order kept is the conservative one -- nothing is loaded into
        // the operand registers of position e before the six MFMAs of position e + 1 have been issued: the V terms of position e + 1 and the U
        // terms of position e + 3 (ring of four) go into the registers of position e - 1 BEHIND the MFMAs of e; every reload is preceded by an
        // empty asm use of the old contents (the registers are not handed to anything in between), and the last position's registers stay
        // reserved to the chunk's barrier.  0 bad rows in 72 launches x 12 shapes, 60 launches x 6 shapes bit-equal; the mode is opt-in.
#pragma unroll
        for (int t = 0; t < 3; ++t) bb[0][t] = *reinterpret_cast<const f32x4 *>(stage + b_off_b + t * 1024);
        constexpr int BM = NB == 3 ? 3 : 1;               // slot mask
        if (NB == 3) {                                     // ... and position 1 (into the slot of the previous chunk's position 5)
#pragma unroll
            for (int t = 0; t < 3; ++t) bb[1][t] = *reinterpret_cast<const f32x4 *>(stage + b_off_b + (3 + t) * 1024);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (OLD && e == 4) read_patch();
            if (IPDM_WINO3_DBG & 512) {      // (debugging arm: the FIRST version's order for the V terms -- position e + 1 read in front of the MFMAs of e)
                if (e + 1 < 8) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) bb[(e + 1) & 1][t] = *reinterpret_cast<const f32x4 *>(stage + b_off_b + ((e + 1) * 3 + t) * 1024);
                }
                if (IPDM_WINO3_DBG & 1024) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, ua[e & 3][0]), a1 = __builtin_bit_cast(bf16x8, ua[e & 3][1]), a2 = __builtin_bit_cast(bf16x8, ua[e & 3][2]);
            const bf16x8 v0 = __builtin_bit_cast(bf16x8, bb[e & BM][0]), v1 = __builtin_bit_cast(bf16x8, bb[e & BM][1]), v2 = __builtin_bit_cast(bf16x8, bb[e & BM][2]);
            // the small products first
            if (FIRST) {
                const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, v2, zero, 0, 0, 0);
                // (with C = 0 the accumulator is dead in front of this instruction, and this compiler placed the B operand that dies
                //  here INSIDE the destination's sixteen registers -- `v_mfma_f32_32x32x16_bf16 v[64:79], v[96:99], v[76:79], 0` --:
                //  the f32 MFMA's destination is early-clobber in LLVM, this gfx950 instruction's is not: keep the operand alive past it)
                asm volatile("" ::"v"(bb[e & BM][2]), "v"(ua[e & 3][0]));
            } else {
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, v2, acc[e], 0, 0, 0);
            }
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, v0, acc[e], 0, 0, 0);
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, v1, acc[e], 0, 0, 0);
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, v1, acc[e], 0, 0, 0);
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, v0, acc[e], 0, 0, 0);
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, v0, acc[e], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // behind them: the V terms of position e + 1 into the registers of position e - 1 ...
            // (every reload is preceded by an empty asm USE of the registers' old contents: between the MFMAs that read them last and
            //  this point the allocator must not hand them to a temporary -- a VALU write there lands inside the window in which the
            //  queued MFMA still reads them; that was the rare residue after the reordering alone)
            if (NB == 3) {                                     // position e + 2 into the registers of position e - 2
                asm volatile("" ::"v"(bb[(e + 2) & 3][0]), "v"(bb[(e + 2) & 3][1]), "v"(bb[(e + 2) & 3][2]));
                if (e + 2 < 8) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) bb[(e + 2) & 3][t] = *reinterpret_cast<const f32x4 *>(stage + b_off_b + ((e + 2) * 3 + t) * 1024);
                }
            } else if (!(IPDM_WINO3_DBG & 512)) {
                // (e = 7 too, although nothing is loaded behind the last position: the registers of position 6 are DEAD behind its MFMAs, and
                //  without this use the allocator handed one of them to a temporary three instructions behind the chain's last MFMA -- a
                //  `v_cndmask_b32 v196, 0, 1, s[74:75]` for the loop's own `ch + 1 < nchunks` -- in every instantiation.  Whether an MFMA that
                //  waits in the matrix unit for its predecessor's accumulator has read its 128-bit B operand by then is not documented; the
                //  write is kept out of that window on principle -- tools/check_mfma_war.py scans the compiled code for such writes, the build
                //  is refused over one, IPDM_WINO3_DBG & 16384 brings this one back.  Closing it did NOT cure the first-forward defect of the
                //  multiply-first orders (8 of 24 fresh processes still wrong: profiles/r06l_fix.log).)
                if (e + 1 < 8 || !(IPDM_WINO3_DBG & 16384))
                    asm volatile("" ::"v"(bb[(e + 1) & 1][0]), "v"(bb[(e + 1) & 1][1]), "v"(bb[(e + 1) & 1][2]));
                if (e + 1 < 8) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) bb[(e + 1) & 1][t] = *reinterpret_cast<const f32x4 *>(stage + b_off_b + ((e + 1) * 3 + t) * 1024);
                }
            }
            // ... and the U terms of position e + 3 (this chunk's, or the next chunk's e - 5) into ring slot (e - 1) & 3
            if (IPDM_WINO3_DBG & 2048) {      // (debugging arm: the FIRST version's order for the U terms -- position e + 4 IN PLACE right behind the MFMAs of e)
                if (e < 4) issue_u(e + 4, ch);
                else if (ch + 1 < nchunks) issue_u(e - 4, ch + 1);
            } else {
                asm volatile("" ::"v"(ua[(e + 3) & 3][0]), "v"(ua[(e + 3) & 3][1]), "v"(ua[(e + 3) & 3][2]));
                if (e + 3 < 8) issue_u(e + 3, ch);
                else if (ch + 1 < nchunks) issue_u(e - 5, ch + 1);          // (the next TILE's first positions: at its first chunk)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // drain: the transform's temporaries (and the next chunk's staging) may take the registers of the last position's operands
#pragma unroll
        for (int d = 0; d < IPDM_WINO3_DRAIN; ++d) asm volatile("s_nop 15");
        __builtin_amdgcn_sched_barrier(0);
        IPDM_STAMP(1)
        if (OLD) transform_patch((s + 1) & 1);
        else if (!early) stage_all();
        // (the registers of the last position -- bb[1], ring slot 3 -- stay reserved up to here: its accumulate chain may still have been
        //  waiting in the matrix unit while the staging of waves 4-7 looked for temporaries; waves 0-3 go from their MFMAs to the barrier,
        //  where they wait for that staging -- an order of magnitude longer than a chain)
        if (!(IPDM_WINO3_DBG & 8192))
            asm volatile("" ::"v"(bb[NB == 3 ? 3 : 1][0]), "v"(bb[NB == 3 ? 3 : 1][1]), "v"(bb[NB == 3 ? 3 : 1][2]), "v"(ua[3][0]), "v"(ua[3][1]), "v"(ua[3][2]));
        if (NB == 3) asm volatile("" ::"v"(bb[2][0]), "v"(bb[2][1]), "v"(bb[2][2]));
        IPDM_STAMP(2)
        __syncthreads();                                   // V(s + 1) complete; every wave is done with V(s)
        IPDM_STAMP(3)
        ++s;
    };
    for (; k < n_my; ++k) {
        chunk(std::true_type{}, 0);
        for (int ch = 1; ch < nchunks; ++ch) chunk(std::false_type{}, ch);
        // ---------------------------------------------------------------- tile epilogue
        if (IPDM_WINO2_KO & 4) { cur = TileId{g_n, g_oy, g_ox, g_co * BN, g_ks}; continue; }
        // + bias through position (1, 1) (accumulator 5 of the ih = 0 waves), whose output coefficients are all 1
        if (ih == 0) {
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(nb, 1.0f, acc[5], 0, 0, 0);
            if (k + 1 < n_my) fetch_bias(g_co * BN);        // (describe(k + 1) ran two chunks ago)
        }
        const TileId t = cur;
        cur = TileId{g_n, g_oy, g_ox, g_co * BN, g_ks};
        const size_t sample = ((size_t)t.ks * a.B + t.n) * a.Cout * out_plane;      // (ks > 0 only with a.out = the split workspace [ksplit][B,Cout,Ho,Wo])
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + sample), 0, a.Cout * out_plane * 4, 0x00020000);
        // (no residual: zero records -- the loads below return 0 and the add stays unconditional)
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((a.res ? a.res : a.out) + sample), 0,
                                                                                   a.res ? a.Cout * out_plane * 4 : 0, 0x00020000);
        // The lanes l and l + 16 (tile columns 2m, 2m + 1) hold 4 consecutive pixels of the row between them.  Registers are
        // taken in pairs (couts c, c + 1): v_permlane16_swap exchanges one half each, after which the lane of the even column
        // owns the 4 pixels of cout c and the other one those of cout c + 1 -- 16-byte stores and residual loads.
        const int py = t.oy0 + 2 * ty + ih, px4 = t.ox0 + 4 * txh;
        const bool rok = py < a.Ho;
        const int lane_off4 = ((lk * 4 + odd) * out_plane + (2 * ty + ih) * a.Wo + 4 * txh) * 4;
        const int voff4 = (rok && px4 + 3 < a.Wo) ? lane_off4 : OOB;             // all four pixels of the lane's run
        const bool ragged = t.ox0 + TW > a.Wo && (a.Wo & 3) != 0;                // (wave-uniform) a run straddles the right edge
        const bool clipped = t.oy0 + TH > a.Ho || t.ox0 + TW > a.Wo;             // (wave-uniform) some lanes own no pixels
        const int nval = a.Wo - px4;                                              // ... then it has 1..3 pixels
        const bool part = ragged && rok && nval > 0 && nval < 4;
        const int so0 = ((t.co0 + hq * 32) * out_plane + min(t.oy0, a.Ho - 1) * a.Wo + t.ox0) * 4;
        // the exchange buffer: the stage the tile's last chunk has just consumed (s & 1 is the NEXT chunk's) plus the pad between the stages
        float *const xch = reinterpret_cast<float *>(ldsb + ((s & 1) ? 0 : V3_STAGE_BYTES));
        const float *xr2 = xch + ((swave ^ 1) * 8) * 256 + lane * 4;
        float *sb = xw;                                      // statistics staging: the wave's scratch is idle here
        const f32x2 sgn2 = {sgn, sgn};
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        if (IPDM_WINO3_DBG & 32) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (IPDM_WINO3_DBG & 16) __syncthreads();                                                   // (... and one in front of the epilogue)
        // ALL residual loads of the tile are issued together, ahead of the transform
        f32x4 rv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            rv[i] = ((IPDM_WINO2_KO & 32) || !RES) ? f32x4{0.0f, 0.0f, 0.0f, 0.0f}
                                         : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, voff4, so0 + (8 * (i >> 1) + 2 * (i & 1)) * plane4, 0));
        __builtin_amdgcn_sched_barrier(0);
        // columns first (in-lane): T_i[b] = sum_j M[i][j] A[j][b],  A^T = [[1,1,1,0],[0,1,-1,-1]]; then the wave's own two
        // rows (first = accumulators 0-3, second = 4-7; ih = 0: rows 0, 1, ih = 1: rows 3, 2 -- row_slot): both waves keep
        // K = first + second and send the second (T1 resp. T2); output row 0 = (T0 + T1) + T2, row 1 = T1 - (T2 + T3).
        f32x2 K0[8], K1[8];                                 // [pair]: output column 0 / 1 of {cout c, cout c + 1}
        {
            float *xo = xch + (swave * 8) * 256 + lane * 4;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int r = 2 * p;
#define IPDM_M(e) f32x2{acc[e][r], acc[e][r + 1]}
                const f32x2 lo0 = (IPDM_M(0) + IPDM_M(1)) + IPDM_M(2), lo1 = (IPDM_M(1) - IPDM_M(2)) - IPDM_M(3);
                const f32x2 hi0 = (IPDM_M(4) + IPDM_M(5)) + IPDM_M(6), hi1 = (IPDM_M(5) - IPDM_M(6)) - IPDM_M(7);
#undef IPDM_M
                K0[p] = lo0 + hi0;
                K1[p] = lo1 + hi1;
                if (!(IPDM_WINO2_KO & 64)) *reinterpret_cast<f32x4 *>(xo + p * 256) = f32x4{hi0[0], hi0[1], hi1[0], hi1[1]};
                else { K0[p] += hi0 * 0.5f; K1[p] += hi1 * 0.5f; }      // (timing only: keeps the values alive without the exchange)
            }
        }
        IPDM_STAMP(5)
        if (!(IPDM_WINO2_KO & 64)) __syncthreads();        // E: both halves of every (tile, cout) are in LDS
        IPDM_STAMP(6)
        f32x4 ko_sum = {0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 gotv[8];                                     // partner's {T[0] c, T[0] c+1, T[1] c, T[1] c+1} of every pair
#pragma unroll
        for (int i = 0; i < 8; ++i) gotv[i] = (IPDM_WINO2_KO & 64) ? f32x4{K0[i][0], K0[i][1], K1[i][0], K1[i][1]} : *reinterpret_cast<const f32x4 *>(xr2 + i * 256);
        __syncthreads();                                   // E2: everybody has READ the exchange buffer -- the next chunk's transform writes that stage
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int g = i >> 1, u = 2 * (i & 1);
            const f32x4 got = gotv[i];
            // ih = 0: K + T2 (output row 0);  ih = 1: T1 - K (output row 1)
            const f32x2 y0 = __builtin_elementwise_fma(K0[i], sgn2, f32x2{got[0], got[1]});       // column 0 of {cout c, c + 1}
            const f32x2 y1 = __builtin_elementwise_fma(K1[i], sgn2, f32x2{got[2], got[3]});       // column 1
            // rows r (even tile column) and r + 1 (odd) of the DPP row pair: the even one gives its cout c + 1 and takes the
            // odd one's cout c  (inline asm: through the builtin this compiler passed component 0 of y0 / y1 as BOTH operands
            // of the swap; s_nop: the swap reads need two wait states after the VALU writes)
            float ya0 = y0[0], yb0 = y0[1], ya1 = y1[0], yb1 = y1[1];
            asm("s_nop 1\n\t"
                "v_permlane16_swap_b32 %0, %1\n\t"
                "v_permlane16_swap_b32 %2, %3"
                : "+v"(ya0), "+v"(yb0), "+v"(ya1), "+v"(yb1));
            f32x4 v = {ya0, ya1, yb0, yb1};
            const int so = so0 + (8 * g + u) * plane4;
            v += rv[i];
            if (IPDM_WINO2_KO & 32) {      // (timing only: ONE store per tile, of the sum of all eight results, keeps the transform alive)
                ko_sum += v;
                if (i == 7) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ko_sum), o_rsrc, voff4, so, 0);
                continue;
            }
            // (IPDM_WINO2_DEFER: an interior tile -- wave-uniform: every lane owns its whole run -- keeps its runs for the next tile's
            //  first MFMAs; d_v is DEFINED on every path so that its live range ends at the deferred stores)
            if (IPDM_WINO2_DEFER) d_v[i] = v;
            if (!(IPDM_WINO2_DEFER && !clipped && !ragged)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, voff4, so, IPDM_WINO2_NTSTORE ? 2 : 0);
            if (ragged) {            // (wave-uniform) the run that straddles the edge: element by element, by its lane
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int vo1 = (part && e < nval) ? lane_off4 + 4 * e : OOB;
                    float x = v[e];
                    if (RES) x += bload(r_rsrc, vo1, so);          // (the lane's 16-byte residual load was out of range: + 0 above)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), o_rsrc, vo1, so, 0);
                    if (part) v[e] = e < nval ? x : 0.0f;
                }
            }
            if (a.stats) {
                // fused GroupNorm statistics of the output: one row of per-cout {sum, sum of squares} per PIXEL ROW and
                // 32-pixel column block, as conv_ws.hip writes them; the 8 lanes of one tile row in a DPP row share a cout
                float s1 = (v[0] + v[1]) + (v[2] + v[3]);
                float s2 = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
                if (clipped) {
                    const bool ok = voff4 != OOB || part;
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
                if (txh == 0) *reinterpret_cast<f32x2 *>(sb + (ty * 32 + 8 * g + u + odd + 4 * lk) * 2) = f32x2{s1, s2};
            }
        }
        if (a.stats) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int row_y = t.oy0 + 2 * lk + ih;          // lanes 0-31: tile row 0, lanes 32-63: tile row 1; cout = l31
            if (row_y < a.Ho) {
                float *dst = a.stats + (((size_t)t.n * a.stats_rows + (size_t)row_y * a.tiles_x + t.ox0 / TW) * a.Cout + t.co0 + hq * 32 + l31) * 2;
                *reinterpret_cast<f32x2 *>(dst) = *reinterpret_cast<const f32x2 *>(sb + lane * 2);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (IPDM_WINO2_DEFER && !clipped && !ragged) { d_voff = d_lane_off4; d_sample = sample; d_so0 = so0; }
        // (wait states between the tile's last 16-byte stores and whatever writes their data registers next: gfx950 needs ONE for a
        //  buffer store with an SGPR soffset too, the compiler inserts none -- NOTEBOOK.md round 5; tools/check_store_hazard.py scans
        //  every kernel of the library for the pair at build time)
        asm volatile("s_nop 7");
        IPDM_STAMP(4)
    }
    if (IPDM_WINO2_DEFER) {
#pragma unroll
        for (int i = 0; i < 8; ++i) deferred_store(i);
    }
    if (stamp && tid == 0) {
        unsigned long long *d = a.dbg_buf + (size_t)blockIdx.x * 8;
        d[0] = st_acc[1]; d[1] = st_acc[4]; d[2] = st_acc[3]; d[3] = __builtin_amdgcn_s_memtime() - st_begin;
        d[4] = st_acc[0]; d[5] = st_acc[2]; d[6] = st_acc[5]; d[7] = st_acc[6];
    }
#undef IPDM_STAMP
}

}  // namespace

namespace ipdm {

// The layers of conv_wino2's SHAPE that are not K-split, when the option asks for the bf16 x 3 evaluation.  A rule of the layer
// alone: this kernel's bits differ from the float32 Winograd kernels' (which are bit-identical to each other, so THEIR choice may
// look at the batch), and a slice's result must not depend on the batch it was sharded into -- so, unlike conv_wino2_eligible,
// no tile-count threshold here: under the option a lone slice runs these layers on 128-cout tiles too (slower for a lone slice
// on the 64-tile levels; identical bits at every batch size: tests/test_gpu_parity.py::test_wino3_batch_is_its_slices).
bool conv_wino3_eligible(const ConvArgs &a)
{
    if (opt(OPT_CONV_BF16X3) <= 0 || opt(OPT_WINO_V1) || a.ksplit > 1) return false;
    const int Ctot = a.C1 + a.C2;
    return a.Cout % BN == 0 && Ctot % KC == 0 && (!a.C2 || a.C1 % KC == 0) && Ctot >= 2 * KC;
}

// `prepared`: as for conv2d_wino2_launch (a.w = the Winograd-domain weights: the f32 image, followed by the bf16 x 3 one)
int conv2d_wino3_launch(const ConvArgs &prepared, hipStream_t st)
{
    ConvArgs a = prepared;
    a.co_tiles = a.Cout / BN;
    a.ksplit = 1;
    const long ntiles = (long)a.tiles_x * a.tiles_y * a.co_tiles * a.B;
    IPDM_REQUIRE(ntiles < (1L << 31), "conv2d_wino3: too many tiles");
    IPDM_REQUIRE((long)(a.C1 + a.C2) / KC * a.co_tiles * U3_BLOCK_BYTES < (1L << 31), "conv2d_wino3: weight image exceeds the buffer-addressing range");
    const int cus = device_cu_count();
    long G = ntiles < cus ? ntiles : cus;
    G = (G + 7) / 8 * 8;
    const bool res = a.res != nullptr;
    const void *fn = a.x1_planar ? (res ? (const void *)conv_wino3_kernel<true, true> : (const void *)conv_wino3_kernel<true, false>)
                                 : (res ? (const void *)conv_wino3_kernel<false, true> : (const void *)conv_wino3_kernel<false, false>);
    if (int rc = ensure_dynamic_lds(fn, LDS_BYTES)) return rc;
    if (a.x1_planar && res) hipLaunchKernelGGL((conv_wino3_kernel<true, true>), dim3((unsigned)G), dim3(512), LDS_BYTES, st, a, (int)ntiles);
    else if (a.x1_planar) hipLaunchKernelGGL((conv_wino3_kernel<true, false>), dim3((unsigned)G), dim3(512), LDS_BYTES, st, a, (int)ntiles);
    else if (res) hipLaunchKernelGGL((conv_wino3_kernel<false, true>), dim3((unsigned)G), dim3(512), LDS_BYTES, st, a, (int)ntiles);
    else hipLaunchKernelGGL((conv_wino3_kernel<false, false>), dim3((unsigned)G), dim3(512), LDS_BYTES, st, a, (int)ntiles);
    return IPDM_OK;
}

}  // namespace ipdm
