// Winograd F(2x2, 3x3) form of the wide 3x3 stride-1 convolutions on the exact-f32 MFMA (gfx950).
//
// The wide ResidualBlock convolutions (Model/model.py:100-117) are 57 % of a step and sit at the hard ceiling of the f32
// matrix pipe (157 TFLOP/s); the reference's own backend (cuDNN) evaluates such layers in the Winograd domain.  Here:
//
//     Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 multiply-adds per 2x2 outputs and (cin, cout) pair instead of 36: the contraction over channels is 16 independent
// GEMMs [tiles x cin] x [cin x cout], one per position xi = (i, j) of the 4x4 transform domain, on v_mfma_f32_32x32x2_f32.
// U = G g G^T is formed in double precision when the weights are packed and rounded once.
//
// Structure: the persistent, wave-specialised scheme of conv_ws.hip (one 512-thread workgroup per CU, static XCD-aware
// tile schedule, waves 4-7 stage the next K chunk while waves 0-3 multiply), with what Winograd changes:
//
//   * workgroup tile = 32 tiles (2 tile rows x 16 tile columns = 4 x 32 output pixels) x 64 couts.  Consumer wave w owns
//     EIGHT positions -- two rows of the 4x4 domain (ih = w & 1: rows 0, 1 or 3, 2 -- row_slot), all four j -- of one cout
//     half h = w >> 1 for all 32 tiles: 8 accumulators of 32x32 (128 registers).  One 16-byte LDS read per operand and
//     position feeds the four K steps of a chunk (layouts below): 0.5 LDS instructions per MFMA.  The first chunk of a tile
//     starts the accumulators (C = 0), so they are dead from the output transform to the next tile.
//   * producers (waves 4-7; wave pw: tile row pw & 1, four channels of the 8-channel chunk): per chunk and lane three
//     16-byte buffer loads of the wave's window rows (34 columns, 9 lanes per row; parity-planar x1: 10 lanes, two planes),
//     two chunks ahead; GroupNorm(+SiLU) ONCE per window element, beside the consumers' MFMAs, into an LDS scratch that
//     is double-buffered by chunk parity; the weights' LDS image by eight 16-byte loads/stores.  Every step issues the
//     same 13 loads (one explicit s_waitcnt vmcnt(13)).
//   * consumers read their 4x4 patches back from the scratch beside the MFMAs and do B^T d B themselves between two
//     chunks (32 VALU, 16 ds_write_b32 into the stage); ONE hand-over barrier per chunk.
//   * the output transform needs all 16 positions of a (tile, cout): in-lane over the wave's own 8 (j, then its two rows;
//     packed-f32), and ONE exchange of 2 values per (tile, cout) with the partner wave (w ^ 1) through LDS behind one barrier
//     per TILE: wave ih = 0 finishes output row 0 of every 2x2 tile, wave ih = 1 row 1.  The lanes of a tile-column pair sit
//     16 apart, so v_permlane16_swap gives every lane 4 consecutive pixels of one cout: 16-byte stores and residual loads.
//     The epilogue is built around LATENCY, not instruction count: the tile's origin comes from the producers through LDS
//     (no integer divisions in the MFMA waves), all 8 residual loads of a wave are in flight before the transform starts,
//     the 8 exchange reads are issued together.  Fused GroupNorm statistics with conv_ws.hip's geometry: one row of
//     per-cout partial sums per pixel row and 32-pixel column block.
//   * bias: one extra MFMA into position (1, 1), whose output-transform coefficients are 1 for all four outputs.
//
// LDS stage (48 KB, two stages): V [xi 16][lk 2][tile 32][kp 4] and U [xi 16][h 2][lk 2][cout 32][kp 4], channel of a
// value = 2 kp + lk: the four K steps of a lane are one ds_read_b128.  Plus 32 KB exchange, 2 KB statistics staging, 28 KB
// producer scratch, 32 bytes of tile descriptors.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifndef IPDM_CONV_STAMPS
#define IPDM_CONV_STAMPS 0          // `make stamps`: in-kernel s_memtime stamps of the phases (a stamped build changes what it measures)
#endif

namespace {

constexpr int KC = 8;                                  // channels per K chunk (4 MFMA k-steps of 2)
constexpr int TH = 4, TW = 32, BN = 64;                // output pixels / couts of a workgroup tile
constexpr int V_FLOATS = 16 * 2 * 32 * 4;
constexpr int U_FLOATS = 16 * 2 * 2 * 32 * 4;
constexpr int STAGE = V_FLOATS + U_FLOATS;             // 12288 floats = 48 KB
constexpr int XCH_FLOATS = 4 * 8 * 64 * 4;             // per consumer wave: 8 x (64 lanes x 16 bytes)
constexpr int STAT_FLOATS = 4 * 128;                   // per consumer wave: 2 pixel rows x 32 couts x {sum, sum of squares}
constexpr int XP = 40;                                 // scratch row pitch: 34 window columns, 16-byte aligned rows, and room for the
                                                       // stride-2 stores of a parity-planar source (columns up to 39)
constexpr int XWAVE = 16 * XP + 64 * 4 + 8;            // per producer wave: 16 row segments + a dump slot per lane (for the load
                                                       // slots that do not exist; + 8: the planar form stores 4 dwords at stride 2)
constexpr int XSCR_FLOATS = 2 * 4 * XWAVE;             // the producers' activated windows, double-buffered by chunk parity
constexpr int TDESC_FLOATS = 8;                        // {sample, cout origin, row origin, column origin} of the tiles k, k + 1 (by parity)
constexpr size_t LDS_BYTES = (size_t)(2 * STAGE + XCH_FLOATS + STAT_FLOATS + XSCR_FLOATS + TDESC_FLOATS) * sizeof(float);
static_assert(LDS_BYTES <= 160 * 1024, "conv_wino: LDS budget exceeded");
constexpr int U_CHUNK_FLOATS = U_FLOATS;               // packed weights of one (chunk, cout tile): the U stage image

struct TileId { int n, oy0, ox0, co0; };

// The stage images keep the rows i of the 4x4 transform domain in the order 0, 1, 3, 2: consumer wave ih then holds rows
// (0, 1) or (3, 2) in its accumulators 0-3 / 4-7, and in the output transform BOTH waves keep first + second and send the
// second (T1 resp. T2) -- no wave-dependent selects.
__host__ __device__ constexpr int row_slot(int i) { return i ^ (i >> 1); }

__device__ inline TileId decode_tile(const ConvArgs &a, int tile)
{
    TileId t;
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

// per-lane buffer offset that is out of range: loads return 0, stores are dropped (the scalar offset stays in range)
constexpr int OOB = 0x7fffffff;

__device__ inline float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

#define IPDM_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))
__device__ inline float sum_row16(float x)              // over the 16 lanes of the lane's DPP row; every lane gets the total
{
    x += IPDM_DPP_F(x, 0x121);                           // row_ror:1
    x += IPDM_DPP_F(x, 0x122);                           // row_ror:2
    x += IPDM_DPP_F(x, 0x124);                           // row_ror:4
    x += IPDM_DPP_F(x, 0x128);                           // row_ror:8
    return x;
}
#undef IPDM_DPP_F

// RES: the layer adds a residual; layers without one run an instantiation that issues no residual loads (round 5: a
// vector-memory instruction costs its issue slot whatever it fetches, conv_wino2.hip has the measurement)
template <bool PLANAR, bool RES>
__global__ void __launch_bounds__(512) conv_wino_kernel(ConvArgs a, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const xch = lds + 2 * STAGE;
    float *const stat_lds = xch + XCH_FLOATS;
    int *const tdesc = reinterpret_cast<int *>(lds + 2 * STAGE + XCH_FLOATS + STAT_FLOATS + XSCR_FLOATS);

    // static tile schedule of conv_ws.hip: the workgroups of one XCD take a contiguous run of tiles, slot rotated per round
    const int G = gridDim.x, per = G >> 3;
    const int local = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int rounds = (ntiles + G - 1) / G;
    auto tile_of = [&](int k) { return k * G + (local + 5 * k) % G; };
    const int n_my = rounds == 0 ? 0 : (tile_of(rounds - 1) < ntiles ? rounds : rounds - 1);
    const int Ctot = a.C1 + a.C2;
    const int nchunks = Ctot / KC;                       // launcher: Ctot % KC == 0, C1 % KC == 0
    const int S = n_my * nchunks;
    const int plane_bytes = a.Hs * a.Ws * 4;

    if (threadIdx.x >= 256) {
        // =========================================================================== PRODUCERS
        // Wave pw stages one tile row (16 tiles: 4 input rows x 34 columns) of FOUR channels of the chunk:
        //   pw & 1 = tile row, pw >> 1 = channel group.  Per chunk and lane, a three-stage software pipeline:
        //   G  (beside the consumers' MFMAs)  three 16-byte buffer loads: the wave's 16 row segments of 34 floats, 9 lanes
        //      per segment; the lane's channel is fixed (lane & 3), so its GroupNorm parameters are two registers.  Wide
        //      loads: a dword buffer load costs the memory pipeline as much as a 16-byte one (~100 cycles per wave
        //      instruction beside three other loading waves), and per-thread 4x4 patches fetched as 16 dwords made the
        //      producers, not the matrix pipe, the bound of the first version (stamps: 2.9k cycles of issue per 2.3k-cycle
        //      chunk);
        //   W1 (in the window)  GroupNorm(+SiLU) of those 12 values -> the wave's private LDS scratch X [ch][row][36];
        //      zero padding re-imposed on border tiles.  Each window element is activated ONCE (the 4x4 patches of
        //      neighbouring tiles overlap 2.5x);
        //   R  (beside the MFMAs)  the lane's own 4x4 patch (tile, channel) back from X: 8 ds_read_b64;
        //   W2 (in the next window)  B^T d B, 16 LDS stores into the stage.
        // Lane map for R / W2: 16 tiles x 2 k-steps x 2 channel parities, so that the 32 lanes of an LDS store group write
        // a 64-float span of the [tile][k-step] image at most 2-way conflicted (free).
        const int tid = threadIdx.x - 256, lane = tid & 63;
        const int pw = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int tyw = pw & 1, cg = pw >> 1;
        const int q = lane & 3;                                          // G / W1: the lane's channel inside the group
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)a.w, 0, nchunks * a.co_tiles * U_CHUNK_FLOATS * 4, 0x00020000);
        float *const xwP[2] = {lds + 2 * STAGE + XCH_FLOATS + STAT_FLOATS + pw * XWAVE,
                               lds + 2 * STAGE + XCH_FLOATS + STAT_FLOATS + (4 + pw) * XWAVE};      // scratch of even / odd chunks
        // slot u = (lane >> 2) + 16 j  ->  (row r, 4-float part) of the lane's channel; 36 of the 48 slots exist
        int lconst[3], xoff[3];
        unsigned slot_rp[3];                                             // r | part << 4 | valid << 8 (tile-independent)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int u = (lane >> 2) + 16 * j, r = u / 9, part = u - r * 9;
            const bool v = u < 36;
            // (the last part holds window columns 32, 33 and two floats past the window: the next pixels of the row, the
            //  next row, or -- at the very end of the tensor -- dwords past num_records, which a raw buffer load range-checks
            //  one by one and returns as 0 (tools/ubench/oob_probe.hip); they land in the scratch rows' padding)
            lconst[j] = v ? (r * a.Ws + 4 * part) * 4 + q * plane_bytes : OOB;
            // (slots that do not exist: the lane's dump slot; in the PLANAR instantiation every scratch row is de-interleaved,
            //  [even window columns: 20][odd: 20], also for the NCHW half of a concat: see below)
            //  (its dump stores go to +0, +1, +20, +21: two floats per lane keep them inside the dump area)
            xoff[j] = v ? (q * 4 + r) * XP + (PLANAR ? 2 : 4) * part : 16 * XP + lane * (PLANAR ? 2 : 4);
            slot_rp[j] = (unsigned)r | (unsigned)part << 4 | (v ? 256u : 0u);
        }
        // A parity-planar x1 (the output of an up2 convolution: [ch][row & 1][col & 1][H/2][W/2]): a window row lies in two
        // planes -- its even window columns (odd image columns: the window starts one pixel left of an even tile origin) and
        // its odd ones, 17 floats each.  10 lanes per row (5 x 16 bytes per plane) instead of 9, 40 of the 48 slots.  The
        // four floats of a load are every other window column; the scratch row keeps them de-interleaved -- [even columns:
        // 20][odd columns: 20] -- so that the store stays one ds_write_b128 (as four stride-2 dword stores they were 8-way
        // bank conflicts, the window LDS-bound), and the consumers gather their patches with ds_read2_b32.
        int lconstp[PLANAR ? 3 : 1], xoffp[PLANAR ? 3 : 1];
        unsigned slot_p[PLANAR ? 3 : 1];                                // r | first window column << 4 | valid << 12
        const int h2 = a.Hs >> 1, w2 = a.Ws >> 1;
        if (PLANAR) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int u = (lane >> 2) + 16 * j, r = u / 10, part = u - r * 10;
                const bool v = u < 40;
                const int px = part < 5 ? 1 : 0, py = (r + 1) & 1, yo = r == 0 ? -1 : (r == 3 ? 1 : 0);
                const int xo = px ? 4 * part - 1 : 4 * (part - 5), c0 = px ? 8 * part : 8 * (part - 5) + 1;
                lconstp[PLANAR ? j : 0] = v ? (((py * 2 + px) * h2 + yo) * w2 + xo) * 4 + q * plane_bytes : OOB;
                xoffp[PLANAR ? j : 0] = v ? (q * 4 + r) * XP + (px ? 4 * part : 20 + 4 * (part - 5)) : 16 * XP + lane * 4;      // (else: dump slot)
                slot_p[PLANAR ? j : 0] = (unsigned)r | (unsigned)c0 << 4 | (v ? 4096u : 0u);
            }
        }
        // per-lane LDS bases of both stages, so that every store below is base + a 16-bit immediate (no per-store VALU)
        float *const wdstP[2] = {lds + V_FLOATS + tid * 4, lds + STAGE + V_FLOATS + tid * 4};              // + e * 1024
        // tile descriptors: issue side (G) and activation side (W1, one or two chunks behind)
        int g_base = 0, g_n = 0, g_co = 0;                              // byte offset of the wave's window origin, sample, cout tile
        const float *g_src1 = a.x1, *g_src2 = a.x2 ? a.x2 : a.x1;      // the sample's planes in the two sources
        int w_co = 0;                                                    // cout tile of the weights being issued (one step behind G)
        bool g_bord = false;
        int vo[3] = {lconst[0], lconst[1], lconst[2]};                   // per-lane load offsets of the tile being loaded
        int g_so = 0;                                                    // ... and the scalar part of its window origin
        int vop[PLANAR ? 3 : 1] = {}, g_sop = 0;                        // the same for chunks of a parity-planar x1
        // The offsets the window loads actually use: long-lived registers, switched where a tile's chunks change source
        // (describe(): planar x1; issue_raw(): the first chunk of the NCHW skip half).  Selected per step they were
        // v_cndmask results placed in the loads' own destination registers, and the waitcnt pass -- which has to assume the
        // prologue's loads into them still in flight at the loop head -- made every step wait for the PREVIOUS step's loads
        // before issuing its own: the two-chunk prefetch was gone, the planar kernel producer-bound and 26 % behind.
        int va[3] = {lconst[0], lconst[1], lconst[2]}, g_sa = 0;
        unsigned g_vmp = 0xfffu, a_vmp = 0xfffu;
        unsigned g_vm = 0xfffu, g_lsh = 0;                              // validity of the 12 loaded elements; left-edge shift
        unsigned a_vm = 0xfffu, a_lsh = 0;
        bool a_bord = false;
        auto describe = [&](int k) __attribute__((always_inline)) {
            const TileId tl = decode_tile(a, tile_of(k));
            const int iy0 = tl.oy0 - 1 + 2 * tyw, ix0 = tl.ox0 - 1;
            g_n = tl.n; g_co = tl.co0 / BN;
            // the consumers take the tile's origin from here: decoding it themselves was ~50 scalar instructions per tile
            // IN the MFMA waves (every instruction of those is additive); all lanes store the same 16 bytes
            if (pw == 0) *reinterpret_cast<i32x4 *>(tdesc + (k & 1) * 4) = i32x4{tl.n, tl.co0, tl.oy0, tl.ox0};
            g_src1 = a.x1 + (size_t)tl.n * a.C1 * (plane_bytes / 4);
            g_src2 = a.x2 ? a.x2 + (size_t)tl.n * a.C2 * (plane_bytes / 4) : g_src1;
            g_bord = tl.oy0 - 1 < 0 || tl.ox0 - 1 < 0 || tl.oy0 + TH + 1 > a.H || tl.ox0 + TW + 1 > a.W;
            g_base = (iy0 * a.Ws + ix0) * 4;
            g_so = g_bord ? 0 : g_base;
#pragma unroll
            for (int j = 0; j < 3; ++j) vo[j] = lconst[j];
            if (PLANAR) {
                const int basep = (((tl.oy0 >> 1) + tyw) * w2 + (tl.ox0 >> 1)) * 4;
                g_sop = g_bord ? 0 : basep;
                g_vmp = 0xfffu;
#pragma unroll
                for (int j = 0; j < 3; ++j) vop[PLANAR ? j : 0] = lconstp[PLANAR ? j : 0];
                if (g_bord) {
                    g_vmp = 0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const unsigned sp = slot_p[PLANAR ? j : 0];
                        const int r = sp & 15, c0 = sp >> 4 & 255;
                        const bool rowok = (sp & 4096u) && iy0 + r >= 0 && iy0 + r < a.H;
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
                for (int j = 0; j < 3; ++j) {
                    const int r = slot_rp[j] & 15, part = slot_rp[j] >> 4 & 15;
                    const bool rowok = (slot_rp[j] & 256u) && iy0 + r >= 0 && iy0 + r < a.H;
                    // the 16 bytes of the leftmost part of an image row start one pixel before the row: shifted by one
                    // pixel and rotated back after the load (at the very first row they would start before the buffer)
                    const bool lsh = rowok && ix0 + 4 * part < 0;
                    g_lsh |= lsh ? 1u << j : 0u;
                    vo[j] = rowok ? lconst[j] + g_base + (lsh ? 4 : 0) : OOB;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ix = ix0 + 4 * part + e;
                        g_vm |= (rowok && ix >= 0 && ix < a.W && 4 * part + e < 34) ? 1u << (4 * j + e) : 0u;
                    }
                    // (a load of a border tile may straddle the end of an image row: the next row's pixels, or -- at the last
                    //  row of the tensor -- dwords past num_records, which a raw buffer load range-checks one by one and
                    //  returns as 0: tools/ubench/oob_probe.hip; either way those elements are masked by g_vm)
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) va[j] = PLANAR ? vop[PLANAR ? j : 0] : vo[j];      // chunk 0 of a tile is x1's
            g_sa = PLANAR ? g_sop : g_so;
        };
        struct Raw { f32x4 v[3]; float sc, sh; bool planar; };
        Raw rawA, rawB;
        f32x4 wA[8], wB[8];
        // (no integer division, 64-bit address arithmetic or per-step address VALU outside the window: beside the partner's
        //  MFMA stream every VALU instruction of this wave waits for a stall gap -- the first pipelined version spent 2.6k
        //  cycles per 2.3k-cycle chunk in its "issue" phase on three uniform s / nchunks divisions and two 64-bit addresses)
        const __amdgpu_buffer_rsrc_t gsc_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.act ? a.gn_scale : a.out), 0, a.act ? (a.B * Ctot + 64) * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t gsh_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.act ? a.gn_shift : a.out), 0, a.act ? (a.B * Ctot + 64) * 4 : 0, 0x00020000);
        // Every step issues the SAME 13 loads, needed or not: past the end of the stream they re-read chunks of the last
        // tile (valid addresses, results unused; without a GroupNorm prologue the two parameter loads go through a
        // descriptor with zero records and are dropped by the range check, still counted by vmcnt).  With conditional issue the compiler's
        // waitcnt pass has to assume the shortest path and made the LDS stores of w(s) wait for vmcnt(0) -- i.e. for the
        // loads this very step had just issued: no prefetch at all (2.8k cycles of "issue" per 2.3k-cycle chunk).
        auto issue_w = [&](int ch, f32x4 (&w)[8]) __attribute__((always_inline)) {
            if (ch == 0) w_co = g_co;                                    // the weights enter the tile described last
            const int w_soff = (ch * a.co_tiles + w_co) * (U_CHUNK_FLOATS * 4);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                w[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (tid + e * 256) * 16, w_soff, 0));
        };
        auto issue_raw = [&](int ch, Raw &r) __attribute__((always_inline)) {
            const int c0 = ch * KC;
            const bool from1 = c0 < a.C1;
            const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(from1 ? g_src1 : g_src2), 0, (from1 ? a.C1 : a.C2) * plane_bytes, 0x00020000);
            // interior tiles: tile origin through the scalar offset (g_so), per-lane offsets fixed for the kernel; border
            // tiles: range-checked per-lane offsets; describe() put whichever applies into vo[]
            const int cb = ((from1 ? c0 : c0 - a.C1) + 4 * cg) * plane_bytes;
            r.planar = PLANAR && from1;
            if (PLANAR && c0 == a.C1) {  // (uniform, once per tile) only x1 is stored parity-planar; the skip half of a concat is NCHW
#pragma unroll
                for (int j = 0; j < 3; ++j) va[j] = vo[j];
                g_sa = g_so;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
                r.v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, va[j], cb + g_sa, 0));
            const int gso = (g_n * Ctot + c0 + 4 * cg) * 4;
            r.sc = bload(gsc_rsrc, q * 4, gso);
            r.sh = bload(gsh_rsrc, q * 4, gso);
        };
        // W1: activate the 12 landed values, zero what lies outside the image, park them in the wave's scratch
        auto activate = [&](Raw &r, float *xw) __attribute__((always_inline)) {
            // pairs: the four non-transcendental operations of an element are packed-f32 instructions (the window's cost
            // is its instruction count: ~10 cycles per instruction of a lone wave, stamps)
            f32x2 d[6];
#pragma unroll
            for (int j = 0; j < 3; ++j) { d[2 * j] = f32x2{r.v[j][0], r.v[j][1]}; d[2 * j + 1] = f32x2{r.v[j][2], r.v[j][3]}; }
            const bool pl = PLANAR && r.planar;
            if (a_bord && !pl) {  // (uniform) undo the left-edge shift: {x0, x1, x2, x3} loaded from one pixel further right
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (a_lsh >> j & 1) { d[2 * j + 1] = f32x2{d[2 * j][1], d[2 * j + 1][0]}; d[2 * j] = f32x2{0.0f, d[2 * j][0]}; }
            }
            if (a.act) {
                const f32x2 sc2 = {r.sc, r.sc}, sh2 = {r.sh, r.sh};
#pragma unroll
                for (int e = 0; e < 6; ++e) d[e] = __builtin_elementwise_fma(d[e], sc2, sh2);
                if (a.act == 2) {
                    f32x2 ex[6];
#pragma unroll
                    for (int e = 0; e < 6; ++e) ex[e] = d[e] * -1.4426950408889634f;
#pragma unroll
                    for (int e = 0; e < 6; ++e) { ex[e][0] = __builtin_amdgcn_exp2f(ex[e][0]); ex[e][1] = __builtin_amdgcn_exp2f(ex[e][1]); }
#pragma unroll
                    for (int e = 0; e < 6; ++e) ex[e] = ex[e] + 1.0f;
#pragma unroll
                    for (int e = 0; e < 6; ++e) { ex[e][0] = __builtin_amdgcn_rcpf(ex[e][0]); ex[e][1] = __builtin_amdgcn_rcpf(ex[e][1]); }
#pragma unroll
                    for (int e = 0; e < 6; ++e) d[e] = d[e] * ex[e];
                }
            }
            if (a_bord) {
                const unsigned vm = pl ? a_vmp : a_vm;
#pragma unroll
                for (int e = 0; e < 12; ++e) d[e >> 1][e & 1] = (vm >> e & 1) ? d[e >> 1][e & 1] : 0.0f;
            }
            if (pl) {            // every other window column: the de-interleaved half of the scratch row
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    *reinterpret_cast<f32x4 *>(xw + xoffp[PLANAR ? j : 0]) = f32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]};
            } else if (PLANAR) { // four consecutive columns into the de-interleaved row: one scratch format per instantiation, so
                                 // that the consumers' patch reads are branch-free (two formats cost them 12 v_mov and a
                                 // full lgkmcnt wait in front of the first MFMA of every chunk)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float *dst = xw + xoff[j];
                    dst[0] = d[2 * j][0]; dst[1] = d[2 * j + 1][0];
                    dst[20] = d[2 * j][1]; dst[21] = d[2 * j + 1][1];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    *reinterpret_cast<f32x4 *>(xw + xoff[j]) = f32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]};
            }
        };
        const bool pstamp = IPDM_CONV_STAMPS && (a.dbg & 8) != 0 && !(a.dbg & 32);      // (32: consumer stamps only -- the producers' perturb more)
        unsigned long long p_issue = 0, p_wait = 0, p_math = 0, p_hand = 0, p_t = 0;
        // one step: on entry  w(s) is in `wc` (issued one step ago), raw(s+1) in `rc` (issued one step ago), patch(s) in
        // registers; the step issues w(s+1) -> `wn` and raw(s+2) -> `rn`
        int k = 0, ch = 0;                                         // tile / chunk-in-tile of the step's chunk s (running counters)
        auto step = [&](int s, auto par, f32x4 (&wc)[8], f32x4 (&wn)[8], Raw &rc, Raw &rn) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par)::value;                // s & 1: the stage addresses are compile-time constants
            if (pstamp) p_t = __builtin_amdgcn_s_memtime();
            const bool more1 = s + 1 < S;
            const int ch1 = ch + 1 == nchunks ? 0 : ch + 1, ch2 = ch1 + 1 == nchunks ? 0 : ch1 + 1;
            issue_w(ch1, wn);
            issue_raw(ch2, rn);
            // w(s), raw(s+1) have landed; this step's own 13 loads stay in flight across the window
            asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
            // stage (s&1) was last read by chunk s-2, which the consumers finished before the previous hand-over
#pragma unroll
            for (int e = 0; e < 8; ++e) *reinterpret_cast<f32x4 *>(wdstP[PAR] + e * 1024) = wc[e];
            if (pstamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); p_issue += now - p_t; p_t = now; }
            // W1(s+1) beside the consumers' MFMAs of chunk s-1: no barrier in front of it.  The scratch is double-buffered by
            // chunk parity (the consumers fetched X(s) from the other half right after the previous hand-over), the stage
            // half written above was last read two chunks ago.  A producer's VALU only gets the stall gaps of the MFMA wave it
            // shares a SIMD with -- the Winograd stream has one every four MFMAs -- and costs that stream nothing; the window in
            // which the consumers used to wait for this burst (barrier A) is gone.
            const bool after_tile = ch == 0 && s > 0;
            if (more1) {
                if (ch == nchunks - 1) { a_vm = g_vm; a_vmp = g_vmp; a_lsh = g_lsh; a_bord = g_bord; }      // chunk s+1 opens the tile described last
                activate(rc, xwP[PAR ^ 1]);
            }
            // the next tile's descriptors (its first raw loads are issued two steps from now)
            if (ch == nchunks - 3 && k + 1 < n_my) describe(k + 1);
            if (pstamp) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                          const unsigned long long now = __builtin_amdgcn_s_memtime(); p_math += now - p_t; p_t = now; }
            if (after_tile) __syncthreads();               // E: the consumers' exchange barrier of the tile that just ended
            __syncthreads();                               // hand-over: stage (s&1) and X(s+1) are complete
            if (pstamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); p_wait += now - p_t; p_t = now; }
            ch = ch1;
            k += ch1 == 0 ? 1 : 0;
            if (pstamp) p_hand += __builtin_amdgcn_s_memtime() - p_t;
        };
        if (S > 0) {
            // prologue (nobody multiplies yet: the SIMDs are free): tile 0, G(0), G(1), W1(0), R(0)
            describe(0);
            a_vm = g_vm; a_vmp = g_vmp; a_lsh = g_lsh; a_bord = g_bord;
            issue_w(0, wA);
            issue_raw(0, rawB);
            issue_raw(1, rawA);                                // (nchunks >= 4: chunk 1 belongs to tile 0)
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            activate(rawB, xwP[0]);
            __syncthreads();                                   // P: X(0) is in the scratch; the consumers fetch their patches
            for (int s = 0; s < S; s += 2) {
                step(s, std::integral_constant<int, 0>{}, wA, wB, rawA, rawB);
                if (s + 1 < S) step(s + 1, std::integral_constant<int, 1>{}, wB, wA, rawB, rawA);
            }
            __syncthreads();                               // E of the last tile
        }
        if (pstamp && !(a.dbg & 16) && tid == 0) {
            unsigned long long *dd = a.dbg_buf + (size_t)blockIdx.x * 8 + 4;
            dd[0] = p_issue; dd[1] = p_wait; dd[2] = p_math; dd[3] = p_hand;
        }
        return;
    }

    // =============================================================================== CONSUMERS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 5, l31 = lane & 31;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int ih = swave & 1, h = swave >> 1;              // which two rows of the transform domain, which cout half
    // the lane's tile inside the workgroup tile: row ty, column 2 txh + odd.  The two tiles of a column pair sit 16 lanes
    // apart (DPP rows r, r + 1), so that the epilogue's exchange of halves is one v_permlane16_swap per register pair
    const int odd = l31 >> 4, ty = l31 & 1, txh = (l31 & 15) >> 1;
    f32x16 acc[8];
    const int out_plane = a.Ho * a.Wo;
    const int plane4 = out_plane * 4;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
    float nb = 0.0f;
    auto fetch_bias = [&](int co0) __attribute__((always_inline)) {
        nb = bload(b_rsrc, lk ? OOB : l31 * 4, (co0 + h * 32) * 4);
    };
    if (S > 0 && ih == 0) fetch_bias(decode_tile(a, tile_of(0)).co0);
    const float sgn = ih == 0 ? 1.0f : -1.0f;
    // R / W2 (the consumers' share of the window): wave w transforms the (tile, channel) pairs of producer wave w's
    // scratch -- lane map 16 tiles x 2 k-steps x 2 channel parities, so that the 32 lanes of an LDS store group write a
    // 64-float span of the [tile][k-step] image at most 2-way conflicted (free)
    const int w_t16 = lane & 15, w_kpl = (lane >> 4) & 1;
    const float *const xr0 = lds + 2 * STAGE + XCH_FLOATS + STAT_FLOATS + swave * XWAVE + ((2 * w_kpl + lk) * 4) * XP + 2 * w_t16;
    const int v_slot = (w_t16 & 1) * 16 + (w_t16 >> 1) * 2 + (swave & 1);           // MFMA lane of tile (row swave & 1, column w_t16)
    const int v_lane = (lk * 32 + v_slot) * 4 + 2 * (swave >> 1) + w_kpl;                        // + xi * 256 (+ stage)
    float patch[16];
    // (PLANAR instantiation: every scratch row is [even window columns: 20][odd: 20]; the patch registers keep the order the
    //  two ds_read2_b32 of a row deliver -- columns 0, 2, 1, 3 -- and W2 indexes them through pcol)
    constexpr int pcol[4] = {0, PLANAR ? 2 : 1, PLANAR ? 1 : 2, 3};                  // register position of patch column c
    auto read_patch = [&](int par) __attribute__((always_inline)) {
        const float *const xr = xr0 + par * (4 * XWAVE);                            // the scratch half of the chunk's parity
        const float *const xrp = xr - w_t16;                                        // column pairs (t16, t16 + 1) of both halves
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
                patch[4 * r] = lo[0]; patch[4 * r + 1] = lo[1]; patch[4 * r + 2] = hi[0]; patch[4 * r + 3] = hi[1];
            }
        }
    };
    const int a_off = V_FLOATS + ((((8 * ih) * 2 + h) * 2 + lk) * 32 + l31) * 4;      // + e * 512 floats
    const int b_off = (((8 * ih) * 2 + lk) * 32 + l31) * 4;                           // + e * 256 floats
    const bool stamp = IPDM_CONV_STAMPS && (a.dbg & 8) != 0;
    const bool estamp = IPDM_CONV_STAMPS && (a.dbg & 16) != 0;      // epilogue split into the producers' four slots
    unsigned long long e_p1 = 0, e_wait = 0, e_p2 = 0, e_p3 = 0, e_t = 0;
    unsigned long long t_mma = 0, t_epi = 0, t_bar = 0;
    const unsigned long long t_begin = stamp ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long t_last = t_begin;
    if (S > 0) {
        __syncthreads();                                   // P: the producers' prologue has put X(0) into the scratch
        read_patch(0);
    }
    // One chunk: W2(s), hand-over, R(s + 1), the 32 MFMAs.  The first chunk of a tile STARTS its accumulators (C = 0 in the
    // first MFMA of each) instead of clearing them beforehand, and the tile loop below peels it: cleared under `if (ch == 0)`
    // inside one flat chunk loop the 128 accumulator registers were a loop-carried value the compiler could not see dying,
    // i.e. live through the whole epilogue -- no room to have the tile's residual loads in flight together (and 128 v_mov
    // per tile in the MFMA waves).
    int s = 0;                                             // running chunk index (stage / scratch parity): no division in the MFMA wave
    auto chunk = [&](auto first, int ch) __attribute__((always_inline)) {
        {
            // W2(s): B^T d B of the lane's 4x4 patch (read from the producers' scratch one chunk ago, beside the MFMAs)
            float tt[16];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                tt[0 + x] = patch[0 + x] - patch[8 + x];
                tt[4 + x] = patch[4 + x] + patch[8 + x];
                tt[8 + x] = patch[8 + x] - patch[4 + x];
                tt[12 + x] = patch[4 + x] - patch[12 + x];
            }
            float *vdst = lds + (s & 1) * STAGE + v_lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rs = row_slot(i);
                vdst[(rs * 4 + 0) * 256] = tt[i * 4 + pcol[0]] - tt[i * 4 + pcol[2]];
                vdst[(rs * 4 + 1) * 256] = tt[i * 4 + pcol[1]] + tt[i * 4 + pcol[2]];
                vdst[(rs * 4 + 2) * 256] = tt[i * 4 + pcol[2]] - tt[i * 4 + pcol[1]];
                vdst[(rs * 4 + 3) * 256] = tt[i * 4 + pcol[1]] - tt[i * 4 + pcol[3]];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                   // hand-over: stage (s&1) is complete
        __builtin_amdgcn_sched_barrier(0);
        const float *stage = lds + (s & 1) * STAGE;
        {
            constexpr bool FIRST = decltype(first)::value;
            // the operands of the first two positions first: the LDS returns in order, and behind the patch reads of R(s+1) the
            // first MFMA of every chunk started ~200 cycles later
            f32x4 a_c = *reinterpret_cast<const f32x4 *>(stage + a_off), b_c = *reinterpret_cast<const f32x4 *>(stage + b_off), a_n, b_n;
            if (stamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_bar += now - t_last; t_last = now; }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e + 1 < 8) {
                    a_n = *reinterpret_cast<const f32x4 *>(stage + a_off + (e + 1) * 512);
                    b_n = *reinterpret_cast<const f32x4 *>(stage + b_off + (e + 1) * 256);
                }
                if (e == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    // R(s+1), used after the MFMAs (unconditional: past the last chunk it reads stale scratch, unused -- under
                    // `if (s + 1 < S)` the waitcnt in front of the first MFMA has to assume the path without these reads)
                    read_patch((s + 1) & 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (FIRST) {
                    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[0], b_c[0], zero, 0, 0, 0);
                } else {
                    acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[0], b_c[0], acc[e], 0, 0, 0);
                }
#pragma unroll
                for (int q = 1; q < 4; ++q) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[q], b_c[q], acc[e], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (e + 1 < 8) { a_c = a_n; b_c = b_n; }
            }
        }
        if (stamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_mma += now - t_last; t_last = now; }
        ++s;
    };
    for (int k = 0; k < n_my; ++k) {
        chunk(std::true_type{}, 0);
        for (int ch = 1; ch < nchunks; ++ch) chunk(std::false_type{}, ch);
        // ---------------------------------------------------------------- tile epilogue
        if (estamp) e_t = __builtin_amdgcn_s_memtime();
        // the tile's origin, published by the producers when they described it (two chunks before its first load)
        const i32x4 td = *reinterpret_cast<const i32x4 *>(tdesc + (k & 1) * 4);
        // + bias through position (1, 1) (accumulator 5 of the ih = 0 waves), whose output coefficients are all 1
        if (ih == 0) {
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(nb, 1.0f, acc[5], 0, 0, 0);
            if (k + 1 < n_my) fetch_bias(__builtin_amdgcn_readfirstlane(tdesc[((k + 1) & 1) * 4 + 1]));
        }
        TileId t;
        t.n = __builtin_amdgcn_readfirstlane(td[0]); t.co0 = __builtin_amdgcn_readfirstlane(td[1]);
        t.oy0 = __builtin_amdgcn_readfirstlane(td[2]); t.ox0 = __builtin_amdgcn_readfirstlane(td[3]);
        const size_t sample = (size_t)t.n * a.Cout * out_plane;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + sample), 0, a.Cout * out_plane * 4, 0x00020000);
        // (no residual: zero records -- the loads below return 0 and the add stays unconditional; as a select on the uniform
        //  a.res it was four v_cndmask per pair)
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((a.res ? a.res : a.out) + sample), 0,
                                                                                   a.res ? a.Cout * out_plane * 4 : 0, 0x00020000);
        // The lanes l and l + 16 (tile columns 2m, 2m + 1) hold 4 consecutive pixels of the row between them.  Registers are
        // taken in pairs (couts c, c + 1): v_permlane16_swap exchanges one half each, after which the lane of the even column
        // owns the 4 pixels of cout c and the other one those of cout c + 1 -- 16-byte stores and residual loads, half as
        // many memory instructions (the store path retires about one instruction per 100 cycles and wave whatever its width:
        // with 8-byte stores this phase took 5.3k cycles per tile without residual, 11.7k with residual and statistics), and
        // the statistics need 3 DPP steps per pair instead of 4 per register.
        const int py = t.oy0 + 2 * ty + ih, px4 = t.ox0 + 4 * txh;
        const bool rok = py < a.Ho;
        const int lane_off4 = ((lk * 4 + odd) * out_plane + (2 * ty + ih) * a.Wo + 4 * txh) * 4;
        const int voff4 = (rok && px4 + 3 < a.Wo) ? lane_off4 : OOB;             // all four pixels of the lane's run
        const bool ragged = t.ox0 + TW > a.Wo && (a.Wo & 3) != 0;                // (wave-uniform) a run straddles the right edge
        const bool clipped = t.oy0 + TH > a.Ho || t.ox0 + TW > a.Wo;             // (wave-uniform) some lanes own no pixels
        const int nval = a.Wo - px4;                                              // ... then it has 1..3 pixels
        const bool part = ragged && rok && nval > 0 && nval < 4;
        const int so0 = ((t.co0 + h * 32) * out_plane + min(t.oy0, a.Ho - 1) * a.Wo + t.ox0) * 4;
        const float *xr2 = xch + ((swave ^ 1) * 8) * 256 + lane * 4;
        float *sb = stat_lds + swave * 128;
        const f32x2 sgn2 = {sgn, sgn};
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        // ALL residual loads of the tile are issued together, ahead of the transform: issued one pair ahead inside the store
        // loop, every pair waited a full memory latency (and, vmcnt counting in order, for the stores in between) -- 6.0k of
        // the epilogue's 7.5k cycles per tile (stamps)
        f32x4 rv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            rv[i] = RES ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, voff4, so0 + (8 * (i >> 1) + 2 * (i & 1)) * plane4, 0))
                        : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        __builtin_amdgcn_sched_barrier(0);
        // columns first (in-lane): T_i[b] = sum_j M[i][j] A[j][b],  A^T = [[1,1,1,0],[0,1,-1,-1]]; then the wave's own two
        // rows (first = accumulators 0-3, second = 4-7; ih = 0: rows 0, 1, ih = 1: rows 3, 2 -- row_slot): both waves keep
        // K = first + second and send the second (T1 resp. T2); output row 0 = (T0 + T1) + T2, row 1 = T1 - (T2 + T3).
        // Register pairs (r, r + 1) = couts (c, c + 1) as packed-f32 operations: half the instructions.
        f32x2 K0[8], K1[8];                                 // [pair]: output column 0 / 1 of {cout c, cout c + 1}
        {
            float *xw = xch + (swave * 8) * 256 + lane * 4;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int r = 2 * p;
#define IPDM_M(e) f32x2{acc[e][r], acc[e][r + 1]}
                const f32x2 lo0 = (IPDM_M(0) + IPDM_M(1)) + IPDM_M(2), lo1 = (IPDM_M(1) - IPDM_M(2)) - IPDM_M(3);
                const f32x2 hi0 = (IPDM_M(4) + IPDM_M(5)) + IPDM_M(6), hi1 = (IPDM_M(5) - IPDM_M(6)) - IPDM_M(7);
#undef IPDM_M
                K0[p] = lo0 + hi0;
                K1[p] = lo1 + hi1;
                *reinterpret_cast<f32x4 *>(xw + p * 256) = f32x4{hi0[0], hi0[1], hi1[0], hi1[1]};
            }
        }
        if (estamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); e_p1 += now - e_t; e_t = now; }
        __syncthreads();                                   // E: both halves of every (tile, cout) are in LDS
        if (estamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); e_wait += now - e_t; e_t = now; }
        f32x4 gotv[8];                                     // partner's {T[0] c, T[0] c+1, T[1] c, T[1] c+1} of every pair: one LDS latency
#pragma unroll
        for (int i = 0; i < 8; ++i) gotv[i] = *reinterpret_cast<const f32x4 *>(xr2 + i * 256);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int g = i >> 1, u = 2 * (i & 1);
            const f32x4 got = gotv[i];
            // ih = 0: K + T2 (output row 0);  ih = 1: T1 - K (output row 1)
            const f32x2 y0 = __builtin_elementwise_fma(K0[i], sgn2, f32x2{got[0], got[1]});       // column 0 of {cout c, c + 1}
            const f32x2 y1 = __builtin_elementwise_fma(K1[i], sgn2, f32x2{got[2], got[3]});       // column 1
            // rows r (even tile column) and r + 1 (odd) of the DPP row pair: the even one gives its cout c + 1 and takes the
            // odd one's cout c
            // (inline asm: through the builtin, with the halves of the 2-vectors as operands, this compiler passed component 0
            //  of y0 / y1 as BOTH operands of the swap; s_nop: the swap reads need two wait states after the VALU writes)
            float ya0 = y0[0], yb0 = y0[1], ya1 = y1[0], yb1 = y1[1];
            asm("s_nop 1\n\t"
                "v_permlane16_swap_b32 %0, %1\n\t"
                "v_permlane16_swap_b32 %2, %3"
                : "+v"(ya0), "+v"(yb0), "+v"(ya1), "+v"(yb1));
            f32x4 v = {ya0, ya1, yb0, yb1};
            const int so = so0 + (8 * g + u) * plane4;
            v += rv[i];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, voff4, so, 0);
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
                // row_ror:2, 4, 8 with the DPP operand on the add itself (as update_dpp + add the compiler emits a zeroing
                // move, a DPP move and an add per step); s_nop: a DPP read needs two wait states after the VALU write
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
        if (estamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); e_p2 += now - e_t; e_t = now; }
        if (a.stats) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int row_y = t.oy0 + 2 * lk + ih;          // lanes 0-31: tile row 0, lanes 32-63: tile row 1; cout = l31
            if (row_y < a.Ho) {
                float *dst = a.stats + (((size_t)t.n * a.stats_rows + (size_t)row_y * a.tiles_x + t.ox0 / TW) * a.Cout + t.co0 + h * 32 + l31) * 2;
                *reinterpret_cast<f32x2 *>(dst) = *reinterpret_cast<const f32x2 *>(sb + lane * 2);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (estamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long now = __builtin_amdgcn_s_memtime(); e_p3 += now - e_t; }
        if (stamp) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_epi += now - t_last; t_last = now; }
    }
    if (estamp && tid == 0) {
        unsigned long long *d = a.dbg_buf + (size_t)blockIdx.x * 8 + 4;
        d[0] = e_p1; d[1] = e_wait; d[2] = e_p2; d[3] = e_p3;
    }
    if (stamp && tid == 0) {
        unsigned long long *d = a.dbg_buf + (size_t)blockIdx.x * 8;
        d[0] = t_mma; d[1] = t_epi; d[2] = t_bar; d[3] = __builtin_amdgcn_s_memtime() - t_begin;
    }
}

}  // namespace

namespace ipdm {

// Which convolutions run in the Winograd domain: wide 3x3 stride-1 layers with packed U weights, whole 64-cout tiles and
// whole 8-channel chunks, no resampling on the way in.  Layers the direct tiling splits along K (conv_ws_split(a) > 1: too few
// tiles for any tiling) run here too when the 128-cout kernel can split them itself (conv_wino_split: K slices INSIDE
// conv_wino2, round 4), otherwise they stay on the K-split direct kernel.  The rule looks at the layer only, never at the batch.
bool conv_wino_eligible(const ConvArgs &a)
{
    if (opt(OPT_CONV_NO_WINO) || !a.w_wino) return false;
    if (a.ksize != 3 || a.stride != 1 || a.upsample || a.H != a.Ho || a.W != a.Wo) return false;
    if (a.w_interleave != 2 && a.w_interleave != 4) return false;
    const int Ctot = a.C1 + a.C2;
    if (a.Cout % BN || Ctot % KC || Ctot < 32 || (a.C2 && a.C1 % KC)) return false;
    if ((a.x1_planar && ((a.Hs | a.Ws) & 1)) || conv_up2_eligible(a)) return false;
    if (conv_ws_split(a) == 1) return true;
    return !opt(OPT_WINO_V1) && conv_wino_split(a) > 1;
}

bool conv_wino_shape_ok(int Cout, int Cin, int ks, int stride, int interleave)
{
    return ks == 3 && stride == 1 && (interleave == 2 || interleave == 4) && Cout % BN == 0 && Cin % KC == 0 && Cin >= 32;
}

// [chunk q][cout tile][xi][h][lk][cout 32][kp], xi = row_slot(i) * 4 + j: U = G g G^T in double, rounded once; channel = 8 q + 2 kp + lk
// Layers conv_wino3.hip can take (whole 128-cout tiles, whole 16-channel chunks) get a second image behind the first: the same float32
// values split into three bf16 terms by truncation (u = u1 + u2 + u3 exactly),
// [16-channel chunk Q][128-cout tile][xi][term 3][cout quarter 4][k half 2][cout 32][8 x bf16], channel = 16 Q + 8 (k half) + element
void conv_pack_weights_wino(const float *w, int Cout, int Cin, std::vector<float> &packed)
{
    const int nq = (Cin + KC - 1) / KC, nct = Cout / BN;
    const bool with3 = Cout % 128 == 0 && Cin % 16 == 0;
    const size_t f32_floats = (size_t)nq * nct * U_CHUNK_FLOATS, u3_block_halves = (size_t)16 * 3 * 4 * 2 * 32 * 8;
    packed.assign(f32_floats + (with3 ? (size_t)(Cin / 16) * (Cout / 128) * u3_block_halves / 2 : 0), 0.0f);
    unsigned short *u3 = reinterpret_cast<unsigned short *>(packed.data() + f32_floats);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float *g = w + ((size_t)co * Cin + ci) * 9;
            double gg[4][3];
            for (int i = 0; i < 4; ++i)
                for (int x = 0; x < 3; ++x) gg[i][x] = G[i][0] * g[0 * 3 + x] + G[i][1] * g[1 * 3 + x] + G[i][2] * g[2 * 3 + x];
            const int q = ci / KC, kp = (ci % KC) >> 1, lk = ci & 1, ct = co / BN, hh = (co % BN) / 32, cl = co % 32;
            float *base = packed.data() + ((size_t)q * nct + ct) * U_CHUNK_FLOATS;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = gg[i][0] * G[j][0] + gg[i][1] * G[j][1] + gg[i][2] * G[j][2];
                    base[(((((row_slot(i) * 4 + j) * 2 + hh) * 2 + lk) * 32 + cl) * 4) + kp] = (float)u;
                    if (with3) {
                        const float uf = (float)u;
                        unsigned b1, b2, b3;
                        float r1, r2;
                        memcpy(&b1, &uf, 4); b1 &= 0xffff0000u;
                        float t1; memcpy(&t1, &b1, 4); r1 = uf - t1;
                        memcpy(&b2, &r1, 4); b2 &= 0xffff0000u;
                        float t2; memcpy(&t2, &b2, 4); r2 = r1 - t2;
                        memcpy(&b3, &r2, 4);
                        const int Q = ci / 16, kb = (ci % 16) / 8, el = ci % 8, TT = co / 128, hq = (co % 128) / 32;
                        unsigned short *blk = u3 + ((size_t)Q * (Cout / 128) + TT) * u3_block_halves;
                        const unsigned terms[3] = {b1, b2, b3};
                        for (int t = 0; t < 3; ++t)
                            blk[((((size_t)((row_slot(i) * 4 + j) * 3 + t) * 4 + hq) * 2 + kb) * 32 + cl) * 8 + el] = (unsigned short)(terms[t] >> 16);
                    }
                }
        }
}

int conv2d_wino_launch(const ConvArgs &args, hipStream_t st)
{
    ConvArgs a = args;
    a.w = args.w_wino;
    a.tiles_x = cdiv(a.Wo, TW);
    a.tiles_y = cdiv(a.Ho, TH);
    a.co_tiles = a.Cout / BN;
    a.ksplit = args.ksplit > 1 ? args.ksplit : 1;
    a.dbg = (a.dbg_buf ? (opt(OPT_CONV_DBG) & 56) : 0) | (opt(OPT_CONV_DBG) & 7);
    IPDM_REQUIRE(conv_wino_eligible(args), "conv2d_wino: layer not eligible");
    IPDM_REQUIRE((long)a.C1 * a.Hs * a.Ws < (1L << 29) && (long)(a.C2 + 1) * a.Hs * a.Ws < (1L << 29) &&
                     (long)a.Cout * a.Ho * a.Wo < (1L << 29) && (long)(a.C1 + a.C2) / KC * a.co_tiles * U_CHUNK_FLOATS < (1L << 29),
                 "conv2d_wino: per-sample tensor exceeds the 2 GiB buffer-addressing range");
    const long ntiles = (long)a.tiles_x * a.tiles_y * a.co_tiles * a.B;
    IPDM_REQUIRE(ntiles < (1L << 31), "conv2d_wino: too many tiles");
    IPDM_REQUIRE(!a.stats || a.stats_rows == a.tiles_x * a.Ho, "conv2d_wino: statistics rows %d != %d", a.stats_rows, a.tiles_x * a.Ho);
    const int cus = device_cu_count();
    int G = (int)(ntiles < cus ? ntiles : cus);
    G = (G + 7) / 8 * 8;
    const bool res = a.res != nullptr;
    const void *fn = a.x1_planar ? (res ? (const void *)conv_wino_kernel<true, true> : (const void *)conv_wino_kernel<true, false>)
                                 : (res ? (const void *)conv_wino_kernel<false, true> : (const void *)conv_wino_kernel<false, false>);
    if (int rc = ensure_dynamic_lds(fn, LDS_BYTES)) return rc;
    const bool v3 = conv_wino3_eligible(a);               // (opt-in, option conv_bf16x3: a rule of the layer alone)
    const bool prof = prof_enabled(), v2 = v3 || a.ksplit > 1 || (!opt(OPT_WINO_V1) && conv_wino2_eligible(a));
    IPDM_REQUIRE(a.ksplit == 1 || conv_wino2_eligible(a), "conv2d_wino: this layer cannot be split into %d K slices", a.ksplit);
    if (prof) prof_before(v2 ? 5 : 3, st);
    if (v2) {
        if (int rc = v3 ? conv2d_wino3_launch(a, st) : conv2d_wino2_launch(a, st)) return rc;
    } else if (a.x1_planar && res) hipLaunchKernelGGL((conv_wino_kernel<true, true>), dim3((unsigned)G), dim3(512), LDS_BYTES, st, a, (int)ntiles);
    else if (a.x1_planar) hipLaunchKernelGGL((conv_wino_kernel<true, false>), dim3((unsigned)G), dim3(512), LDS_BYTES, st, a, (int)ntiles);
    else if (res) hipLaunchKernelGGL((conv_wino_kernel<false, true>), dim3((unsigned)G), dim3(512), LDS_BYTES, st, a, (int)ntiles);
    else hipLaunchKernelGGL((conv_wino_kernel<false, false>), dim3((unsigned)G), dim3(512), LDS_BYTES, st, a, (int)ntiles);
    // EXECUTED flops: 16 multiply-adds per 2x2 output tile and (cin, cout) pair (the 3x3 form counts 36)
    if (prof) prof_after(v2 ? 5 : 3, 2.0 * a.B * (double)cdiv(a.Ho, 2) * cdiv(a.Wo, 2) * 16.0 * a.Cout * (a.C1 + a.C2), st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

}  // namespace ipdm
