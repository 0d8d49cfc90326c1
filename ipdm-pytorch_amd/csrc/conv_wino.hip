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
//     EIGHT positions -- rows i in {2 ih, 2 ih + 1}, ih = w & 1, all four j -- of one cout half h = w >> 1 for all 32
//     tiles: 8 accumulators of 32x32 (128 registers).  One 16-byte LDS read per operand and position feeds the four
//     K steps of a chunk (layouts below): 0.5 LDS instructions per MFMA.
//   * the producers' VALU job grows from GroupNorm+SiLU to GroupNorm+SiLU + B^T d B: a producer thread owns one
//     (tile, channel) of the chunk, loads its 4x4 patch itself (16 dword buffer loads with offsets fixed per tile; the
//     overlap between neighbouring patches is served by L1/L2), activates it, transforms it (32 add/sub) and writes the
//     16 positions' values into the stage.  The f32 MFMA occupies the SIMD's vector ALU, so this burst runs in the window
//     between barrier A and the hand-over in which the consumers wait (conv_ws.hip, issue model) -- it is additive.
//   * the output transform needs all 16 positions of a (tile, cout): in-lane over the wave's own 8 (j, then its two rows),
//     and ONE exchange of 2 values per (tile, cout) with the partner wave (w ^ 1) through LDS behind a third barrier per
//     TILE: wave ih = 0 finishes output row 0 of every 2x2 tile, wave ih = 1 row 1.  So a wave stores whole pixel rows
//     (8-byte stores, 128-byte segments) and the fused GroupNorm statistics keep conv_ws.hip's geometry: one row of
//     per-cout partial sums per pixel row and 32-pixel column block.
//   * bias: one extra MFMA into position (1, 1), whose output-transform coefficients are 1 for all four outputs.
//
// LDS stage (48 KB, two stages): V [xi 16][lk 2][tile 32][kp 4] and U [xi 16][h 2][lk 2][cout 32][kp 4], channel of a
// value = 2 kp + lk: the four K steps of a lane are one ds_read_b128.  Plus 32 KB exchange, 4 KB statistics staging.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int KC = 8;                                  // channels per K chunk (4 MFMA k-steps of 2)
constexpr int TH = 4, TW = 32, BN = 64;                // output pixels / couts of a workgroup tile
constexpr int V_FLOATS = 16 * 2 * 32 * 4;
constexpr int U_FLOATS = 16 * 2 * 2 * 32 * 4;
constexpr int STAGE = V_FLOATS + U_FLOATS;             // 12288 floats = 48 KB
constexpr int XCH_FLOATS = 4 * 8 * 64 * 4;             // per consumer wave: 8 x (64 lanes x 16 bytes)
constexpr int STAT_FLOATS = 4 * 256;
constexpr size_t LDS_BYTES = (size_t)(2 * STAGE + XCH_FLOATS + STAT_FLOATS) * sizeof(float);
constexpr int U_CHUNK_FLOATS = U_FLOATS;               // packed weights of one (chunk, cout tile): the U stage image

struct TileId { int n, oy0, ox0, co0; };

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

template <bool PLANAR>
__global__ void __launch_bounds__(512) conv_wino_kernel(ConvArgs a, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const xch = lds + 2 * STAGE;
    float *const stat_lds = xch + XCH_FLOATS;

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
        const int tid = threadIdx.x - 256;
        const int t = tid & 31, lk = (tid >> 5) & 1;                     // tile, channel parity
        const int kp = __builtin_amdgcn_readfirstlane(tid >> 6);         // k-step (wave-uniform): channel = 2 kp + lk
        const int ty = t >> 4, tx = t & 15;
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)a.w, 0, nchunks * a.co_tiles * U_CHUNK_FLOATS * 4, 0x00020000);
        // Patch offsets of this thread's (tile, channel parity), fixed for a tile.  A producer's VALU instructions outside
        // the window only get the stall gaps of the partner wave's MFMA stream (~280 cycles each), so interior tiles take
        // ONE per-lane base plus 16 wave-uniform constants (v_add with a scalar operand); only tiles that touch the image
        // border pay per-element range checks.  voffp: the same elements of a parity-planar x1 -- the patch origin is odd
        // in both axes (tile origins are even), so the parity pattern of the 16 elements is fixed too.
        int voff[16], voffp[PLANAR ? 16 : 1];
        unsigned okmask = 0xffffu;
        bool border = false;
        TileId tl = {0, 0, 0, 0};
        float *const vdst0 = lds + (lk * 32 + t) * 4 + kp;               // + xi * 256 + stage
        const int h2 = a.Hs >> 1, w2 = a.Ws >> 1;
        for (int s = 0; s < S; ++s) {
            const int k = s / nchunks, ch = s - k * nchunks;
            if (ch == 0) {
                tl = decode_tile(a, tile_of(k));
                const int iy0 = tl.oy0 - 1 + 2 * ty, ix0 = tl.ox0 - 1 + 2 * tx;
                border = tl.oy0 - 1 < 0 || tl.ox0 - 1 < 0 || tl.oy0 + TH + 1 > a.H || tl.ox0 + TW + 1 > a.W;
                if (!border) {
                    const int base = (iy0 * a.Ws + ix0) * 4 + lk * plane_bytes;
#pragma unroll
                    for (int e = 0; e < 16; ++e) voff[e] = base + ((e >> 2) * a.Ws + (e & 3)) * 4;
                    if (PLANAR) {
                        const int basep = (((iy0 - 1) >> 1) * w2 + ((ix0 - 1) >> 1)) * 4 + lk * plane_bytes;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int dy = e >> 2, dx = e & 3;      // element parity = (1 + d) & 1, plane index offset = (d + 1) >> 1
                            voffp[PLANAR ? e : 0] = basep + (((((1 + dy) & 1) * 2 + ((1 + dx) & 1)) * h2 + ((dy + 1) >> 1)) * w2 + ((dx + 1) >> 1)) * 4;
                        }
                    }
                    okmask = 0xffffu;
                } else {
                    okmask = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int iy = iy0 + (e >> 2), ix = ix0 + (e & 3);
                        const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                        okmask |= ok ? 1u << e : 0u;
                        voff[e] = ok ? (iy * a.Ws + ix) * 4 + lk * plane_bytes : OOB;
                        if (PLANAR) voffp[PLANAR ? e : 0] = ok ? ((((iy & 1) * 2 + (ix & 1)) * h2 + (iy >> 1)) * w2 + (ix >> 1)) * 4 + lk * plane_bytes : OOB;
                    }
                }
            }
            const int c0 = ch * KC;
            float *const stage = lds + (s & 1) * STAGE;
            const bool from1 = c0 < a.C1;
            const int csrc = from1 ? a.C1 : a.C2;
            const float *src = from1 ? a.x1 + (size_t)tl.n * a.C1 * (plane_bytes / 4) : a.x2 + (size_t)tl.n * a.C2 * (plane_bytes / 4);
            const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, csrc * plane_bytes, 0x00020000);
            const int cs0 = (from1 ? c0 : c0 - a.C1) + 2 * kp;               // (+ lk through the per-lane offset)
            // weights first (no transform: to LDS as soon as they land), then the raw patch
            f32x4 w_reg[8];
            const int w_soff = (ch * a.co_tiles + tl.co0 / BN) * (U_CHUNK_FLOATS * 4);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                w_reg[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (tid + e * 256) * 16, w_soff, 0));
            float d[16];
            if (PLANAR && from1) {       // (uniform branch) only x1 is stored parity-planar; the skip half of a concat is NCHW
#pragma unroll
                for (int e = 0; e < 16; ++e) d[e] = bload(x_rsrc, voffp[PLANAR ? e : 0], cs0 * plane_bytes);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) d[e] = bload(x_rsrc, voff[e], cs0 * plane_bytes);
            }
            float sc = 1.0f, sh = 0.0f;
            if (a.act) {
                sc = a.gn_scale[(size_t)tl.n * Ctot + c0 + 2 * kp + lk];
                sh = a.gn_shift[(size_t)tl.n * Ctot + c0 + 2 * kp + lk];
            }
            // stage (s&1) was last read by chunk s-2, which the consumers finished before the previous hand-over
#pragma unroll
            for (int e = 0; e < 8; ++e) *reinterpret_cast<f32x4 *>(stage + V_FLOATS + (tid + e * 256) * 4) = w_reg[e];
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            // E (the consumers' exchange barrier of the tile that just ended) already says "the matrix pipe is idle": the
            // window of a tile's first chunk then overlaps the consumers' store epilogue instead of following it
            const bool after_tile = ch == 0 && s > 0;
            if (after_tile) __syncthreads();               // E
            else __syncthreads();                          // A: consumers have finished chunk s-1
            // ---- the VALU window: GroupNorm(+SiLU), zero padding re-imposed, B^T d B, 16 LDS stores
            if (a.act) {
#pragma unroll
                for (int e = 0; e < 16; ++e) d[e] = fmaf(d[e], sc, sh);
                if (a.act == 2) {
                    float ex[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) ex[e] = d[e] * -1.4426950408889634f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) ex[e] = __builtin_amdgcn_exp2f(ex[e]);
#pragma unroll
                    for (int e = 0; e < 16; ++e) ex[e] = ex[e] + 1.0f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) ex[e] = __builtin_amdgcn_rcpf(ex[e]);
#pragma unroll
                    for (int e = 0; e < 16; ++e) d[e] = d[e] * ex[e];
                }
                if (border) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) d[e] = (okmask >> e & 1) ? d[e] : 0.0f;
                }
            }
            {
                float tt[16];      // B^T d: rows of the patch combined
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    tt[0 + x] = d[0 + x] - d[8 + x];
                    tt[4 + x] = d[4 + x] + d[8 + x];
                    tt[8 + x] = d[8 + x] - d[4 + x];
                    tt[12 + x] = d[4 + x] - d[12 + x];
                }
                float *vdst = vdst0 + (s & 1) * STAGE;
#pragma unroll
                for (int i = 0; i < 4; ++i) {                 // (B^T d) B: columns combined
                    vdst[(i * 4 + 0) * 256] = tt[i * 4 + 0] - tt[i * 4 + 2];
                    vdst[(i * 4 + 1) * 256] = tt[i * 4 + 1] + tt[i * 4 + 2];
                    vdst[(i * 4 + 2) * 256] = tt[i * 4 + 2] - tt[i * 4 + 1];
                    vdst[(i * 4 + 3) * 256] = tt[i * 4 + 1] - tt[i * 4 + 3];
                }
            }
            if (after_tile) __syncthreads();               // A (the consumers arrive after their stores)
            __syncthreads();                               // hand-over: stage (s&1) is complete
        }
        if (S > 0) __syncthreads();                        // E of the last tile
        return;
    }

    // =============================================================================== CONSUMERS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 5, l31 = lane & 31;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int ih = swave & 1, h = swave >> 1;              // which two rows of the transform domain, which cout half
    const int ty = l31 >> 4, tx = l31 & 15;                // the lane's tile inside the workgroup tile
    f32x16 acc[8];
    const int out_plane = a.Ho * a.Wo;
    const int plane4 = out_plane * 4;
    const int lane_off = (lk * 4 * out_plane + (2 * ty + ih) * a.Wo + 2 * tx) * 4;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.bias ? a.bias : a.out), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
    float nb = 0.0f;
    auto fetch_bias = [&](int k) __attribute__((always_inline)) {
        const TileId t = decode_tile(a, tile_of(k));
        nb = bload(b_rsrc, lk ? OOB : l31 * 4, (t.co0 + h * 32) * 4);
    };
    if (S > 0 && ih == 0) fetch_bias(0);
    const int a_off = V_FLOATS + ((((8 * ih) * 2 + h) * 2 + lk) * 32 + l31) * 4;      // + e * 512 floats
    const int b_off = (((8 * ih) * 2 + lk) * 32 + l31) * 4;                           // + e * 256 floats
    for (int s = 0; s < S; ++s) {
        const int k = s / nchunks, ch = s - k * nchunks;
        __syncthreads();                                   // A: chunk s-1 done -> the producers' burst may use the SIMD
        __syncthreads();                                   // hand-over: stage (s&1) is complete
        if (ch == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[e][r] = 0.0f;
        }
        const float *stage = lds + (s & 1) * STAGE;
        {
            f32x4 a_c = *reinterpret_cast<const f32x4 *>(stage + a_off), b_c = *reinterpret_cast<const f32x4 *>(stage + b_off), a_n, b_n;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e + 1 < 8) {
                    a_n = *reinterpret_cast<const f32x4 *>(stage + a_off + (e + 1) * 512);
                    b_n = *reinterpret_cast<const f32x4 *>(stage + b_off + (e + 1) * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[q], b_c[q], acc[e], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (e + 1 < 8) { a_c = a_n; b_c = b_n; }
            }
        }
        if (ch != nchunks - 1) continue;
        // ---------------------------------------------------------------- tile epilogue
        // + bias through position (1, 1) (accumulator 5 of the ih = 0 waves), whose output coefficients are all 1
        if (ih == 0) {
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(nb, 1.0f, acc[5], 0, 0, 0);
            if (k + 1 < n_my) fetch_bias(k + 1);
        }
        const TileId t = decode_tile(a, tile_of(k));
        // columns first (in-lane): T_i[b] = sum_j M[i][j] A[j][b],  A^T = [[1,1,1,0],[0,1,-1,-1]]
        // then the wave's own two rows: ih = 0 keeps P = T0 + T1 (output row 0) and sends T1; ih = 1 keeps R = T2 + T3
        // (output row 1 = T1 - R) and sends T2
        float keep[16][2];
        {
            float *xw = xch + (swave * 8) * 256 + lane * 4;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x4 snd;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float m0 = acc[0][r + u], m1 = acc[1][r + u], m2 = acc[2][r + u], m3 = acc[3][r + u];
                    const float n0 = acc[4][r + u], n1 = acc[5][r + u], n2 = acc[6][r + u], n3 = acc[7][r + u];
                    const float lo0 = (m0 + m1) + m2, lo1 = (m1 - m2) - m3;      // first of the wave's rows (i = 2 ih)
                    const float hi0 = (n0 + n1) + n2, hi1 = (n1 - n2) - n3;      // second (i = 2 ih + 1)
                    keep[r + u][0] = lo0 + hi0;
                    keep[r + u][1] = lo1 + hi1;
                    snd[2 * u] = ih == 0 ? hi0 : lo0;                            // T1 from ih = 0, T2 from ih = 1
                    snd[2 * u + 1] = ih == 0 ? hi1 : lo1;
                }
                *reinterpret_cast<f32x4 *>(xw + (r >> 1) * 256) = snd;
            }
        }
        __syncthreads();                                   // E: both halves of every (tile, cout) are in LDS
        const size_t sample = (size_t)t.n * a.Cout * out_plane;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + sample), 0, a.Cout * out_plane * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((a.res ? a.res : a.out) + sample), 0, a.Cout * out_plane * 4, 0x00020000);
        const int py = t.oy0 + 2 * ty + ih, px = t.ox0 + 2 * tx;
        const bool rok = py < a.Ho;
        const int voff2 = (rok && px + 1 < a.Wo) ? lane_off : OOB;               // both pixels of the lane's run
        const bool ragged = t.ox0 + TW > a.Wo && (a.Wo & 1) != 0;                // (wave-uniform) a run straddles the right edge
        const int voff1 = (ragged && rok && px + 1 == a.Wo) ? lane_off : OOB;    // ... then only its first pixel exists
        const int so0 = ((t.co0 + h * 32) * out_plane + min(t.oy0, a.Ho - 1) * a.Wo + t.ox0) * 4;
        const float *xr = xch + ((swave ^ 1) * 8) * 256 + lane * 4;
        float *sb = stat_lds + swave * 256;
        f32x2 rv[2][4];
        float rv1[2][4];
        auto load_res = [&](int g, f32x2 (&dst)[4], float (&dst1)[4]) __attribute__((always_inline)) {
            // registers 4 g .. 4 g + 3 are couts 8 g + {0..3} (+ 4 lk in lane_off)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int so = so0 + (8 * g + u) * plane4;
                dst[u] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, voff2, so, 0));
                if (ragged) dst1[u] = bload(r_rsrc, voff1, so);
            }
        };
        if (a.res) load_res(0, rv[0], rv1[0]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (a.res && g + 1 < 4) load_res(g + 1, rv[(g + 1) & 1], rv1[(g + 1) & 1]);
            const f32x4 got0 = *reinterpret_cast<const f32x4 *>(xr + (2 * g) * 256);
            const f32x4 got1 = *reinterpret_cast<const f32x4 *>(xr + (2 * g + 1) * 256);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * g + u;
                const float p0 = u < 2 ? got0[2 * u] : got1[2 * (u - 2)], p1 = u < 2 ? got0[2 * u + 1] : got1[2 * (u - 2) + 1];
                // output row 0 = (T0 + T1) + T2;  output row 1 = T1 - (T2 + T3)
                f32x2 y = ih == 0 ? f32x2{keep[r][0] + p0, keep[r][1] + p1} : f32x2{p0 - keep[r][0], p1 - keep[r][1]};
                const int so = so0 + (8 * g + u) * plane4;
                if (a.res) y += rv[g & 1][u];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, y), o_rsrc, voff2, so, 0);
                float y0 = y[0];
                if (ragged) {
                    y0 = ih == 0 ? keep[r][0] + p0 : p0 - keep[r][0];
                    if (a.res) y0 += rv1[g & 1][u];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y0), o_rsrc, voff1, so, 0);
                }
                if (a.stats) {
                    // fused GroupNorm statistics of the output: one row of per-cout {sum, sum of squares} per PIXEL ROW and
                    // 32-pixel column block, as conv_ws.hip writes them; the 16 lanes of a DPP row share (pixel row, cout)
                    const bool ok2 = voff2 != OOB, ok1 = ragged && voff1 != OOB;
                    float s1 = ok2 ? y[0] + y[1] : (ok1 ? y0 : 0.0f);
                    float s2 = ok2 ? fmaf(y[1], y[1], y[0] * y[0]) : (ok1 ? y0 * y0 : 0.0f);
                    s1 = sum_row16(s1);
                    s2 = sum_row16(s2);
                    if (tx == 0) *reinterpret_cast<f32x2 *>(sb + (ty * 32 + 8 * g + u + 4 * lk) * 2) = f32x2{s1, s2};
                }
            }
        }
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
    }
}

}  // namespace

namespace ipdm {

// Which convolutions run in the Winograd domain: wide 3x3 stride-1 layers with packed U weights, whole 64-cout tiles and
// whole 8-channel chunks, no resampling on the way in, not a K-split layer (those have too few tiles for any tiling).  The
// rule looks at the layer only, never at the batch size.
bool conv_wino_eligible(const ConvArgs &a)
{
    if (opt(OPT_CONV_NO_WINO) || !a.w_wino) return false;
    if (a.ksize != 3 || a.stride != 1 || a.upsample || a.H != a.Ho || a.W != a.Wo) return false;
    if (a.w_interleave != 2 && a.w_interleave != 4) return false;
    const int Ctot = a.C1 + a.C2;
    if (a.Cout % BN || Ctot % KC || Ctot < 32 || (a.C2 && a.C1 % KC)) return false;
    if ((a.x1_planar && ((a.Hs | a.Ws) & 1)) || conv_up2_eligible(a)) return false;
    return conv_ws_split(a) == 1;
}

bool conv_wino_shape_ok(int Cout, int Cin, int ks, int stride, int interleave)
{
    return ks == 3 && stride == 1 && (interleave == 2 || interleave == 4) && Cout % BN == 0 && Cin % KC == 0 && Cin >= 32;
}

// [chunk q][cout tile][xi][h][lk][cout 32][kp]: U = G g G^T in double, rounded once; channel = 8 q + 2 kp + lk
void conv_pack_weights_wino(const float *w, int Cout, int Cin, std::vector<float> &packed)
{
    const int nq = (Cin + KC - 1) / KC, nct = Cout / BN;
    packed.assign((size_t)nq * nct * U_CHUNK_FLOATS, 0.0f);
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
                    base[(((((i * 4 + j) * 2 + hh) * 2 + lk) * 32 + cl) * 4) + kp] = (float)u;
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
    a.ksplit = 1;
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
    const void *fn = a.x1_planar ? (const void *)conv_wino_kernel<true> : (const void *)conv_wino_kernel<false>;
    if (int rc = ensure_dynamic_lds(fn, LDS_BYTES)) return rc;
    const bool prof = prof_enabled();
    if (prof) prof_before(3, st);
    if (a.x1_planar) hipLaunchKernelGGL(conv_wino_kernel<true>, dim3((unsigned)G), dim3(512), LDS_BYTES, st, a, (int)ntiles);
    else hipLaunchKernelGGL(conv_wino_kernel<false>, dim3((unsigned)G), dim3(512), LDS_BYTES, st, a, (int)ntiles);
    // EXECUTED flops: 16 multiply-adds per 2x2 output tile and (cin, cout) pair (the 3x3 form counts 36)
    if (prof) prof_after(3, 2.0 * a.B * (double)cdiv(a.Ho, 2) * cdiv(a.Wo, 2) * 16.0 * a.Cout * (a.C1 + a.C2), st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

}  // namespace ipdm
