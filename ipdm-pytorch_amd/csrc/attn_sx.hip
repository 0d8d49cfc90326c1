// Opt-in split-bf16 evaluation of the attention core (IPDM_ATTN_SPLIT=3; the default stays the exact-f32 kernel of
// attn.hip).  Same mathematics and tiling idea as attention_ws_kernel -- flash-style, swapped QK^T, the score
// accumulators are the B operand of P.V without data movement -- but every f32 operand is carried as three bf16 pieces
// (x = x1 + x2 + x3, 3 x 8 significand bits = exact) and a product is six v_mfma_f32_32x32x16_bf16 terms accumulated in
// f32, smallest first: (a1+a2+a3)(b1+b2+b3) minus the three terms below 2^-24.  DESIGN.md section 6c.
//
// Differences forced / allowed by the bf16 matrix pipe (tools/ubench/coissue_bf16.hip: it leaves the vector ALU free):
//   * 12 waves per CU: 8 consumer waves (32 queries each, two per SIMD, so one wave's softmax + P-splitting VALU runs
//     under the other's MFMAs) + 4 producer waves that split K (pre-scaled) and V into pieces while staging them;
//   * K pieces are staged [key][64 ch] and V pieces [ch][64 keys] in bf16, pitch 72 (144 B: conflict-free 16-byte reads);
//     a lane's MFMA operand (8 consecutive k values) is one ds_read_b128;
//   * the P.V contraction runs over the 32 keys of a block in the order the score accumulators hold them: lane-half h,
//     registers 8t..8t+7 = keys 16t + {0..3, 8..11} + 4h; V is staged with its key axis permuted the same way
//     (key 16t + 8u + 4h + i  ->  position 16t + 8h + 4u + i), so P never moves between lanes.
#include <cstdlib>
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int D = 64;           // head dim (both reference configurations)
constexpr int KV = 64;          // keys per LDS tile
constexpr int PITCH = 72;       // bf16 elements per staged row
constexpr int PIECE = 64 * PITCH;                 // one piece of K ([64 keys][72]) or of V ([64 ch][72])
constexpr int STAGE = 6 * PIECE;                  // K1 K2 K3 V1 V2 V3
constexpr int NCONS = 8;        // consumer waves
constexpr float LOG2E = 1.4426950408889634f;

__device__ inline int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ inline unsigned bf16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ inline float bf16_val(unsigned bits) { return __builtin_bit_cast(float, bits << 16); }

// x (8 floats) -> three packed bf16x8 operands
__device__ inline void split8(const float (&x)[8], u32x4 (&p)[3])
{
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = x[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const unsigned lo = bf16_bits(r[j]), hi = bf16_bits(r[j + 1]);
            w[j / 2] = lo | (hi << 16);
            if (i < 2) {
                r[j] -= bf16_val(lo);
                r[j + 1] -= bf16_val(hi);
            }
        }
        p[i] = u32x4{w[0], w[1], w[2], w[3]};
    }
}

// The same split by truncation, for the probabilities in the consumers' inner loop (4.5 instead of ~7 VALU ops per
// element; a partner wave's VALU op costs ~8 cycles under the MFMA stream): piece 1 = the upper 16 bits of x, the
// residual x - piece1 is exact, piece 2 = its upper 16 bits, piece 3 = the (exact, <= 8-bit) rest.  The three pieces
// again sum to x exactly.
__device__ inline void split8_trunc(const float (&x)[8], u32x4 (&p)[3])
{
    unsigned w[3][4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const unsigned a0 = __builtin_bit_cast(unsigned, x[j]), a1 = __builtin_bit_cast(unsigned, x[j + 1]);
        const float r0 = x[j] - __builtin_bit_cast(float, a0 & 0xffff0000u);
        const float r1 = x[j + 1] - __builtin_bit_cast(float, a1 & 0xffff0000u);
        const unsigned b0 = __builtin_bit_cast(unsigned, r0), b1 = __builtin_bit_cast(unsigned, r1);
        const float q0 = r0 - __builtin_bit_cast(float, b0 & 0xffff0000u);
        const float q1 = r1 - __builtin_bit_cast(float, b1 & 0xffff0000u);
        const unsigned c0 = __builtin_bit_cast(unsigned, q0), c1 = __builtin_bit_cast(unsigned, q1);
        // pack the upper halves: (hi16(x1) << 16) | hi16(x0)
        w[0][j / 2] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
        w[1][j / 2] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        w[2][j / 2] = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = u32x4{w[i][0], w[i][1], w[i][2], w[i][3]};
}

#define SX6(acc, A, B)                                                                                                  \
    do {                                                                                                                \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[2]), __builtin_bit_cast(bf16x8, B[0]), acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[1]), __builtin_bit_cast(bf16x8, B[1]), acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[0]), __builtin_bit_cast(bf16x8, B[2]), acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[1]), __builtin_bit_cast(bf16x8, B[0]), acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[0]), __builtin_bit_cast(bf16x8, B[1]), acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[0]), __builtin_bit_cast(bf16x8, B[0]), acc, 0, 0, 0); \
    } while (0)

// Pre-pass: K (scaled) and V of every (sample, head) split into bf16 pieces ONCE, written as the exact LDS image of
// each 64-key tile (K1 K2 K3 [key][72], V1 V2 V3 [ch][72 permuted keys]); the main kernel's producers then only copy.
// (Splitting inside the main kernel repeated this VALU work for each of the T/256 query workgroups and left it
// VALU-bound: under a bf16 MFMA stream a partner wave's VALU op costs ~8 cycles.)
__global__ void __launch_bounds__(256) attention_sx_split_kernel(const float *__restrict__ qkv,
                                                                 unsigned short *__restrict__ pieces, int heads, int T,
                                                                 float scale)
{
    __shared__ __attribute__((aligned(16))) unsigned short st[STAGE];
    const int bh = blockIdx.y, it = blockIdx.x;
    const int b = bh / heads, head = bh % heads;
    const float *kp = qkv + ((size_t)b * heads * 3 * D + (size_t)head * 3 * D + D) * T;
    const float *vp = kp + (size_t)D * T;
    const int ntiles = (T + KV - 1) / KV;
    const int tid = threadIdx.x;
    const int s = tid & 63, cg = tid >> 6;              // this thread's key and its 16 channels cg*16 .. +15
    const int vpos = (s & 0x33) | ((s & 4) << 1) | ((s & 8) >> 1);      // position of key s on V's permuted key axis
    const int s0 = it * KV;
    const bool ok = (s0 + s) < T;
    const size_t col = (size_t)min(s0 + s, T - 1);
    float kr[16], vr[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const size_t g = (size_t)(cg * 16 + e) * T + col;
        kr[e] = ok ? kp[g] * scale : 0.0f;
        vr[e] = ok ? vp[g] : 0.0f;
    }
    for (int i = tid; i < STAGE / 2; i += 256) reinterpret_cast<unsigned *>(st)[i] = 0u;        // the row padding
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float x[8];
        u32x4 p[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = kr[half * 8 + j];
        split8(x, p);
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4 *>(st + i * PIECE + s * PITCH + cg * 16 + half * 8) = p[i];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float r = vr[e];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned bits = bf16_bits(r);
            st[(3 + i) * PIECE + (cg * 16 + e) * PITCH + vpos] = (unsigned short)bits;
            if (i < 2) r -= bf16_val(bits);
        }
    }
    __syncthreads();
    constexpr int CHUNKS = STAGE * 2 / 16;
    u32x4 *dst = reinterpret_cast<u32x4 *>(pieces) + ((size_t)bh * ntiles + it) * CHUNKS;
    for (int i = tid; i < CHUNKS; i += 256) dst[i] = reinterpret_cast<const u32x4 *>(st)[i];
}

__global__ void __launch_bounds__(768) attention_sx_kernel(const float *__restrict__ qkv,
                                                           const unsigned short *__restrict__ pieces,
                                                           float *__restrict__ out, int heads, int T, float scale)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int bh = blockIdx.y;
    const int b = bh / heads, head = bh % heads;
    const float *qp = qkv + ((size_t)b * heads * 3 * D + (size_t)head * 3 * D) * T;
    const int ntiles = (T + KV - 1) / KV;

    if (threadIdx.x >= NCONS * 64) {
        // ------------------------------------------------------------------ producers: copy the pre-split tile image
        const int tid = threadIdx.x - NCONS * 64;
        constexpr int CHUNKS = STAGE * 2 / 16;              // 16-byte chunks per tile image
        const u32x4 *src = reinterpret_cast<const u32x4 *>(pieces) + (size_t)bh * ntiles * CHUNKS;
        for (int it = 0; it < ntiles; ++it) {
            u32x4 *dst = reinterpret_cast<u32x4 *>(smem + (it & 1) * STAGE);
            u32x4 r[(CHUNKS + 255) / 256];
#pragma unroll
            for (int k = 0; k < (CHUNKS + 255) / 256; ++k) {
                const int i = tid + k * 256;
                if (i < CHUNKS) r[k] = src[(size_t)it * CHUNKS + i];
            }
            // stage (it&1) was last read for tile it-2, which the consumers finished before the previous hand-over
#pragma unroll
            for (int k = 0; k < (CHUNKS + 255) / 256; ++k) {
                const int i = tid + k * 256;
                if (i < CHUNKS) dst[i] = r[k];
            }
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int t0 = blockIdx.x * (32 * NCONS) + wave * 32;
    const int tq = t0 + l31;
    // Q pieces: k-step s contracts channels 16 s + 8 lh + j
    u32x4 qpc[4][3];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = tq < T ? qp[(size_t)(16 * s + 8 * lh + j) * T + tq] * scale : 0.0f;
        split8(x, qpc[s]);
    }
    f32x16 o[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    for (int it = 0; it < ntiles; ++it) {
        const int s0 = it * KV;
        __syncthreads();                           // hand-over: stage (it&1) is complete
        const unsigned short *st = smem + (it & 1) * STAGE;
#pragma unroll
        for (int sb = 0; sb < KV / 32; ++sb) {
            if (s0 + sb * 32 >= T) break;          // wave-uniform
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
            const unsigned short *krow = st + (sb * 32 + l31) * PITCH + 8 * lh;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4 ka[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) ka[i] = *reinterpret_cast<const u32x4 *>(krow + i * PIECE + 16 * s);
                SX6(sacc, ka, qpc[s]);
            }
            if (__builtin_amdgcn_readfirstlane(s0 + sb * 32 + 32 > T)) {        // scalar branch: only the ragged last block
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (s0 + sb * 32 + crow(r, lh) >= T) sacc[r] = -INFINITY;
            }
            float mx = sacc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float mb = -m_new * LOG2E;
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
            float rs = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r], LOG2E, mb));
                rs += sacc[r];
            }
            rs += __shfl_xor(rs, 32, 64);
            l_run = l_run * alpha + rs;
            m_run = m_new;
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
            }
            // P pieces straight from the score registers: k-step t = registers 8t .. 8t+7
            u32x4 pp[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = sacc[8 * t + j];
                split8_trunc(x, pp[t]);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const unsigned short *vrow = st + 3 * PIECE + (cb * 32 + l31) * PITCH + sb * 32 + 8 * lh;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    u32x4 va[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) va[i] = *reinterpret_cast<const u32x4 *>(vrow + i * PIECE + 16 * t);
                    SX6(o[cb], va, pp[t]);
                }
            }
        }
    }
    float *op = out + ((size_t)b * heads * D + (size_t)head * D) * T;
    if (tq < T) {
        const float inv = 1.0f / l_run;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) op[(size_t)(cb * 32 + crow(r, lh)) * T + tq] = o[cb][r] * inv;
    }
}

}  // namespace

namespace ipdm {

static size_t attention_sx_scratch_floats_impl(int B, int heads, int T)
{
    return (size_t)B * heads * cdiv(T, KV) * STAGE / 2;        // bf16 tile images, counted in floats
}

static int attention_sx_launch_impl(const float *qkv, float *scratch, float *out, int B, int heads, int T, float scale, hipStream_t st)
{
    IPDM_REQUIRE(scratch, "attention (split-bf16): no scratch for the pre-split K/V pieces");
    constexpr size_t lds = (size_t)2 * STAGE * sizeof(unsigned short);
    if (int rc = ensure_dynamic_lds((const void *)attention_sx_kernel, lds)) return rc;
    unsigned short *pieces = reinterpret_cast<unsigned short *>(scratch);
    hipLaunchKernelGGL(attention_sx_split_kernel, dim3(cdiv(T, KV), B * heads), dim3(256), 0, st, qkv, pieces, heads, T, scale);
    dim3 grid(cdiv(T, 32 * NCONS), B * heads);
    hipLaunchKernelGGL(attention_sx_kernel, grid, dim3(768), lds, st, qkv, pieces, out, heads, T, scale);
    return IPDM_OK;
}

}  // namespace ipdm

// entry points of libipdm_hip_optin.so (csrc/optin.hip)
extern "C" size_t ipdm_optin_attention_sx_scratch_floats(int B, int heads, int T) { return ipdm::attention_sx_scratch_floats_impl(B, heads, T); }
extern "C" int ipdm_optin_attention_sx_launch(const float *qkv, float *scratch, float *out, int B, int heads, int T, float scale, hipStream_t st)
{
    return ipdm::attention_sx_launch_impl(qkv, scratch, out, B, heads, T, scale, st);
}
