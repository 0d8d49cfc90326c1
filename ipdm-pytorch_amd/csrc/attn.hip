// Self-attention core of AttentionBlock (Model/model.py:148-153) for gfx950: flash-style, scores
// never leave registers/LDS (the reference materialises a T x T score tensor: 268-812 MB per
// sample and block), exact-f32 MFMA for both contractions.
//
//   qkv [B, heads*3*d, T] with per-head (q,k,v) chunks (the reshape/chunk of :148), d = 64
//   S[s,t]  = sum_c (k[c,s]*scale) * (q[c,t]*scale),  scale = d^(-1/4)            (:149-150)
//   P       = softmax over keys s                                                     (:151)
//   out[c,t]= sum_s v[c,s] * P[s,t]                                                   (:152)
//
// Mapping: a workgroup = 4 waves = 128 queries of one (sample, head); each wave owns 32 queries.
// QK^T is computed "swapped" (keys on the MFMA row index, queries on the lane) so that
//   * a lane holds 16 of the 32 scores of ITS query per key block -> row max / row sum are 16
//     in-register ops + one cross-half shuffle;
//   * the score accumulator registers are directly the B operand (k-pair = keys crow(r,0),crow(r,1))
//     of the P.V MFMA, with V read from LDS as the A operand -- no data movement for P at all;
//   * the output accumulator has the query on the lane too, so the online-softmax rescale is a
//     per-lane multiply and the final store is coalesced along t.
// Issue model (tools/ubench/coissue.hip): the f32 MFMA holds its wave's issue for its whole 64 cycles, so every
// LDS read and VALU op of this kernel is ADDED to the MFMA time.  Hence
//   * operands come in 16-byte LDS reads, 4 MFMAs per read: K is staged key-major [s][c] (the contraction runs over
//     channels in the order c = 32*half + p, which makes a lane's 4 consecutive k-steps adjacent), V channel-major
//     [c][s] (a lane's 4 consecutive P registers are 4 consecutive keys); pitch 68 keeps both conflict-free;
//   * softmax is 4 VALU ops per score: max, fma, v_exp_f32 (base 2, log2(e) folded into the fma), add;
//   * out-of-range keys are masked only in the ragged last block.
#include <cstdlib>
#include "unet_kernels.h"

using namespace ipdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int KV = 64;         // keys per LDS tile
constexpr int KP = 68;         // K pitch  [s][c]  (16-byte aligned rows, conflict-free b128 reads)
constexpr int VP = 68;         // V pitch  [c][s]
constexpr float LOG2E = 1.4426950408889634f;

__device__ inline int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Maximum / sum over the two halves of the wave (lanes l and l ^ 32) by v_permlane32_swap (gfx950): one VALU instruction
// turns two copies of x into {lo, lo} and {hi, hi}.  As __shfl_xor(x, 32) the exchange is a ds_bpermute, an LDS round trip
// of ~100 cycles that the softmax waits for once per key block (the row maximum feeds every exponential).  Inline
// assembly: the builtin mis-pairs its two results when both inputs are the same value; the two wait states are the
// VALU-write -> permlane-read hazard the compiler would insert itself.
__device__ inline float halves_max(float x)
{
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ inline float halves_sum(float x)
{
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

template <int D, int QT>   // D: head dim C/heads (64 in both reference configs, 32 for reduced test nets); QT: 32-query tiles per wave
__global__ void __launch_bounds__(256, 2) attention_kernel(const float *__restrict__ qkv, float *__restrict__ out,
                                                           int heads, int T, float scale)
{
    __shared__ __attribute__((aligned(16))) float k_lds[KV * KP];
    __shared__ __attribute__((aligned(16))) float v_lds[D * VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int bh = blockIdx.y;                      // sample*heads + head
    const int b = bh / heads, head = bh % heads;
    const float *qp = qkv + ((size_t)b * heads * 3 * D + (size_t)head * 3 * D) * T;
    const float *kp = qp + (size_t)D * T;
    const float *vp = qp + (size_t)2 * D * T;
    const int t0 = blockIdx.x * (128 * QT) + wave * (32 * QT);
    constexpr int HP = D / 2;                       // k-steps; step p of half lh contracts channel lh*HP + p

    // Q as the B operand of S = K^T Q: lane holds q[lh*HP + p][t] * scale, for QT query tiles (each LDS operand read
    // then feeds QT MFMAs instead of one)
    float qreg[QT][HP];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int t = t0 + qt * 32 + l31;
#pragma unroll
        for (int p = 0; p < HP; ++p) qreg[qt][p] = t < T ? qp[(size_t)(lh * HP + p) * T + t] * scale : 0.0f;
    }

    constexpr int CB = D / 32;
    f32x16 o[QT][CB];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][cb][r] = 0.0f;
    float m_run[QT], l_run[QT];                     // running max (natural-log domain scores), running sum
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.0f; }

    for (int s0 = 0; s0 < T; s0 += KV) {
        __syncthreads();   // previous tile fully consumed
        // stage K (scaled, transposed to [s][c]) and V ([c][s]); global reads are coalesced along the keys
#pragma unroll
        for (int e = 0; e < (D * KV) / 256; ++e) {
            const int idx = tid + e * 256;
            const int c = idx / KV, s = idx % KV;
            const bool ok = (s0 + s) < T;
            k_lds[s * KP + c] = ok ? kp[(size_t)c * T + s0 + s] * scale : 0.0f;
            v_lds[c * VP + s] = ok ? vp[(size_t)c * T + s0 + s] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int sb = 0; sb < KV / 32; ++sb) {
            if (s0 + sb * 32 >= T) break;          // wave-uniform
            // ---- S[s, t] for 32 keys x (QT x 32) queries
            f32x16 sacc[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[qt][r] = 0.0f;
            const float *krow = k_lds + (sb * 32 + l31) * KP + lh * HP;
#pragma unroll
            for (int g = 0; g < HP / 4; ++g) {
                const f32x4 kv = *reinterpret_cast<const f32x4 *>(krow + g * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        sacc[qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[j], qreg[qt][g * 4 + j], sacc[qt], 0, 0, 0);
            }
            // ---- online softmax over keys (rows), per query (lane & 31)
            const bool ragged = s0 + sb * 32 + 32 > T;     // last block only (wave-uniform)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (ragged) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (s0 + sb * 32 + crow(r, lh) >= T) sacc[qt][r] = -INFINITY;
                }
                float mx = sacc[qt][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[qt][r]);
                mx = halves_max(mx);
                const float m_new = fmaxf(m_run[qt], mx);
                const float mb = -m_new * LOG2E;
                // exactly 1 while the running max stands (an fma against the rounded mb would leave a 1e-6 residual that
                // compounds over the ~T/32 blocks); m_run = -inf on the first block -> 0
                const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * LOG2E);
                float rs = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[qt][r] = __builtin_amdgcn_exp2f(fmaf(sacc[qt][r], LOG2E, mb));     // exp(s - m_new)
                    rs += sacc[qt][r];
                }
                rs = halves_sum(rs);
                l_run[qt] = l_run[qt] * alpha + rs;
                m_run[qt] = m_new;
                // the running max settles after the first few key blocks: skip the 32 multiplies by exactly 1.0 (wave-uniform)
                if (__any(alpha != 1.0f)) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qt][cb][r] *= alpha;
                }
            }
            // ---- O[c, t] += sum_s V[c, s] P[s, t]; accumulator register r of P is the k-pair
            //      (s = crow(r,0) for lanes 0-31, crow(r,1) for lanes 32-63): registers 4g..4g+3 are 4 consecutive keys
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const float *vrow = v_lds + (cb * 32 + l31) * VP + sb * 32 + 4 * lh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 vv = *reinterpret_cast<const f32x4 *>(vrow + 8 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
                            o[qt][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[j], sacc[qt][4 * g + j], o[qt][cb], 0, 0, 0);
                }
            }
        }
    }
    float *op = out + ((size_t)b * heads * D + (size_t)head * D) * T;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int t = t0 + qt * 32 + l31;
        if (t < T) {
            const float inv = 1.0f / l_run[qt];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = cb * 32 + crow(r, lh);
                    op[(size_t)c * T + t] = o[qt][cb][r] * inv;
                }
        }
    }
}

// Wave-specialised variant (same arithmetic, same instruction order per query): waves 0-3 are consumers (Q in
// registers, MFMAs, softmax), waves 4-7 stage the NEXT K/V tile into the other LDS stage while the consumers work on the
// current one -- global loads, LDS stores and the barrier latency leave the MFMA waves; one hand-over barrier per tile.
// The only producer VALU is the 16 multiplies of K by the scale per tile (they fit the MFMA wave's stall gaps).
// Key slices (short sequences, attention_kv_split): every slice is reduced with a FRESH running maximum / sum / output, and
// the slices are folded in ascending order by this recurrence -- by the workgroup itself when it walks all slices of its
// queries (zseq), or by attention_combine_kernel when the slices ran as separate workgroups (zsplit).  The same float
// operations in the same order in both, so how a launch is scheduled (it depends on the batch size) never changes a bit.
//   M' = max(M, m_k);  a = 2^((M - M') log2e);  b = 2^((m_k - M') log2e);  num = num a + o_k b;  den = den a + l_k b
struct SliceWeights { float a, b; };
__device__ inline SliceWeights slice_weights(float &M, float m_k)
{
    const float Mn = fmaxf(M, m_k);
    SliceWeights w;
    w.a = __builtin_amdgcn_exp2f((M - Mn) * LOG2E);        // first slice: M = -inf -> 0
    w.b = __builtin_amdgcn_exp2f((m_k - Mn) * LOG2E);
    M = Mn;
    return w;
}
__device__ inline float slice_fold(float acc, float v, SliceWeights w) { return fmaf(acc, w.a, v * w.b); }

template <int D, int QT, bool ZSEQ = false>      // ZSEQ: a workgroup walks all key slices of its queries (zseq > 1)
__global__ void __launch_bounds__(512, ZSEQ ? 2 : 1) attention_ws_kernel(const float *__restrict__ qkv, float *__restrict__ out,
                                                           int heads, int T, float scale, int zsplit, float *__restrict__ part, int zseq)
{
    // zsplit > 1 (few queries: batch 1, low resolutions): blockIdx.z takes a slice of the key tiles and leaves its
    // UNNORMALISED output, running maximum and sum in `part`; attention_combine_kernel merges the slices.
    // zseq > 1 (the same layer with enough queries to fill the chip): this workgroup walks all zseq slices itself
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = KV * KP + D * VP;
    const int bh = blockIdx.y;                      // sample*heads + head
    const int b = bh / heads, head = bh % heads;
    const float *qp = qkv + ((size_t)b * heads * 3 * D + (size_t)head * 3 * D) * T;
    const int ntiles = (T + KV - 1) / KV;
    const int nslice = ZSEQ ? zseq : zsplit;
    const int tps = (ntiles + nslice - 1) / nslice;                 // key tiles per slice
    const int it0 = ZSEQ ? 0 : blockIdx.z * tps, it1 = ZSEQ ? ntiles : min(ntiles, it0 + tps);

    if (threadIdx.x >= 256) {
        // ------------------------------------------------------------------ producers
        // VALU-free inside the tile loop (the MFMA waves leave a producer's vector ALU only their stall gaps): buffer
        // loads take a per-thread byte offset that is fixed for the whole kernel plus a scalar offset per tile, LDS
        // stores take one base register per stage plus immediates.  The scale is folded into Q (consumers), K is
        // staged as it is.  Keys beyond T exist only in the last tile: their lanes get an out-of-range offset (= 0).
        const int tid = threadIdx.x - 256;
        constexpr int NE = (D * KV) / 256;                  // elements of K (and of V) per thread per tile
        const int s_l = tid % KV, c_l = tid / KV;           // element e of the thread: key s_l, channel c_l + 4 e
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)qp, 0, 3 * D * T * 4, 0x00020000);
        int voff[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) voff[e] = ((c_l + (256 / KV) * e) * T + s_l) * 4;
        const int k_base = s_l * KP + c_l, v_base = c_l * VP + s_l;       // + (256/KV) e  resp.  + (256/KV) e VP
        for (int it = it0; it < it1; ++it) {
            const int s0 = it * KV;
            float *k_lds = smem + (it & 1) * STAGE, *v_lds = k_lds + KV * KP;
            const bool last = s0 + KV > T;                  // wave-uniform
            float kr[NE], vr[NE];
            if (!last) {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    kr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[e], (D * T + s0) * 4, 0));
                    vr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[e], (2 * D * T + s0) * 4, 0));
                }
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int vo = s0 + s_l >= T ? 0x7fffffff : voff[e];
                    kr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (D * T + s0) * 4, 0));
                    vr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (2 * D * T + s0) * 4, 0));
                }
            }
            // stage (it&1) was last read for tile it-2, which the consumers finished before the previous hand-over
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                k_lds[k_base + (256 / KV) * e] = kr[e];
                v_lds[v_base + (256 / KV) * e * VP] = vr[e];
            }
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int t0 = blockIdx.x * (128 * QT) + wave * (32 * QT);
    constexpr int HP = D / 2;
    float qreg[QT][HP];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int t = t0 + qt * 32 + l31;
#pragma unroll
        for (int p = 0; p < HP; ++p) qreg[qt][p] = t < T ? qp[(size_t)(lh * HP + p) * T + t] * (scale * scale) : 0.0f;   // both d^(-1/4) factors
    }
    constexpr int CB = D / 32;
    f32x16 o[QT][CB];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][cb][r] = 0.0f;
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.0f; }
    // ZSEQ: the fold of the finished slices (QT = 1 only, like zsplit)
    static_assert(!ZSEQ || QT == 1, "attention: key slices exist for 32-query waves");
    f32x16 num[ZSEQ ? CB : 1];
    float M_all = -INFINITY, den = 0.0f;
#pragma unroll
    for (int cb = 0; cb < (ZSEQ ? CB : 1); ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) num[cb][r] = 0.0f;
    int slice_end = it0 + tps;                     // first tile of the next slice

    for (int it = it0; it < it1; ++it) {
        const int s0 = it * KV;
        __syncthreads();                           // hand-over: stage (it&1) is complete
        const float *k_lds = smem + (it & 1) * STAGE, *v_lds = k_lds + KV * KP;
#pragma unroll
        for (int sb = 0; sb < KV / 32; ++sb) {
            if (s0 + sb * 32 >= T) break;          // wave-uniform
            f32x16 sacc[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[qt][r] = 0.0f;
            const float *krow = k_lds + (sb * 32 + l31) * KP + lh * HP;
#pragma unroll
            for (int g = 0; g < HP / 4; ++g) {
                const f32x4 kv = *reinterpret_cast<const f32x4 *>(krow + g * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        sacc[qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[j], qreg[qt][g * 4 + j], sacc[qt], 0, 0, 0);
            }
            const bool ragged = s0 + sb * 32 + 32 > T;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (ragged) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (s0 + sb * 32 + crow(r, lh) >= T) sacc[qt][r] = -INFINITY;
                }
                float mx = sacc[qt][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[qt][r]);
                mx = halves_max(mx);
                const float m_new = fmaxf(m_run[qt], mx);
                const float mb = -m_new * LOG2E;
                const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * LOG2E);
                // exp(s - m_new), two scores per packed multiply-add / add (the softmax is VALU beside the MFMA stream)
                f32x2 rs2 = {0.0f, 0.0f};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 v = f32x2{sacc[qt][r], sacc[qt][r + 1]} * LOG2E + mb;
                    v[0] = __builtin_amdgcn_exp2f(v[0]);
                    v[1] = __builtin_amdgcn_exp2f(v[1]);
                    sacc[qt][r] = v[0];
                    sacc[qt][r + 1] = v[1];
                    rs2 += v;
                }
                float rs = halves_sum(rs2[0] + rs2[1]);
                l_run[qt] = l_run[qt] * alpha + rs;
                m_run[qt] = m_new;
                if (__any(alpha != 1.0f)) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qt][cb][r] *= alpha;
                }
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const float *vrow = v_lds + (cb * 32 + l31) * VP + sb * 32 + 4 * lh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 vv = *reinterpret_cast<const f32x4 *>(vrow + 8 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
                            o[qt][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[j], sacc[qt][4 * g + j], o[qt][cb], 0, 0, 0);
                }
            }
        }
        if constexpr (ZSEQ) if (it + 1 == slice_end || it + 1 == it1) {
            // the slice is complete: fold it (ascending order) and start the next one with a fresh state
            const SliceWeights w = slice_weights(M_all, m_run[0]);
            den = slice_fold(den, l_run[0], w);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) { num[cb][r] = slice_fold(num[cb][r], o[0][cb][r], w); o[0][cb][r] = 0.0f; }
            m_run[0] = -INFINITY;
            l_run[0] = 0.0f;
            slice_end += tps;
        }
    }
    if (zsplit > 1) {
        float *pp = part + ((size_t)blockIdx.z * gridDim.y + bh) * (D + 2) * T;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int t = t0 + qt * 32 + l31;
            if (t < T) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pp[(size_t)(cb * 32 + crow(r, lh)) * T + t] = o[qt][cb][r];
                if (lh == 0) { pp[(size_t)D * T + t] = m_run[qt]; pp[(size_t)(D + 1) * T + t] = l_run[qt]; }
            }
        }
        return;
    }
    float *op = out + ((size_t)b * heads * D + (size_t)head * D) * T;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int t = t0 + qt * 32 + l31;
        if (t < T) {
            constexpr bool seq = ZSEQ;
            const float inv = 1.0f / (seq ? den : l_run[qt]);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = cb * 32 + crow(r, lh);
                    op[(size_t)c * T + t] = (seq ? num[seq ? cb : 0][r] : o[qt][cb][r]) * inv;
                }
        }
    }
}

// merges the key slices of a split launch (the recurrence above, ascending slices).
// Block = (256 queries, sample*head, 16 of the D channels): the weights are recomputed per channel group (2 Z loads) so that
// four times as many blocks stream the partial outputs.
template <int D>
__global__ void __launch_bounds__(256) attention_combine_kernel(const float *__restrict__ part, float *__restrict__ out, int T, int Z, int ntiles)
{
    const int t = blockIdx.x * 256 + threadIdx.x, bh = blockIdx.y, c0 = blockIdx.z * 16;
    if (t >= T) return;
    const size_t slice = (size_t)gridDim.y * (D + 2) * T;
    const float *p = part + (size_t)bh * (D + 2) * T;
    const int tps = (ntiles + Z - 1) / Z;
    // the recurrence of slice_weights / slice_fold over the non-empty slices, ascending: exactly what a workgroup that
    // walks all slices itself (zseq) computes
    SliceWeights w[8];
    float M = -INFINITY, den = 0.0f;
#pragma unroll
    for (int z = 0; z < 8; ++z)
        if (z < Z && z * tps < ntiles) {
            w[z] = slice_weights(M, p[z * slice + (size_t)D * T + t]);
            den = slice_fold(den, p[z * slice + (size_t)(D + 1) * T + t], w[z]);
        }
    const float inv = 1.0f / den;
#pragma unroll 4
    for (int c = c0; c < c0 + 16; ++c) {
        float acc = 0.0f;
#pragma unroll
        for (int z = 0; z < 8; ++z)
            if (z < Z && z * tps < ntiles) acc = slice_fold(acc, p[z * slice + (size_t)c * T + t], w[z]);
        out[((size_t)bh * D + c) * T + t] = acc * inv;
    }
}

// Key slices of the wave-specialised kernel.  Like the K split of the convolutions the count depends on the layer alone
// (T and the head count, never the batch size), so that a slice of a batch stays bit-equal to the slice sampled alone:
// only short sequences are split (at most 128 query workgroups per sample: T <= 4096 with 4 heads -- one slice alone fills
// half of the chip; at 8 slices per GPU the split costs those launches a few per cent for the combine pass).
int attention_kv_split(int B, int heads, int d, int T)
{
    const bool off = opt(OPT_ATTN_NO_KVSPLIT) != 0;
    (void)B;
    if (off || d != 64) return 1;
    const long wg = (long)cdiv(T, 128) * heads;
    if (wg > 128) return 1;
    const int ntiles = cdiv(T, KV);
    int Z = wg <= 32 ? 8 : (wg <= 64 ? 4 : 2);
    if (Z > ntiles / 2) Z = ntiles / 2;
    return Z < 2 ? 1 : Z;
}

}  // namespace

namespace ipdm {

size_t attention_scratch_floats(int B, int heads, int d, int T)
{
    const int Z = attention_kv_split(B, heads, d, T);
    return Z > 1 ? (size_t)Z * B * heads * (d + 2) * T : 0;
}

int attention_launch(const float *qkv, float *out, int B, int heads, int d, int T, hipStream_t st, float *scratch)
{
    IPDM_REQUIRE(qkv && out && B > 0 && heads > 0 && T > 0, "attention: bad argument");
    if (d != 64 && d != 32) { set_error("attention: head dim %d unsupported (kernel is specialised for 64 and 32)", d); return IPDM_ERR_UNSUPPORTED; }
    // scale = 1/sqrt(sqrt(C/heads)) (Model/model.py:149); python double -> f32 scalar
    const float scale = (float)(1.0 / sqrt(sqrt((double)d)));
    const bool prof = prof_enabled();
    if (prof) prof_before(2, st);
    const bool legacy = opt(OPT_ATTN_LEGACY) != 0;
    if (d == 64 && !legacy) {
        // wave-specialised kernel, one 512-thread workgroup per CU.  64 queries per consumer wave (K/V operand reads
        // shared by two query tiles) when the 256-query workgroups come in whole rounds of the CUs, else 32
        constexpr size_t lds = (size_t)2 * (KV * KP + 64 * VP) * sizeof(float);
        if (int rc = ensure_dynamic_lds((const void *)attention_ws_kernel<64, 1>, lds)) return rc;
        if (int rc = ensure_dynamic_lds((const void *)attention_ws_kernel<64, 2>, lds)) return rc;
        if (int rc = ensure_dynamic_lds((const void *)attention_ws_kernel<64, 1, true>, lds)) return rc;
        const long wg2 = (long)cdiv(T, 256) * B * heads, wg1 = (long)cdiv(T, 128) * B * heads;
        const auto eff = [](long wg) { return (double)wg / (double)(((wg + 255) / 256) * 256); };    // round quantisation
        const int Z = scratch ? attention_kv_split(B, heads, d, T) : 1;
        const bool q2 = Z == 1 && wg2 >= 256 && eff(wg2) >= eff(wg1) - 0.03;     // (the key-slice form exists for 32-query waves)
        // a sliced layer whose query workgroups fill the chip anyway (8 slices per GPU) walks its key slices inside the
        // workgroup: same arithmetic as the split grid + combine pass, no partial outputs in memory
        const bool no_seq = opt(OPT_ATTN_NO_ZSEQ) != 0;
        const bool seq = Z >= 4 && !no_seq && wg1 >= 192;     // (2 slices: the split grid + combine pass measured 3 % faster)
        dim3 grid(cdiv(T, q2 ? 256 : 128), B * heads, seq ? 1 : Z);
        if (q2) hipLaunchKernelGGL((attention_ws_kernel<64, 2>), grid, dim3(512), lds, st, qkv, out, heads, T, scale, 1, (float *)nullptr, 1);
        else if (seq) hipLaunchKernelGGL((attention_ws_kernel<64, 1, true>), grid, dim3(512), lds, st, qkv, out, heads, T, scale, 1, (float *)nullptr, Z);
        else hipLaunchKernelGGL((attention_ws_kernel<64, 1>), grid, dim3(512), lds, st, qkv, out, heads, T, scale, Z, scratch, 1);
        if (Z > 1 && !seq)
            hipLaunchKernelGGL((attention_combine_kernel<64>), dim3(cdiv(T, 256), B * heads, 4), dim3(256), 0, st, scratch, out, T, Z, cdiv(T, KV));
    } else if (d == 64) {
        const long wg2 = (long)cdiv(T, 256) * B * heads;
        const bool q2 = wg2 >= 512 && (wg2 % 512 == 0 || wg2 >= 4 * 512);
        dim3 grid(cdiv(T, q2 ? 256 : 128), B * heads);
        if (q2) hipLaunchKernelGGL((attention_kernel<64, 2>), grid, dim3(256), 0, st, qkv, out, heads, T, scale);
        else hipLaunchKernelGGL((attention_kernel<64, 1>), grid, dim3(256), 0, st, qkv, out, heads, T, scale);
    } else {
        dim3 grid(cdiv(T, 128), B * heads);
        hipLaunchKernelGGL((attention_kernel<32, 1>), grid, dim3(256), 0, st, qkv, out, heads, T, scale);
    }
    if (prof) prof_after(2, 4.0 * B * heads * (double)T * T * d, st);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

}  // namespace ipdm

extern "C" int32_t ipdm_attention_kernel_code(int32_t d)
{
    return (d == 64 && !ipdm::opt(ipdm::OPT_ATTN_LEGACY)) ? 1 : 0;
}

extern "C" int ipdm_op_attention(const float *d_qkv, float *d_out, int32_t B, int32_t heads, int32_t d, int32_t T,
                                 void *stream)
{
    // test helper: allocates the split mode's scratch itself (the UNet executor takes it from its workspace)
    float *scratch = nullptr;
    if (ipdm::attention_scratch_floats(B, heads, d, T)) {
        IPDM_HIP_CHECK(hipMalloc((void **)&scratch, ipdm::attention_scratch_floats(B, heads, d, T) * sizeof(float)));
    }
    const int rc = ipdm::attention_launch(d_qkv, d_out, B, heads, d, T, (hipStream_t)stream, scratch);
    if (scratch) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipFree(scratch);
    }
    return rc;
}
