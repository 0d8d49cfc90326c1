// GroupNorm statistics for gfx950 (norm_layer, Model/model.py:82-90; nn.GroupNorm eps=1e-5, affine).
// Only the statistics are a kernel of their own: the normalisation (+SiLU) itself is applied by the
// consuming convolution while it stages its input tile (conv.hip), so the normalised tensor never
// exists in HBM.  HBM-bound: float4 loads, fp64 accumulation, wave64 shuffles, fixed-order
// two-level reduction (deterministic, per sample).
#include "unet_kernels.h"

using namespace ipdm;

namespace {
constexpr long GN_ONE_LAUNCH_MAX = 32768;      // float2 partial sums per group and sample a single block may fold

__device__ __forceinline__ void acc4(const float4 v, double &sum, double &sq)
{
    sum += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    sq += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
}

__global__ void __launch_bounds__(256) gn_partial_kernel(GnArgs a)
{
    const int s = blockIdx.x, g = blockIdx.y, n = blockIdx.z;
    const int Ctot = a.C1 + a.C2;
    const int cpg = Ctot / a.groups;
    double sum = 0.0, sq = 0.0;
    for (int cc = 0; cc < cpg; ++cc) {
        const int c = g * cpg + cc;
        const float *src = (c < a.C1) ? a.x1 + ((size_t)n * a.C1 + c) * a.HW : a.x2 + ((size_t)n * a.C2 + (c - a.C1)) * a.HW;
        const long nv = ((a.HW & 3) == 0 && ((size_t)src & 15) == 0) ? a.HW / 4 : 0;
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        const long step = (long)a.split * 256;
        long i = (long)s * 256 + threadIdx.x;
        // four independent 16-byte loads in flight per lane (one alone leaves the kernel latency-bound at ~2 TB/s)
        for (; i + 3 * step < nv; i += 4 * step) {
            const float4 v0 = src4[i], v1 = src4[i + step], v2 = src4[i + 2 * step], v3 = src4[i + 3 * step];
            acc4(v0, sum, sq);
            acc4(v1, sum, sq);
            acc4(v2, sum, sq);
            acc4(v3, sum, sq);
        }
        for (; i < nv; i += step) acc4(src4[i], sum, sq);
        for (long i = nv * 4 + (long)s * 256 + threadIdx.x; i < a.HW; i += (long)a.split * 256) {
            float v = src[i];
            sum += v;
            sq += (double)v * v;
        }
    }
    __shared__ double red[2][4];
    sum = wave_sum(sum);
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sum; red[1][threadIdx.x >> 6] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *p = a.partials + (((size_t)n * a.groups + g) * GN_SPLIT + s) * 2;
        p[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        p[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ void __launch_bounds__(64) gn_finalize_kernel(GnArgs a)
{
    const int g = blockIdx.x, n = blockIdx.y;
    const int Ctot = a.C1 + a.C2;
    const int cpg = Ctot / a.groups;
    const double *p = a.partials + ((size_t)n * a.groups + g) * GN_SPLIT * 2;
    double s = threadIdx.x < a.split ? p[threadIdx.x * 2] : 0.0;
    double q = threadIdx.x < a.split ? p[threadIdx.x * 2 + 1] : 0.0;
    s = wave_sum(s);
    q = wave_sum(q);
    s = __shfl(s, 0, 64);
    q = __shfl(q, 0, 64);
    const double cnt = (double)cpg * (double)a.HW;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;       // biased variance (GroupNorm)
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float meanf = (float)mean;
    for (int cc = threadIdx.x; cc < cpg; cc += 64) {
        const int c = g * cpg + cc;
        const float sc = rstd * a.gamma[c];
        a.scale[(size_t)n * Ctot + c] = sc;
        a.shift[(size_t)n * Ctot + c] = a.beta[c] - meanf * sc;
    }
}

// Stage A of the fused path: the convolutions that produced the tensor(s) left per-tile partial sums
// [B][rows][C][2] (float32); block (s, n) folds its share of the rows of both sources into float64 per-group partials in
// the layout gn_finalize_kernel reads.  A thread always owns the same channel (row-major [row][channel] blocks of
// 256 / Cp rows), so every sum is formed in a fixed order: results are bit-reproducible.
__global__ void __launch_bounds__(256) gn_tile_reduce_kernel(GnTileArgs a, int split)
{
    const int s = blockIdx.x, n = blockIdx.y;
    const int Ctot = a.src[0].C + (a.nsrc > 1 ? a.src[1].C : 0);
    const int cpg = Ctot / a.groups;
    __shared__ double lsum[256], lsq[256];
    __shared__ double gsum[512], gsq[512];          // per group (groups <= 512: gn_tiles_launch checks)
    for (int g = threadIdx.x; g < a.groups; g += 256) { gsum[g] = 0.0; gsq[g] = 0.0; }
    int cbase = 0;
    for (int k = 0; k < a.nsrc; ++k) {
        const GnTileSrc src = a.src[k];
        const int per = (src.rows + split - 1) / split;
        const int r0 = s * per, r1 = min(src.rows, r0 + per);
        for (int c0 = 0; c0 < src.C; c0 += 256) {               // channel chunks of <= 256
            const int cw = min(256, src.C - c0);
            int cp = 1;
            while (cp < cw) cp <<= 1;                             // channels of the chunk, rounded up to a power of two
            const int rpi = 256 / cp;                             // rows per iteration
            const int c = threadIdx.x % cp, ro = threadIdx.x / cp;
            double sum = 0.0, sq = 0.0;
            if (c < cw) {
                const float2 *p = reinterpret_cast<const float2 *>(src.stats) + ((size_t)n * src.rows) * src.C + c0 + c;
                int r = r0 + ro;
                for (; r + 3 * rpi < r1; r += 4 * rpi) {          // four independent loads in flight
                    const float2 v0 = p[(size_t)r * src.C], v1 = p[(size_t)(r + rpi) * src.C],
                                 v2 = p[(size_t)(r + 2 * rpi) * src.C], v3 = p[(size_t)(r + 3 * rpi) * src.C];
                    sum += (double)v0.x + (double)v1.x + (double)v2.x + (double)v3.x;
                    sq += (double)v0.y + (double)v1.y + (double)v2.y + (double)v3.y;
                }
                for (; r < r1; r += rpi) {
                    const float2 v = p[(size_t)r * src.C];
                    sum += (double)v.x;
                    sq += (double)v.y;
                }
            }
            __syncthreads();                                      // previous chunk's lsum consumed
            lsum[threadIdx.x] = sum;
            lsq[threadIdx.x] = sq;
            __syncthreads();
            // channel totals of the chunk -> their groups, in a fixed order: thread c folds its channel's row offsets
            if (threadIdx.x < cw) {
                double cs = 0.0, cq = 0.0;
                for (int j = 0; j < rpi; ++j) { cs += lsum[j * cp + threadIdx.x]; cq += lsq[j * cp + threadIdx.x]; }
                lsum[threadIdx.x] = cs;                           // (own slot j = 0 only: no other thread reads it before the barrier)
                lsq[threadIdx.x] = cq;
            }
            __syncthreads();
            // one thread per group touched by this chunk adds the chunk's channels of its group, ascending
            const int g_lo = (cbase + c0) / cpg, g_hi = (cbase + c0 + cw - 1) / cpg;
            for (int g = g_lo + threadIdx.x; g <= g_hi; g += 256) {
                const int ca = max(g * cpg, cbase + c0) - (cbase + c0), cb = min((g + 1) * cpg, cbase + c0 + cw) - (cbase + c0);
                double cs = gsum[g], cq = gsq[g];
                for (int cc = ca; cc < cb; ++cc) { cs += lsum[cc]; cq += lsq[cc]; }
                gsum[g] = cs;
                gsq[g] = cq;
            }
        }
        cbase += src.C;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < a.groups; g += 256) {
        double *p = a.partials + (((size_t)n * a.groups + g) * GN_SPLIT + s) * 2;
        p[0] = gsum[g];
        p[1] = gsq[g];
    }
}

// One launch for the common sizes: block (group, sample) folds ALL rows of its group's channels (a thread walks rows
// tid / cpg', tid / cpg' + 256 / cpg', ... of channel tid % cpg' in float64: a fixed order), the block adds the 256 partial
// sums in a fixed tree and writes the affine scale / shift of its channels itself -- no partials in memory, no second
// kernel.  (The two-stage form above keeps the tensors with very many rows: one block per group would stream them alone.)
__global__ void __launch_bounds__(256) gn_group_kernel(GnTileArgs a)
{
    const int g = blockIdx.x, n = blockIdx.y;
    const int C0 = a.src[0].C, Ctot = C0 + (a.nsrc > 1 ? a.src[1].C : 0);
    const int cpg = Ctot / a.groups;
    int cp = 1;
    while (cp < cpg) cp <<= 1;                        // channels of the group, rounded up to a power of two (<= 256)
    const int cc = threadIdx.x % cp, ro = threadIdx.x / cp, rpi = 256 / cp;
    double sum = 0.0, sq = 0.0;
    if (cc < cpg) {
        const int c = g * cpg + cc;                   // a group may straddle the two sources of a concat
        const GnTileSrc src = c < C0 ? a.src[0] : a.src[1];
        const int cl = c < C0 ? c : c - C0;
        const float2 *p = reinterpret_cast<const float2 *>(src.stats) + (size_t)n * src.rows * src.C + cl;
        int r = ro;
        // sixteen independent loads in flight (round 5): a block walks its rows alone -- 64 per thread on the 512x512 level --
        // and the walk is a chain of memory latencies, 7 us per launch for a lone slice and 70 launches per forward.  The sums
        // are formed in groups of four exactly as the four-wide loop below forms them: the same bits.
        for (; r + 15 * rpi < src.rows; r += 16 * rpi) {
            float2 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = p[(size_t)(r + j * rpi) * src.C];
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                sum += (double)v[j].x + (double)v[j + 1].x + (double)v[j + 2].x + (double)v[j + 3].x;
                sq += (double)v[j].y + (double)v[j + 1].y + (double)v[j + 2].y + (double)v[j + 3].y;
            }
        }
        for (; r + 3 * rpi < src.rows; r += 4 * rpi) {            // four independent loads in flight
            const float2 v0 = p[(size_t)r * src.C], v1 = p[(size_t)(r + rpi) * src.C],
                         v2 = p[(size_t)(r + 2 * rpi) * src.C], v3 = p[(size_t)(r + 3 * rpi) * src.C];
            sum += (double)v0.x + (double)v1.x + (double)v2.x + (double)v3.x;
            sq += (double)v0.y + (double)v1.y + (double)v2.y + (double)v3.y;
        }
        for (; r < src.rows; r += rpi) {
            const float2 v = p[(size_t)r * src.C];
            sum += (double)v.x;
            sq += (double)v.y;
        }
    }
    __shared__ double red[2][4];
    sum = wave_sum(sum);
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sum; red[1][threadIdx.x >> 6] = sq; }
    __syncthreads();
    const double s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double cnt = (double)cpg * (double)a.HW;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;               // biased variance (GroupNorm)
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float meanf = (float)mean;
    for (int k = threadIdx.x; k < cpg; k += 256) {
        const int c = g * cpg + k;
        const float sc = rstd * a.gamma[c];
        a.scale[(size_t)n * Ctot + c] = sc;
        a.shift[(size_t)n * Ctot + c] = a.beta[c] - meanf * sc;
    }
}

}  // namespace

namespace ipdm {

int gn_tiles_launch(const GnTileArgs &a, hipStream_t st)
{
    IPDM_REQUIRE(a.nsrc >= 1 && a.nsrc <= 2 && a.src[0].stats && (a.nsrc == 1 || a.src[1].stats) && a.gamma && a.beta &&
                     a.partials && a.scale && a.shift, "gn_tiles: null argument");
    const int Ctot = a.src[0].C + (a.nsrc > 1 ? a.src[1].C : 0);
    IPDM_REQUIRE(a.groups > 0 && a.groups <= 512 && Ctot % a.groups == 0, "gn_tiles: %d channels not divisible by %d groups",
                 Ctot, a.groups);
    int rows = a.src[0].rows;
    if (a.nsrc > 1 && a.src[1].rows > rows) rows = a.src[1].rows;
    const bool two_stage = opt(OPT_GN_TWO_STAGE) != 0;      // A/B: always the two-launch form
    const int cpg = Ctot / a.groups;
    if (!two_stage && cpg <= 256 && (long)rows * cpg <= GN_ONE_LAUNCH_MAX) {
        hipLaunchKernelGGL(gn_group_kernel, dim3(a.groups, a.B), dim3(256), 0, st, a);
        IPDM_LAUNCH_CHECK();
        return IPDM_OK;
    }
    int split = (rows + 63) / 64;                 // >= 64 rows per block
    split = split < 1 ? 1 : (split > GN_SPLIT ? GN_SPLIT : split);
    hipLaunchKernelGGL(gn_tile_reduce_kernel, dim3(split, a.B), dim3(256), 0, st, a, split);
    GnArgs f;
    f.x1 = f.x2 = nullptr;
    f.C1 = Ctot; f.C2 = 0; f.B = a.B; f.HW = a.HW; f.groups = a.groups; f.gamma = a.gamma; f.beta = a.beta; f.eps = a.eps;
    f.partials = a.partials; f.scale = a.scale; f.shift = a.shift; f.split = split;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.groups, a.B), dim3(64), 0, st, f);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

size_t gn_partials_bytes(int B, int groups) { return (size_t)B * groups * GN_SPLIT * 2 * sizeof(double); }

int gn_stats_launch(const GnArgs &a, hipStream_t st)
{
    IPDM_REQUIRE(a.x1 && a.gamma && a.beta && a.partials && a.scale && a.shift, "gn_stats: null argument");
    IPDM_REQUIRE(a.groups > 0 && (a.C1 + a.C2) % a.groups == 0, "gn_stats: %d channels not divisible by %d groups",
                 a.C1 + a.C2, a.groups);
    // workgroups per (sample, group): enough to stream a large group at HBM speed, but at least ~8k elements each (the
    // 63x29 / 32x32 layers would otherwise launch 8192 workgroups of two elements per thread); the partial sums are
    // combined in a fixed order for any split, so results do not depend on it beyond fp64 rounding of the partials
    GnArgs b = a;
    const long per_group = (long)((a.C1 + a.C2) / a.groups) * a.HW;
    int split = (int)((per_group + 8191) / 8192);
    b.split = split < 1 ? 1 : (split > GN_SPLIT ? GN_SPLIT : split);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(b.split, a.groups, a.B), dim3(256), 0, st, b);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.groups, a.B), dim3(64), 0, st, b);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

}  // namespace ipdm
