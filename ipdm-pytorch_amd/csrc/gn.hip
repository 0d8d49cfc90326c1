// GroupNorm statistics for gfx950 (norm_layer, Model/model.py:82-90; nn.GroupNorm eps=1e-5, affine).
// Only the statistics are a kernel of their own: the normalisation (+SiLU) itself is applied by the
// consuming convolution while it stages its input tile (conv.hip), so the normalised tensor never
// exists in HBM.  HBM-bound: float4 loads, fp64 accumulation, wave64 shuffles, fixed-order
// two-level reduction (deterministic, per sample).
#include "unet_kernels.h"

using namespace ipdm;

namespace {

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

}  // namespace

namespace ipdm {

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
