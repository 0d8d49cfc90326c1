// The "ART" domain convertor for gfx950: SART over a triangle-area lookup table, NSL0-TV steps, and the
// forward projector -- recons_torch / proj_torch of Recon/TASART2DNSL0 (TASART2DNSL0_PyAPI.cpp:33-80 ->
// TASART2DNSL0.cu).  SURVEY section 8(f) rank 3.  Written from scratch for CDNA4, not a translation of the CUDA
// launch sequence (which issues ~10 tiny kernels, 4 memsets and 2 array copies per view, 2000 views per sweep):
//   * what depends only on the view is computed once per plan: the nr+1 bin-edge rays of all views
//     (update_lines_kernel, .cu:270-302) and the normalisation projection A*1 (.cu:871);
//   * ONE launch per view.  A thread owns one pixel for up to 8 slices: it back-projects the previous view's
//     correction from the footprint it stored (lut_bp_kernel x2 + update_kernel, .cu:397-481, fused, the slice-
//     independent normaliser computed once), then forms the footprint of the current view (lut_init_foot_kernel,
//     .cu:304-341) and forward-projects its updated value (lut_fp_kernel, .cu:343-383);
//   * the projector's scatter goes through a 64-bin LDS window per 16x16 pixel tile in 2^-44 fixed point
//     (int64 adds are associative, so the sums are order-independent and the reconstruction is bit-reproducible;
//     the CUDA code scatters float atomicAdds), flushed with one global atomic per touched bin;
//   * the correction of a view (apply_geodiv_kernel + correction_kernel, .cu:385-460) is formed by the NEXT launch,
//     once per workgroup, for the 64-bin window its tile back-projects from; three bin buffers rotate (read / add /
//     being zeroed): no memsets, no extra launch, no device-wide fence (a ticket-counter "last workgroup" tail was
//     measured at 100+ us per view: 1024 agent-scope release fences);
//   * all scalars of the outer loop (dp, dg, alpha, the TV step) live on the device: no host synchronisation.
// Textures: point-filtered ones are clamped indexed loads; the linearly filtered area table (.cu:262) is exact float
// bilinear interpolation (the 8-bit weights of the CUDA texture unit are not modelled).
#include <cmath>
#include <cstdlib>
#include <vector>
#include "common.h"

using namespace ipdm;

// plain IEEE operations, no mul+add contraction: the geometry has cancellations (pos = L.y*x + L.z*y + L.w with
// |L.w| ~ 60 cm against sub-millimetre results) whose rounding would otherwise depend on the compiler's fusion choices
#pragma clang fp contract(off)

namespace {

constexpr int NFOOT = 5;        // .cu:729
constexpr int BMAX = 8;         // slices one thread carries
constexpr int WIN = 64;         // LDS bin window per tile (a 16x16 tile spans <= 53 bins in the reference geometry)
constexpr double FIX = 17592186044416.0;          // 2^44
constexpr double UNFIX = 1.0 / 17592186044416.0;

struct ArtView {
    float src_x, src_y, uvs_x, uvs_y, uvt_x, uvt_y, beta, pad;
};

struct ArtConst {
    int nx, ny, nr, dimx, dimy;
    float dx, dy, offx, offy, dso, dsd, dr, offr, xx, yy, rr, ox, oy, vox, geodiv;
};

// update_lines_kernel, .cu:270-302, for every view at once
__global__ void art_lines_kernel(ArtConst c, const ArtView *__restrict__ views, float4 *__restrict__ lines, int na)
{
    const int is = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y;
    if (is > c.nr || v >= na) return;
    const ArtView w = views[v];
    const float s0 = -c.rr + c.offr * c.dr;
    const float gamma = s0 + is * c.dr;
    const float p1x = w.src_x + c.dsd * sinf(w.beta + gamma);
    const float p1y = w.src_y + -c.dsd * cosf(w.beta + gamma);
    const float rx = p1x - w.src_x, ry = p1y - w.src_y;
    float ang = atan2f(ry, rx) * (360.0f / (2.0f * (float)M_PI));
    if (ang < 0.0f) ang += 360.0f;
    const float A = p1y - w.src_y;
    const float B = w.src_x - p1x;
    const float C = p1x * w.src_y - w.src_x * p1y;
    const float Z = sqrtf(A * A + B * B);
    if (ang <= 45.0f) { ; }
    else if (ang <= 90.0f) ang = 90.0f - ang;
    else if (ang <= 135.0f) ang = ang - 90.0f;
    else if (ang <= 180.0f) ang = 180.0f - ang;
    else if (ang <= 225.0f) ang = ang - 180.0f;
    else if (ang <= 270.0f) ang = 270.0f - ang;
    else if (ang <= 315.0f) ang = ang - 270.0f;
    else ang = 360.0f - ang;
    lines[(size_t)v * (c.nr + 1) + is] = make_float4(ang, A / Z, B / Z, C / Z);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// tex2D(areaTex, x, y): unnormalised, linear filter, clamp (.cu:580-598)
__device__ __forceinline__ float lut_fetch(const ArtConst &c, const float *__restrict__ lut, float x, float y)
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx0 = floorf(xb), fy0 = floorf(yb);
    const float ax = xb - fx0, ay = yb - fy0;
    const int x0 = clampi((int)fx0, 0, c.dimx - 1), x1 = clampi((int)fx0 + 1, 0, c.dimx - 1);
    const int y0 = clampi((int)fy0, 0, c.dimy - 1), y1 = clampi((int)fy0 + 1, 0, c.dimy - 1);
    const float t00 = lut[y0 * c.dimx + x0], t10 = lut[y0 * c.dimx + x1];
    const float t01 = lut[y1 * c.dimx + x0], t11 = lut[y1 * c.dimx + x1];
    return (1.0f - ay) * ((1.0f - ax) * t00 + ax * t10) + ay * ((1.0f - ax) * t01 + ax * t11);
}

// fetchAreaLut, .cu:253-268
__device__ __forceinline__ float fetch_area(const ArtConst &c, const float *__restrict__ lut,
                                            const float4 *__restrict__ lines, int sidx, float x, float y)
{
    const float4 L = lines[clampi(sidx, 0, c.nr)];
    const float pos = L.y * x + L.z * y + L.w;
    const float value = lut_fetch(c, lut, fabsf(pos) * c.ox + 0.5f, L.x * c.oy + 0.5f);
    return pos < 0.0f ? c.vox - value : value;
}

// lut_init_foot_kernel, .cu:304-341
__device__ __forceinline__ void footprint(const ArtConst &c, const ArtView &w, const float *__restrict__ lut,
                                          const float4 *__restrict__ lines, int ix, int iy, float &dist, int &sb,
                                          float (&foot)[NFOOT])
{
    const float x = (float)((ix + 0.5) * c.dx - c.xx * 1.0f + c.offx);
    const float y = (float)((iy + 0.5) * c.dy - c.yy * 1.0f + c.offy);
    dist = sqrtf((x - w.src_x) * (x - w.src_x) + (y - w.src_y) * (y - w.src_y));
    const float ds = w.uvs_x * x + w.uvs_y * y;
    const float dt = w.uvt_x * x + w.uvt_y * y;
    const float gamma = atanf(ds / (dt + c.dso));
    sb = (int)floorf(gamma / c.dr + 0.5f * (c.nr - 1) - c.offr) - NFOOT / 2;
    int is = sb;
    float a0 = fetch_area(c, lut, lines, is, x, y);
    ++is;
#pragma unroll
    for (int k = 0; k < NFOOT; ++k, ++is) {
        const float a1 = fetch_area(c, lut, lines, is, x, y);
        foot[k] = fabsf(a0 - a1);
        a0 = a1;
    }
}

struct SweepArgs {
    ArtConst c;
    const float *lut;
    const float4 *lines;        // [na][nr+1]
    const ArtView *views;       // [na]
    const float *norm;          // [na][nr]   A*1 scaled by geodiv
    const float *proj;          // [B][proj_stride] measured data
    long proj_stride;
    float *vol;                 // [B][ny*nx]
    float4 *foot;               // [ny*nx][2]  (dist, sb, f0, f1) (f2, f3, f4, -)
    const unsigned long long *bins_prev;    // [B][nr] finished fixed-point forward projection of v_prev
    unsigned long long *bins_cur;           // [B][nr] accumulated by this launch (zero on entry)
    unsigned long long *bins_next;          // [B][nr] zeroed by this launch for the next one
    int v_prev, v_cur, B;
    float lamda;
};

// apply_geodiv_kernel + correction_kernel (.cu:385-460) for one bin of the previous view
__device__ __forceinline__ float correction(const SweepArgs &a, int b, int r)
{
    const long long s = (long long)a.bins_prev[b * a.c.nr + r];
    const float p = (float)((double)s * UNFIX) * a.c.geodiv;
    const float n = a.norm[(size_t)a.v_prev * a.c.nr + r];
    const float m = a.proj[b * a.proj_stride + (long)a.v_prev * a.c.nr + r];
    return n > 0.0f ? a.c.geodiv * ((m - p) / n) : 0.0f;
}

// One SART view: back-project + update for v_prev (if >= 0), footprint + forward projection for v_cur (if >= 0).
template <int TS>
__global__ void __launch_bounds__(TS * TS) art_sweep_kernel(SweepArgs a)
{
    __shared__ unsigned long long win[BMAX * WIN];
    __shared__ float lcorr[BMAX * WIN];
    __shared__ int wmin_s, wminp_s;
    const ArtConst &c = a.c;
    constexpr int NT = TS * TS;
    const int tid = threadIdx.y * TS + threadIdx.x;
    const int ix = blockIdx.x * TS + threadIdx.x, iy = blockIdx.y * TS + threadIdx.y;
    const bool inside = ix < c.nx && iy < c.ny;
    const int pix = iy * c.nx + ix;
    const long np = (long)c.nx * c.ny;
    // the bins of the launch after this one
    for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * NT + tid; i < (long)a.B * c.nr;
         i += (long)gridDim.x * gridDim.y * NT)
        a.bins_next[i] = 0ULL;
    if (tid == 0) { wmin_s = 0x7fffffff; wminp_s = 0x7fffffff; }
    for (int i = tid; i < BMAX * WIN; i += NT) win[i] = 0ULL;
    float vol[BMAX];
#pragma unroll
    for (int b = 0; b < BMAX; ++b) vol[b] = (inside && b < a.B) ? a.vol[b * np + pix] : 0.0f;
    __syncthreads();

    if (a.v_prev >= 0) {
        float4 f0 = make_float4(1.0f, 0.0f, 0.0f, 0.0f), f1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (inside) {
            f0 = a.foot[pix * 2];
            f1 = a.foot[pix * 2 + 1];
            atomicMin(&wminp_s, __float_as_int(f0.y));
        }
        __syncthreads();
        // the tile's window of corrected bins, formed once per workgroup
        const int w0 = wminp_s < 0 ? 0 : wminp_s;
        for (int i = tid; i < a.B * WIN; i += NT) {
            const int r = w0 + (i % WIN);
            lcorr[i] = r < c.nr ? correction(a, i / WIN, r) : 0.0f;
        }
        __syncthreads();
        if (inside) {
            const float foot[NFOOT] = {f0.z, f0.w, f1.x, f1.y, f1.z};
            const int sb = __float_as_int(f0.y);
            const float div = 1.0f / f0.x;
            int idx[NFOOT];
            float nb = 0.0f;
#pragma unroll
            for (int k = 0; k < NFOOT; ++k) {
                idx[k] = clampi(sb + k, 0, c.nr - 1);       // clamp-addressed point fetch (CortexRef / geodivTex)
                nb += c.geodiv * div * foot[k];
            }
#pragma unroll
            for (int b = 0; b < BMAX; ++b) {
                if (b < a.B) {
                    float bp = 0.0f;
#pragma unroll
                    for (int k = 0; k < NFOOT; ++k) {
                        const int j = idx[k] - w0;
                        const float cv = (j >= 0 && j < WIN) ? lcorr[b * WIN + j] : correction(a, b, idx[k]);
                        bp += cv * div * foot[k];
                    }
                    const float upd = nb > 0.0f ? a.lamda * (bp / nb) : 0.0f;
                    vol[b] = fmaxf(vol[b] + upd, 0.0f);
                    a.vol[b * np + pix] = vol[b];
                }
            }
        }
    }
    if (a.v_cur < 0) return;

    // ---- footprint of the current view, forward projection through the LDS window
    float dist = 1.0f, foot[NFOOT] = {0, 0, 0, 0, 0};
    int sb = 0;
    if (inside) {
        footprint(c, a.views[a.v_cur], a.lut, a.lines + (size_t)a.v_cur * (c.nr + 1), ix, iy, dist, sb, foot);
        a.foot[pix * 2] = make_float4(dist, __int_as_float(sb), foot[0], foot[1]);
        a.foot[pix * 2 + 1] = make_float4(foot[2], foot[3], foot[4], 0.0f);
        atomicMin(&wmin_s, sb);
    }
    __syncthreads();
    const int wmin = wmin_s;
    if (inside) {
#pragma unroll
        for (int b = 0; b < BMAX; ++b) {
            if (b < a.B && vol[b] != 0.0f) {
                const float dv = vol[b] / dist;
#pragma unroll
                for (int k = 0; k < NFOOT; ++k) {
                    const int is = sb + k;
                    if (is < 0 || is >= c.nr || foot[k] <= 0.0f || dv == 0.0f) continue;
                    const unsigned long long q = (unsigned long long)__double2ll_rn((double)(dv * foot[k]) * FIX);
                    const int j = is - wmin;
                    if (j < WIN) atomicAdd(&win[b * WIN + j], q);
                    else atomicAdd(&a.bins_cur[b * c.nr + is], q);
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < a.B * WIN; i += NT) {
        const unsigned long long q = win[i];
        const int is = wmin + (i % WIN);
        if (q != 0ULL && is >= 0 && is < c.nr) atomicAdd(&a.bins_cur[(i / WIN) * c.nr + is], q);
    }
}

// ---- a whole sweep in ONE launch: the grid stays resident, a pixel's slices and footprint stay in registers
struct PersistArgs {
    ArtConst c;
    const float *lut;
    const float4 *lines;
    const ArtView *views;
    const float *norm;
    const float *proj;
    long proj_stride;
    float *vol;
    unsigned long long *bins;   // 4 x [B][nr], zero on entry; rotation: scatter v%4, zeroed two views ahead
    unsigned int *sync;         // SYNC_WORDS words of barrier state (see grid_barrier), zero on entry
    int na, B;
    float lamda;
};

__device__ __forceinline__ unsigned long long coherent_load(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Grid-wide barrier over co-resident workgroups (the launcher checks that the whole grid fits on the chip).
// Everything the workgroups exchange (the bins) moves through agent-scope atomics and coherent loads, so the barrier
// needs no cache write-back / invalidate (agent-scope release / acquire fences in every workgroup made the first
// version take 270 us per view): a wave only has to see its own atomics acknowledged before it arrives.
// Two levels, because read-modify-writes of ONE address serialise at ~50 ns each (1024 arrivals on one counter: 50 us):
// groups of 32 workgroups count on their own line, the last of a group counts on the root, the last at the root
// publishes the generation to every group's flag, and a workgroup polls only its group's flag.
// Layout of `sync` (uint32, 128-byte lines): [0] abort, [32] root, [64 + 32 g] group g counter, [64 + 32 (64 + g)] flag.
// The wait is bounded: if it ever expired the abort flag makes every workgroup leave and poison its output.
constexpr int SYNC_GROUP = 32, SYNC_MAX_GROUPS = 64, SYNC_WORDS = 64 + 32 * 2 * SYNC_MAX_GROUPS;

__device__ __forceinline__ void grid_arrive(unsigned int *sync, unsigned int gen, unsigned int wg, unsigned int nwg)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's atomics / stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const unsigned int g = wg / SYNC_GROUP, ngroups = (nwg + SYNC_GROUP - 1) / SYNC_GROUP;
        const unsigned int gsize = min((unsigned)SYNC_GROUP, nwg - g * SYNC_GROUP);
        if (__hip_atomic_fetch_add(sync + 64 + 32 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == gen * gsize) {
            if (__hip_atomic_fetch_add(sync + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == gen * ngroups) {
                for (unsigned int i = 0; i < ngroups; ++i)
                    __hip_atomic_store(sync + 64 + 32 * (SYNC_MAX_GROUPS + i), gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__device__ __forceinline__ bool grid_wait(unsigned int *sync, unsigned int gen, unsigned int wg)
{
    __shared__ int ok_s;
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const unsigned int *flag = sync + 64 + 32 * (SYNC_MAX_GROUPS + wg / SYNC_GROUP);
        int ok = 1;
        long spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1L << 22)) {
                __hip_atomic_store(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            if ((spins & 1023) == 0 && __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                ok = 0;
                break;
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

__global__ void __launch_bounds__(256) art_sweep_persistent_kernel(PersistArgs a)
{
    __shared__ unsigned long long win[BMAX * WIN];
    __shared__ float lcorr[BMAX * WIN];
    __shared__ int wmin_s;
    const ArtConst &c = a.c;
    const int tid = threadIdx.y * 16 + threadIdx.x;
    const int ix = blockIdx.x * 16 + threadIdx.x, iy = blockIdx.y * 16 + threadIdx.y;
    const bool inside = ix < c.nx && iy < c.ny;
    const int pix = iy * c.nx + ix;
    const long np = (long)c.nx * c.ny;
    const unsigned int nwg = gridDim.x * gridDim.y;
    const long wg = (long)blockIdx.y * gridDim.x + blockIdx.x;
    const size_t bin_n = (size_t)a.B * c.nr;
    float vol[BMAX];
#pragma unroll
    for (int b = 0; b < BMAX; ++b) vol[b] = (inside && b < a.B) ? a.vol[b * np + pix] : 0.0f;

    // the footprint of a view depends on nothing the sweep computes: the one of view v+1 is formed while the grid
    // barrier of view v collects its arrivals
    float dist = 1.0f, foot[NFOOT] = {0, 0, 0, 0, 0};
    int sb = 0;
    if (inside) footprint(c, a.views[0], a.lut, a.lines, ix, iy, dist, sb, foot);
    for (int v = 0; v < a.na; ++v) {
        unsigned long long *bins_cur = a.bins + (size_t)(v & 3) * bin_n;
        unsigned long long *bins_z = a.bins + (size_t)((v + 2) & 3) * bin_n;
        for (long i = wg * 256 + tid; i < (long)bin_n; i += (long)nwg * 256)
            __hip_atomic_store(&bins_z[i], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) wmin_s = 0x7fffffff;
        for (int i = tid; i < BMAX * WIN; i += 256) win[i] = 0ULL;
        __syncthreads();
        if (inside) atomicMin(&wmin_s, sb);
        __syncthreads();
        const int wmin = wmin_s;
        if (inside) {
#pragma unroll
            for (int b = 0; b < BMAX; ++b) {
                if (b < a.B && vol[b] != 0.0f) {
                    const float dv = vol[b] / dist;
#pragma unroll
                    for (int k = 0; k < NFOOT; ++k) {
                        const int is = sb + k;
                        if (is < 0 || is >= c.nr || foot[k] <= 0.0f || dv == 0.0f) continue;
                        const unsigned long long q = (unsigned long long)__double2ll_rn((double)(dv * foot[k]) * FIX);
                        const int j = is - wmin;
                        if (j < WIN) atomicAdd(&win[b * WIN + j], q);
                        else atomicAdd(&bins_cur[b * c.nr + is], q);
                    }
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < a.B * WIN; i += 256) {
            const unsigned long long q = win[i];
            const int is = wmin + (i % WIN);
            if (q != 0ULL && is >= 0 && is < c.nr) atomicAdd(&bins_cur[(i / WIN) * c.nr + is], q);
        }
        grid_arrive(a.sync, (unsigned)(v + 1), (unsigned)wg, nwg);
        float dist_n = 1.0f, foot_n[NFOOT] = {0, 0, 0, 0, 0};
        int sb_n = 0;
        if (inside && v + 1 < a.na)
            footprint(c, a.views[v + 1], a.lut, a.lines + (size_t)(v + 1) * (c.nr + 1), ix, iy, dist_n, sb_n, foot_n);
        if (!grid_wait(a.sync, (unsigned)(v + 1), (unsigned)wg)) {
            // cannot happen on a grid the launcher found co-resident; if it does, fail loudly: the volume becomes NaN
            if (inside)
                for (int b = 0; b < a.B; ++b) a.vol[b * np + pix] = __builtin_nanf("");
            return;
        }
        // ---- correction of view v for this tile's window, back-projection, update (.cu:385-481)
        const int w0 = wmin < 0 ? 0 : wmin;
        for (int i = tid; i < a.B * WIN; i += 256) {
            const int b = i / WIN, r = w0 + (i % WIN);
            float cv = 0.0f;
            if (r < c.nr) {
                const long long s = (long long)coherent_load(&bins_cur[b * c.nr + r]);
                const float p = (float)((double)s * UNFIX) * c.geodiv;
                const float n = a.norm[(size_t)v * c.nr + r];
                const float m = a.proj[b * a.proj_stride + (long)v * c.nr + r];
                cv = n > 0.0f ? c.geodiv * ((m - p) / n) : 0.0f;
            }
            lcorr[i] = cv;
        }
        __syncthreads();
        if (inside) {
            const float div = 1.0f / dist;
            int idx[NFOOT];
            float nb = 0.0f;
#pragma unroll
            for (int k = 0; k < NFOOT; ++k) {
                idx[k] = clampi(sb + k, 0, c.nr - 1);
                nb += c.geodiv * div * foot[k];
            }
#pragma unroll
            for (int b = 0; b < BMAX; ++b) {
                if (b < a.B) {
                    float bp = 0.0f;
#pragma unroll
                    for (int k = 0; k < NFOOT; ++k) {
                        const int j = idx[k] - w0;
                        float cv;
                        if (j >= 0 && j < WIN) {
                            cv = lcorr[b * WIN + j];
                        } else {            // a footprint bin outside the tile's window (never in the reference geometry)
                            const long long s = (long long)coherent_load(&bins_cur[b * c.nr + idx[k]]);
                            const float p = (float)((double)s * UNFIX) * c.geodiv;
                            const float n = a.norm[(size_t)v * c.nr + idx[k]];
                            const float m = a.proj[b * a.proj_stride + (long)v * c.nr + idx[k]];
                            cv = n > 0.0f ? c.geodiv * ((m - p) / n) : 0.0f;
                        }
                        bp += cv * div * foot[k];
                    }
                    const float upd = nb > 0.0f ? a.lamda * (bp / nb) : 0.0f;
                    vol[b] = fmaxf(vol[b] + upd, 0.0f);
                }
            }
        }
        __syncthreads();        // lcorr / win are rewritten by the next view
        dist = dist_n;
        sb = sb_n;
#pragma unroll
        for (int k = 0; k < NFOOT; ++k) foot[k] = foot_n[k];
    }
    if (inside) {
#pragma unroll
        for (int b = 0; b < BMAX; ++b)
            if (b < a.B) a.vol[b * np + pix] = vol[b];
    }
}

struct ProjArgs {
    ArtConst c;
    const float *lut;
    const float4 *lines;
    const ArtView *views;
    const float *vol;           // [B][ny*nx], or nullptr: a volume of ones (B = 1)
    unsigned long long *bins;   // [B][na][nr]
    int na, B;
};

// The forward projector alone (DoProjection, .cu:1335-1438): all views in one launch, grid.z = view.
__global__ void __launch_bounds__(256) art_project_kernel(ProjArgs a)
{
    __shared__ unsigned long long win[BMAX * WIN];
    __shared__ int wmin_s;
    const ArtConst &c = a.c;
    const int tid = threadIdx.y * 16 + threadIdx.x;
    const int ix = blockIdx.x * 16 + threadIdx.x, iy = blockIdx.y * 16 + threadIdx.y;
    const int v = blockIdx.z;
    const bool inside = ix < c.nx && iy < c.ny;
    const long np = (long)c.nx * c.ny;
    if (tid == 0) wmin_s = 0x7fffffff;
    for (int i = tid; i < BMAX * WIN; i += 256) win[i] = 0ULL;
    __syncthreads();
    float dist = 1.0f, foot[NFOOT] = {0, 0, 0, 0, 0};
    int sb = 0;
    if (inside) {
        footprint(c, a.views[v], a.lut, a.lines + (size_t)v * (c.nr + 1), ix, iy, dist, sb, foot);
        atomicMin(&wmin_s, sb);
    }
    __syncthreads();
    const int wmin = wmin_s;
    if (inside) {
        for (int b = 0; b < a.B; ++b) {
            const float att = a.vol ? a.vol[b * np + iy * c.nx + ix] : 1.0f;
            if (att == 0.0f) continue;
            const float dv = att / dist;
#pragma unroll
            for (int k = 0; k < NFOOT; ++k) {
                const int is = sb + k;
                if (is < 0 || is >= c.nr || foot[k] <= 0.0f || dv == 0.0f) continue;
                const unsigned long long q = (unsigned long long)__double2ll_rn((double)(dv * foot[k]) * FIX);
                const int j = is - wmin;
                if (j < WIN) atomicAdd(&win[b * WIN + j], q);
                else atomicAdd(&a.bins[((size_t)b * a.na + v) * c.nr + is], q);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < a.B * WIN; i += 256) {
        const unsigned long long q = win[i];
        const int is = wmin + (i % WIN);
        if (q != 0ULL && is >= 0 && is < c.nr) atomicAdd(&a.bins[((size_t)(i / WIN) * a.na + v) * c.nr + is], q);
    }
}

// bins -> projection values (apply_geodiv_kernel, .cu:385-393)
__global__ void art_unfix_kernel(const unsigned long long *__restrict__ bins, float *__restrict__ out, long n,
                                 float geodiv)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)((double)(long long)bins[i] * UNFIX) * geodiv;
}

// ---- outer-loop scalars (DoReconstruction, .cu:841-955), one set per slice, kept on the device
struct ArtState {
    float alpha, dp, dtvg, normg;
};

__global__ void art_state_init_kernel(ArtState *st, int B)
{
    const int b = threadIdx.x;
    if (b < B) st[b] = ArtState{0.1f, 0.0f, 0.0f, 0.0f};
}

// partial sums of (x - y)^2 (y may be null) in fp64: [B][nblk]
__global__ void __launch_bounds__(256) art_sq_partial_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                             long np, double *__restrict__ partial)
{
    const int b = blockIdx.y;
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < np; i += (long)gridDim.x * 256) {
        const float d = y ? (-1.0f * x[b * np + i] + y[b * np + i]) : x[b * np + i];
        s += (double)d * d;
    }
    __shared__ double red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[b * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// mode 0: dp = |x_back - x_for|, dtvg = alpha * dp (.cu:895-912);  1: normg = |grad| (.cu:927);
// 2: dg = |x_back - x_for|, alpha *= 0.96 when dg > 0.995 dp (.cu:944-949)
__global__ void art_state_update_kernel(ArtState *st, const double *__restrict__ partial, int nblk, int mode)
{
    const int b = blockIdx.x;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += partial[b * nblk + i];
    const float nrm = (float)sqrt(s);
    ArtState t = st[b];
    if (mode == 0) {
        t.dp = nrm;
        t.dtvg = t.alpha * nrm;
    } else if (mode == 1) {
        t.normg = nrm;
    } else if ((double)nrm > 0.995 * (double)t.dp) {
        t.alpha = (float)((double)t.alpha * 0.96);
    }
    st[b] = t;
}

// Grad_NSL0TV, .cu:483-540 (point-filtered, clamp-addressed volume texture)
__global__ void __launch_bounds__(256) art_tvgrad_kernel(const float *__restrict__ vol, float *__restrict__ grad,
                                                         int nx, int ny, float sigma)
{
    const int ix = blockIdx.x * 16 + threadIdx.x, iy = blockIdx.y * 16 + threadIdx.y;
    if (ix >= nx || iy >= ny) return;
    const float *v = vol + (size_t)blockIdx.z * nx * ny;
    auto tex = [&](int x, int y) { return v[clampi(y, 0, ny - 1) * nx + clampi(x, 0, nx - 1)]; };
    const float mins = 0.0001f;
    const float cc = tex(ix, iy), xp = tex(ix + 1, iy), yp = tex(ix, iy + 1), xm = tex(ix - 1, iy), ym = tex(ix, iy - 1);
    const float xmyp = tex(ix - 1, iy + 1), xpym = tex(ix + 1, iy - 1);
    const float Dxy = sqrtf(mins * mins + (cc - xp) * (cc - xp) + (cc - yp) * (cc - yp));
    const float Dxm = sqrtf(mins * mins + (xm - cc) * (xm - cc) + (xm - xmyp) * (xm - xmyp));
    const float Dym = sqrtf(mins * mins + (ym - cc) * (ym - cc) + (ym - xpym) * (ym - xpym));
    const float e1 = expf(Dxy / (2 * sigma)) + expf(-Dxy / (2 * sigma));
    const float e2 = expf(Dxm / (2 * sigma)) + expf(-Dxm / (2 * sigma));
    const float e3 = expf(Dym / (2 * sigma)) + expf(-Dym / (2 * sigma));
    const float Wxy = (2 / sigma) / (e1 * e1), Wxm = (2 / sigma) / (e2 * e2), Wym = (2 / sigma) / (e3 * e3);
    float t = 0;
    t += Wxy * (cc - xp + cc - yp) / Dxy;
    t -= Wxm * (xm - cc) / Dxm;
    t -= Wym * (ym - cc) / Dym;
    if (t < mins * mins) t = 0;
    grad[(size_t)blockIdx.z * nx * ny + iy * nx + ix] = t;
}

// nonnegative (.cu:543-558) + saxpy_fast(-dtvg / normg, grad, x_for) (.cu:929)
__global__ void art_tvstep_kernel(float *__restrict__ x, const float *__restrict__ grad, const ArtState *st, long np)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= np) return;
    const ArtState t = st[b];
    const float a = -1.0f * t.dtvg / t.normg;
    float v = x[b * np + i];
    if (v < 0) v = 0;
    x[b * np + i] = a * grad[b * np + i] + v;
}

}  // namespace

struct ipdm_art_plan {
    ipdm_art_geom g;
    ArtConst c;
    float *d_lut = nullptr, *d_norm = nullptr;
    float4 *d_lines = nullptr;
    ArtView *d_views = nullptr;
    bool persistent_ok = false;     // the 16x16-tile grid of one sweep is co-resident on this device
    unsigned int *d_abort = nullptr;    // sticky: some one-launch sweep of the running reconstruction gave up on its grid barrier
};

// after every one-launch sweep: remember an expired grid barrier (sync[0]) past the next sweep's reset of `sync`
__global__ void art_abort_latch_kernel(const unsigned int *sync, unsigned int *latch)
{
    if (sync[0]) *latch = 1u;
}

static int project_bins(ipdm_art_plan *p, const float *d_vol, unsigned long long *bins, int na, int B, hipStream_t st)
{
    IPDM_HIP_CHECK(hipMemsetAsync(bins, 0, (size_t)B * na * p->c.nr * sizeof(unsigned long long), st));
    ProjArgs a{p->c, p->d_lut, p->d_lines, p->d_views, d_vol, bins, na, B};
    hipLaunchKernelGGL(art_project_kernel, dim3(cdiv(p->c.nx, 16), cdiv(p->c.ny, 16), na), dim3(16, 16), 0, st, a);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

extern "C" int ipdm_art_plan_create(const ipdm_art_geom *g, const float *lut, const float *betas, ipdm_art_plan **out)
{
    IPDM_REQUIRE(g && lut && betas && out, "art_plan_create: null argument");
    IPDM_REQUIRE(g->nx > 0 && g->ny > 0 && g->nr > 1 && g->na > 0 && g->ta_dimx > 1 && g->ta_dimy > 1 && g->dr > 0 &&
                     g->ta_deltax > 0 && g->ta_deltay > 0, "art_plan_create: bad geometry");
    ipdm_art_plan *p = new ipdm_art_plan();
    p->g = *g;
    ArtConst &c = p->c;
    c.nx = g->nx; c.ny = g->ny; c.nr = g->nr; c.dimx = g->ta_dimx; c.dimy = g->ta_dimy;
    c.dx = g->dx; c.dy = g->dy; c.offx = g->offset_x; c.offy = g->offset_y; c.dso = g->dso; c.dsd = g->dsd;
    c.dr = g->dr; c.offr = g->offset_r;
    c.xx = g->nx * g->dx * 0.5f;                   // .cu:724-726
    c.yy = g->ny * g->dy * 0.5f;
    c.rr = g->nr * g->dr * 0.5f;
    c.ox = 1.0f / g->ta_deltax;                    // BindAreaLut, .cu:594-597
    c.oy = 1.0f / g->ta_deltay;
    c.vox = fabsf(g->dx * g->dy);
    c.geodiv = 1.0f / g->dr;                       // _cmpGeoDiv, .cu:601-611
    std::vector<ArtView> views(g->na);
    for (int ia = 0; ia < g->na; ++ia) {           // .cu:853-861, rotateCCW_z :147-150
        const float beta = (betas[ia] - g->angle_start) * ((float)M_PI / 180.0f);
        const float cs = cosf(beta), sn = sinf(beta);
        ArtView &w = views[ia];
        w.beta = beta;
        w.uvt_x = 0.0f * cs - (-1.0f) * sn;
        w.uvt_y = 0.0f * sn + (-1.0f) * cs;
        w.uvs_x = 1.0f * cs - 0.0f * sn;
        w.uvs_y = 1.0f * sn + 0.0f * cs;
        w.src_x = 0.0f * cs - g->dso * sn;
        w.src_y = 0.0f * sn + g->dso * cs;
        w.pad = 0.0f;
    }
    const size_t lut_n = (size_t)g->ta_dimx * g->ta_dimy, norm_n = (size_t)g->na * g->nr;
    IPDM_HIP_CHECK(hipMalloc((void **)&p->d_lut, lut_n * sizeof(float)));
    IPDM_HIP_CHECK(hipMalloc((void **)&p->d_views, views.size() * sizeof(ArtView)));
    IPDM_HIP_CHECK(hipMalloc((void **)&p->d_lines, (size_t)g->na * (g->nr + 1) * sizeof(float4)));
    IPDM_HIP_CHECK(hipMalloc((void **)&p->d_norm, norm_n * sizeof(float)));
    IPDM_HIP_CHECK(hipMalloc((void **)&p->d_abort, sizeof(unsigned int)));
    IPDM_HIP_CHECK(hipMemcpy(p->d_lut, lut, lut_n * sizeof(float), hipMemcpyHostToDevice));
    IPDM_HIP_CHECK(hipMemcpy(p->d_views, views.data(), views.size() * sizeof(ArtView), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(art_lines_kernel, dim3(cdiv(g->nr + 1, 256), g->na), dim3(256), 0, nullptr, c, p->d_views,
                       p->d_lines, g->na);
    IPDM_LAUNCH_CHECK();
    unsigned long long *bins = nullptr;
    IPDM_HIP_CHECK(hipMalloc((void **)&bins, norm_n * sizeof(unsigned long long)));
    int rc = project_bins(p, nullptr, bins, g->na, 1, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(art_unfix_kernel, dim3(cdiv((long)norm_n, 256)), dim3(256), 0, nullptr, bins, p->d_norm,
                       (long)norm_n, c.geodiv);
    IPDM_LAUNCH_CHECK();
    IPDM_HIP_CHECK(hipDeviceSynchronize());
    (void)hipFree(bins);
    {   // can the one-launch sweep hold its whole grid on the chip? (IPDM_ART_PER_VIEW=1 forces one launch per view)
        int per_cu = 0, dev = 0, cus = 0;
        IPDM_HIP_CHECK(hipGetDevice(&dev));
        IPDM_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        IPDM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, art_sweep_persistent_kernel, 256, 0));
        const long tiles = (long)cdiv(g->nx, 16) * cdiv(g->ny, 16);
        p->persistent_ok = !ipdm::opt(ipdm::OPT_ART_PER_VIEW) && tiles <= (long)per_cu * cus &&
                           tiles <= (long)SYNC_GROUP * SYNC_MAX_GROUPS;
    }
    *out = p;
    return IPDM_OK;
}

extern "C" int ipdm_art_plan_destroy(ipdm_art_plan *p)
{
    if (!p) return IPDM_OK;
    (void)hipFree(p->d_lut);
    (void)hipFree(p->d_views);
    (void)hipFree(p->d_lines);
    (void)hipFree(p->d_norm);
    (void)hipFree(p->d_abort);
    delete p;
    return IPDM_OK;
}

namespace {
constexpr int SQ_BLOCKS = 64;
struct ArtWs {
    float *x_for, *x_back, *grad;
    float4 *foot;
    unsigned long long *bins;       // 4 x [nb][nr], rotated by the views
    unsigned int *sync;             // grid barrier of the one-launch sweep: arrivals, abort flag
    double *partial;
    ArtState *state;
    size_t bytes;
};
// carve the workspace for a chunk of nb <= BMAX slices (reconstruction) or B slices (projection bins)
ArtWs carve(const ipdm_art_plan *p, void *base, int nb)
{
    const size_t np = (size_t)p->c.nx * p->c.ny;
    char *q = (char *)base;
    auto take = [&](size_t n) { char *r = q; q += align_up(n, 256); return (void *)r; };
    ArtWs w;
    w.x_for = (float *)take(nb * np * sizeof(float));
    w.x_back = (float *)take(nb * np * sizeof(float));
    w.grad = (float *)take(nb * np * sizeof(float));
    w.foot = (float4 *)take(np * 2 * sizeof(float4));
    w.bins = (unsigned long long *)take((size_t)4 * nb * p->c.nr * sizeof(unsigned long long));
    w.sync = (unsigned int *)take(SYNC_WORDS * sizeof(unsigned int));
    w.partial = (double *)take((size_t)nb * SQ_BLOCKS * sizeof(double));
    w.state = (ArtState *)take(nb * sizeof(ArtState));
    w.bytes = (size_t)(q - (char *)base);
    return w;
}
}  // namespace

extern "C" size_t ipdm_art_workspace_bytes(const ipdm_art_plan *p, int32_t B)
{
    if (!p || B <= 0) return 0;
    const size_t recon = carve(p, nullptr, B < BMAX ? B : BMAX).bytes;
    const size_t proj = align_up((size_t)B * p->g.na * p->c.nr * sizeof(unsigned long long), 256);
    return recon > proj ? recon : proj;
}

extern "C" int ipdm_art_reconstruct(ipdm_art_plan *p, const float *d_proj, float *d_volume, int32_t B, int32_t nsart,
                                    int32_t ntv, int32_t sample_rate, void *d_ws, size_t ws_bytes, void *stream)
{
    IPDM_REQUIRE(p && d_proj && d_volume && d_ws && B > 0 && nsart >= 0 && ntv >= 0 && sample_rate >= 1,
                 "art_reconstruct: bad argument");
    if (ws_bytes < ipdm_art_workspace_bytes(p, B)) {
        set_error("art_reconstruct: workspace %zu < %zu", ws_bytes, ipdm_art_workspace_bytes(p, B));
        return IPDM_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const ArtConst &c = p->c;
    const int na = p->g.na / sample_rate;            // PyAPI.cpp:37
    IPDM_REQUIRE(na > 0, "art_reconstruct: sample_rate %d leaves no views", sample_rate);
    const long np = (long)c.nx * c.ny;
    const long proj_stride = (long)p->g.na * c.nr;
    const dim3 tiles(cdiv(c.nx, 16), cdiv(c.ny, 16)), blk(16, 16);
    // 16x16-pixel tiles: 8x8 single-wave workgroups were measured 18 % slower (33 vs 28 us per view at B = 8: four times
    // the sparse flush atomics and correction-window fills for the same pixels)
    constexpr int TS = 16;
    const dim3 stiles(cdiv(c.nx, TS), cdiv(c.ny, TS)), sblk(TS, TS);
    const bool one_launch = p->persistent_ok && nsart > 0;
    if (one_launch) IPDM_HIP_CHECK(hipMemsetAsync(p->d_abort, 0, sizeof(unsigned int), st));
    for (int b0 = 0; b0 < B; b0 += BMAX) {
        const int nb = B - b0 < BMAX ? B - b0 : BMAX;
        ArtWs w = carve(p, d_ws, nb);
        IPDM_HIP_CHECK(hipMemsetAsync(w.x_for, 0, nb * np * sizeof(float), st));    // fbp_volume = 0, PyAPI.cpp:41-42
        const size_t bin_n = (size_t)nb * c.nr;
        IPDM_HIP_CHECK(hipMemsetAsync(d_volume + b0 * np, 0, nb * np * sizeof(float), st));
        hipLaunchKernelGGL(art_state_init_kernel, dim3(1), dim3(64), 0, st, w.state, nb);
        float lamda = 0.24f, sigma = 0.8f;           // .cu:727, :839
        for (int it = 0; it < nsart; ++it) {
            IPDM_HIP_CHECK(hipMemcpyAsync(w.x_back, w.x_for, nb * np * sizeof(float), hipMemcpyDeviceToDevice, st));
            IPDM_HIP_CHECK(hipMemsetAsync(w.bins, 0, 4 * bin_n * sizeof(unsigned long long), st));
            if (p->persistent_ok) {
                // the whole sweep in one launch (grid resident: checked at plan creation)
                IPDM_HIP_CHECK(hipMemsetAsync(w.sync, 0, SYNC_WORDS * sizeof(unsigned int), st));
                PersistArgs pa{c, p->d_lut, p->d_lines, p->d_views, p->d_norm, d_proj + b0 * proj_stride, proj_stride,
                               w.x_for, w.bins, w.sync, na, nb, lamda};
                hipLaunchKernelGGL(art_sweep_persistent_kernel, tiles, blk, 0, st, pa);
                hipLaunchKernelGGL(art_abort_latch_kernel, dim3(1), dim3(1), 0, st, w.sync, p->d_abort);
                IPDM_LAUNCH_CHECK();
            } else {
            SweepArgs a{c, p->d_lut, p->d_lines, p->d_views, p->d_norm, d_proj + b0 * proj_stride, proj_stride,
                        w.x_for, w.foot, nullptr, nullptr, nullptr, -1, 0, nb, lamda};
            for (int v = 0; v <= na; ++v) {
                a.v_prev = v - 1;
                a.v_cur = v < na ? v : -1;
                a.bins_prev = w.bins + (size_t)((v + 2) % 3) * bin_n;
                a.bins_cur = w.bins + (size_t)(v % 3) * bin_n;
                a.bins_next = w.bins + (size_t)((v + 1) % 3) * bin_n;
                hipLaunchKernelGGL(art_sweep_kernel<TS>, stiles, sblk, 0, st, a);
            }
            }
            IPDM_LAUNCH_CHECK();
            hipLaunchKernelGGL(art_sq_partial_kernel, dim3(SQ_BLOCKS, nb), dim3(256), 0, st, w.x_for, w.x_back, np, w.partial);
            hipLaunchKernelGGL(art_state_update_kernel, dim3(nb), dim3(1), 0, st, w.state, w.partial, SQ_BLOCKS, 0);
            IPDM_HIP_CHECK(hipMemcpyAsync(w.x_back, w.x_for, nb * np * sizeof(float), hipMemcpyDeviceToDevice, st));
            IPDM_HIP_CHECK(hipMemcpyAsync(d_volume + b0 * np, w.x_for, nb * np * sizeof(float), hipMemcpyDeviceToDevice, st));
            sigma = sigma * 0.90f;
            sigma = sigma > 0.1f ? sigma : 0.1f;
            for (int itv = 0; itv < ntv; ++itv) {
                hipLaunchKernelGGL(art_tvgrad_kernel, dim3(tiles.x, tiles.y, nb), blk, 0, st, w.x_for, w.grad, c.nx, c.ny, sigma);
                hipLaunchKernelGGL(art_sq_partial_kernel, dim3(SQ_BLOCKS, nb), dim3(256), 0, st, w.grad, (const float *)nullptr, np, w.partial);
                hipLaunchKernelGGL(art_state_update_kernel, dim3(nb), dim3(1), 0, st, w.state, w.partial, SQ_BLOCKS, 1);
                hipLaunchKernelGGL(art_tvstep_kernel, dim3(cdiv(np, 256), nb), dim3(256), 0, st, w.x_for, w.grad, w.state, np);
            }
            hipLaunchKernelGGL(art_sq_partial_kernel, dim3(SQ_BLOCKS, nb), dim3(256), 0, st, w.x_for, w.x_back, np, w.partial);
            hipLaunchKernelGGL(art_state_update_kernel, dim3(nb), dim3(1), 0, st, w.state, w.partial, SQ_BLOCKS, 2);
            lamda = (float)(lamda * 0.95);           // .cu:951
            IPDM_LAUNCH_CHECK();
        }
    }
    if (one_launch) {
        // The hand-written grid barrier of the one-launch sweep needs its whole grid resident; residency was checked on an
        // idle device at plan creation, but other work on the device (another stream, another process) can break it.  An
        // expired barrier poisons the volume with NaN; never hand that back as success: this path ends with ONE stream
        // synchronisation, and a reconstruction that gave up is redone with one launch per view (and the plan stays there).
        unsigned int aborted = 0;
        IPDM_HIP_CHECK(hipMemcpyAsync(&aborted, p->d_abort, sizeof(aborted), hipMemcpyDeviceToHost, st));
        IPDM_HIP_CHECK(hipStreamSynchronize(st));
        if (aborted) {
            p->persistent_ok = false;
            return ipdm_art_reconstruct(p, d_proj, d_volume, B, nsart, ntv, sample_rate, d_ws, ws_bytes, stream);
        }
    }
    return IPDM_OK;
}

extern "C" int ipdm_art_project(ipdm_art_plan *p, const float *d_volume, float *d_proj, int32_t B, void *d_ws,
                                size_t ws_bytes, void *stream)
{
    IPDM_REQUIRE(p && d_volume && d_proj && d_ws && B > 0, "art_project: bad argument");
    if (ws_bytes < ipdm_art_workspace_bytes(p, B)) {
        set_error("art_project: workspace %zu < %zu", ws_bytes, ipdm_art_workspace_bytes(p, B));
        return IPDM_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const long np = (long)p->c.nx * p->c.ny;
    const long nproj = (long)p->g.na * p->c.nr;
    for (int b0 = 0; b0 < B; b0 += BMAX) {
        const int nb = B - b0 < BMAX ? B - b0 : BMAX;
        unsigned long long *bins = (unsigned long long *)d_ws;
        int rc = project_bins(p, d_volume + b0 * np, bins, p->g.na, nb, st);
        if (rc) return rc;
        hipLaunchKernelGGL(art_unfix_kernel, dim3(cdiv(nb * nproj, 256)), dim3(256), 0, st, bins, d_proj + b0 * nproj,
                           nb * nproj, p->c.geodiv);
        IPDM_LAUNCH_CHECK();
    }
    return IPDM_OK;
}
