// Fan-beam filtered back-projection (the domain convertor) for gfx950.
//
// Replaces Recon/FBP_kernel.py of the reference: FBP.__init__/getrphi (:28-84) -> plan_create,
// FBP.convert (:86-122) -> ipdm_fbp_forward, conv_pj/conv_kernel (:125-143) -> ramp_kernel,
// fbp_cpu/fbp_kernel (:146-184) -> backproject_kernel.  Written from scratch for CDNA4:
//   * ramp: one workgroup per (view, slice); the flipped, cos-weighted row and the ramp taps live
//     in LDS; only the 912/2+1 non-zero taps of h_RL are visited; no atomics (the reference's CUDA
//     kernel scatters with float atomicAdd);
//   * back-projection: one thread owns one pixel for ALL views (in order, so the result is
//     deterministic and equals the reference's sequential semantics), fp64 geometry shared by
//     the KB slices a thread carries in registers, f32 image accumulated through an fp64 add per
//     view exactly as numpy's scalar expression does.
#include <cmath>
#include <vector>
#include "common.h"

using namespace ipdm;

struct ipdm_fbp_plan {
    ipdm_fbp_geom g;
    std::vector<double> theta, phi, r;
    std::vector<float> nda, h, weight;
    double D, dtheta;
    double *d_theta = nullptr, *d_phi = nullptr, *d_r = nullptr;
    float *d_h = nullptr, *d_weight = nullptr;
};

extern "C" int ipdm_fbp_plan_create(const ipdm_fbp_geom *geom, ipdm_fbp_plan **out)
{
    IPDM_REQUIRE(geom && out, "fbp_plan_create: null argument");
    IPDM_REQUIRE(geom->n_views > 1 && geom->n_det > 1 && geom->grid_n > 0 && geom->da > 0,
                 "fbp_plan_create: bad geometry");
    // the ramp kernel visits only the taps of the parity that is non-zero for an even detector count (h_RL has its
    // non-zero taps at even indices and at the centre N-1, Recon/FBP_kernel.py:52-56, N = 912 there)
    IPDM_REQUIRE(geom->n_det % 2 == 0, "fbp_plan_create: n_det %d must be even", geom->n_det);
    ipdm_fbp_plan *p = new ipdm_fbp_plan();
    p->g = *geom;
    const int M = geom->n_views, N = geom->n_det, G = geom->grid_n;
    const double pi = 3.141592653589793;
    p->D = fabs(-geom->source_origin);
    // theta = arange(0, .., dtheta_deg)/180*pi  (Recon/FBP_kernel.py:38)
    p->theta.resize(M);
    for (int t = 0; t < M; ++t) p->theta[t] = ((double)t * geom->dtheta_deg) / 180 * pi;
    p->dtheta = p->theta[1] - p->theta[0];
    // nda = arange(start, stop, da).astype(f32); numpy fills start + i*((start+da)-start) (:39-40)
    p->nda.resize(N);
    {
        double start = (-(double)N / 2 + 0.5 + geom->det_offset) * geom->da;
        double delta = (start + geom->da) - start;
        for (int i = 0; i < N; ++i) p->nda[i] = (float)(start + (double)i * delta);
    }
    // ramp taps (:52-56): h[2m] = -0.5/pi^2/sin^2((2m-(N-1))*da) * da ; h[N-1] = 1/8/da^2 * da
    p->h.assign(2 * N - 1, 0.0f);
    for (int m = 0; m < N; ++m) {
        double ng = (double)(-N + 1 + 2 * m) * geom->da;
        double s = sin(ng);
        p->h[2 * m] = (float)((-0.5 / (pi * pi) / (s * s)) * geom->da);
    }
    p->h[N - 1] = (float)((1.0 / 8 / (geom->da * geom->da)) * geom->da);
    // per-detector weight D*cos(nda) in float32 (:104: np.cos of a float32 array stays float32)
    p->weight.resize(N);
    for (int i = 0; i < N; ++i) p->weight[i] = (float)p->D * cosf(p->nda[i]);
    // polar pixel coordinates (:69-84)
    p->phi.resize((size_t)G * G);
    p->r.resize((size_t)G * G);
    const double cx = (double)G / 2, cy = (double)G / 2;
    for (int ii = 0; ii < G; ++ii)
        for (int jj = 0; jj < G; ++jj) {
            double i = ii + 1, j = jj + 1;
            double y = ((double)G + 1 - i - cx - 0.5) * 2 * geom->fov_half / G;
            double x = (j - cy - 0.5) * 2 * geom->fov_half / G;
            double rr = sqrt(x * x + y * y);
            double ph = atan(y / x);
            if (x < 0) ph += pi;
            if (ph < 0) ph += 2 * pi;
            p->r[(size_t)ii * G + jj] = rr;
            p->phi[(size_t)ii * G + jj] = ph;
        }
#define UP(dst, vec, T)                                                                     \
    IPDM_HIP_CHECK(hipMalloc((void **)&p->dst, p->vec.size() * sizeof(T)));                 \
    IPDM_HIP_CHECK(hipMemcpy(p->dst, p->vec.data(), p->vec.size() * sizeof(T), hipMemcpyHostToDevice));
    UP(d_theta, theta, double)
    UP(d_phi, phi, double)
    UP(d_r, r, double)
    UP(d_h, h, float)
    UP(d_weight, weight, float)
#undef UP
    *out = p;
    return IPDM_OK;
}

extern "C" int ipdm_fbp_plan_destroy(ipdm_fbp_plan *p)
{
    if (!p) return IPDM_OK;
    (void)hipFree(p->d_theta);
    (void)hipFree(p->d_phi);
    (void)hipFree(p->d_r);
    (void)hipFree(p->d_h);
    (void)hipFree(p->d_weight);
    delete p;
    return IPDM_OK;
}

extern "C" size_t ipdm_fbp_workspace_bytes(const ipdm_fbp_plan *p, int32_t B)
{
    if (!p || B <= 0) return 0;
    return align_up((size_t)B * p->g.n_views * p->g.n_det * sizeof(float), 256);
}

extern "C" int64_t ipdm_fbp_table(const ipdm_fbp_plan *p, int32_t which, void *host_out, int64_t cap)
{
    if (!p) return IPDM_ERR_INVALID;
    const void *src = nullptr;
    int64_t n = 0;
    size_t es = 8;
    switch (which) {
        case 0: src = p->theta.data(); n = p->theta.size(); break;
        case 1: src = p->phi.data(); n = p->phi.size(); break;
        case 2: src = p->r.data(); n = p->r.size(); break;
        case 3: src = p->nda.data(); n = p->nda.size(); es = 4; break;
        case 4: src = p->h.data(); n = p->h.size(); es = 4; break;
        case 5: src = p->weight.data(); n = p->weight.size(); es = 4; break;
        default: set_error("fbp_table: bad selector %d", which); return IPDM_ERR_INVALID;
    }
    if (host_out) {
        if (cap < n) { set_error("fbp_table: capacity %ld < %ld", (long)cap, (long)n); return IPDM_ERR_INVALID; }
        memcpy(host_out, src, (size_t)n * es);
    }
    return n;
}

// ------------------------------------------------------------------------------------ ramp
// out[k,t,n] = sum_j pjw[j] * h[n + N-1 - j],  pjw[j] = (gain*pj[k,t,flip? N-1-j : j]) * w[j] * dtheta
// h is non-zero only for even index (and the centre N-1), i.e. (n - j) odd or n == j.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) ramp_kernel(const float *__restrict__ sino, float *__restrict__ out,
                                                     const float *__restrict__ h, const float *__restrict__ w,
                                                     int N, int flip, float gain, float dtheta)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *row = lds;            // N
    float *taps = lds + N;       // 2N-1
    const size_t rowoff = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * N;
    const float *src = sino + rowoff;
    for (int j = threadIdx.x; j < N; j += BLOCK) {
        float v = src[flip ? (N - 1 - j) : j];
        if (gain != 1.0f) v = gain * v;
        v = v * w[j];
        row[j] = v * dtheta;
    }
    for (int j = threadIdx.x; j < 2 * N - 1; j += BLOCK) taps[j] = h[j];
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += BLOCK) {
        double acc = (double)row[n] * (double)taps[N - 1];
        // j of opposite parity to n
        for (int j = (n + 1) & 1; j < N; j += 2) acc = fma((double)row[j], (double)taps[n + N - 1 - j], acc);
        out[rowoff + n] = (float)acc;
    }
}

// ------------------------------------------------------------------------------ back-projection
template <int KB>
__global__ void __launch_bounds__(256) backproject_kernel(const float *__restrict__ pj, float *__restrict__ img,
                                                          const double *__restrict__ phi, const double *__restrict__ r,
                                                          const double *__restrict__ theta, double D, double da,
                                                          double nda0, int G, int M, int N, int B, int k0, int flip)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= G * G) return;
    const double half_pi = 3.141592653589793 / 2;
    const double rr = r[p], ph = phi[p];
    float acc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) acc[k] = 0.0f;
    const size_t slice_stride = (size_t)M * N;
    const float *base = pj + (size_t)k0 * slice_stride;
    for (int t = 0; t < M; ++t) {
        double beta = theta[t] - half_pi;
        double th = half_pi + beta + ph;
        double s, c;
        sincos(th, &s, &c);
        double alpha = atan(rr * s / (D + rr * c));
        double u = (alpha - nda0) / da + 0.5;
        double curdet = floor(u);
        if (0 < curdet && curdet < (double)N) {
            double lam = u - curdet;
            double L = rr * s / sin(alpha);
            double inv = 1.0 / (L * L);
            (void)inv;
            int cd = (int)curdet;
            const float *rowp = base + (size_t)t * N + cd;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                if (k0 + k < B) {
                    double a = (double)rowp[(size_t)k * slice_stride - 1];
                    double b = (double)rowp[(size_t)k * slice_stride];
                    double inc = ((1 - lam) * a + lam * b) / (L * L);
                    acc[k] = (float)((double)acc[k] + inc);
                }
            }
        }
    }
    const int i = p / G, j = p % G;
    const int jo = flip ? (G - 1 - j) : j;
#pragma unroll
    for (int k = 0; k < KB; ++k)
        if (k0 + k < B) img[((size_t)(k0 + k) * G + i) * G + jo] = acc[k];
}

__global__ void index_map_kernel(const int *__restrict__ pix, int npix, double *__restrict__ u_out,
                                 const double *__restrict__ phi, const double *__restrict__ r,
                                 const double *__restrict__ theta, double D, double da, double nda0, int M)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (q >= npix) return;
    const double half_pi = 3.141592653589793 / 2;
    const int p = pix[q];
    double beta = theta[t] - half_pi;
    double th = half_pi + beta + phi[p];
    double s, c;
    sincos(th, &s, &c);
    double alpha = atan(r[p] * s / (D + r[p] * c));
    u_out[(size_t)t * npix + q] = (alpha - nda0) / da + 0.5;
}

extern "C" int ipdm_fbp_filter(ipdm_fbp_plan *p, const float *d_sino, float *d_filtered, int32_t B, int32_t flip,
                               float gain, void *stream)
{
    IPDM_REQUIRE(p && d_sino && d_filtered && B > 0, "fbp_filter: bad argument");
    const int N = p->g.n_det;
    size_t lds = (size_t)(3 * N - 1) * sizeof(float);
    IPDM_REQUIRE(lds <= 64 * 1024, "fbp_filter: n_det %d too large for the LDS-resident ramp", N);
    dim3 grid(p->g.n_views, B);
    hipLaunchKernelGGL(ramp_kernel<256>, grid, dim3(256), lds, (hipStream_t)stream, d_sino, d_filtered, p->d_h,
                       p->d_weight, N, flip, gain, (float)p->dtheta);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

extern "C" int ipdm_fbp_backproject(ipdm_fbp_plan *p, const float *d_filtered, float *d_img, int32_t B, int32_t flip,
                                    void *stream)
{
    IPDM_REQUIRE(p && d_filtered && d_img && B > 0, "fbp_backproject: bad argument");
    const int G = p->g.grid_n;
    dim3 grid(cdiv((long)G * G, 256));
    hipStream_t st = (hipStream_t)stream;
    const double nda0 = (double)p->nda[0];
#define BP(KB, k0)                                                                                             \
    hipLaunchKernelGGL(backproject_kernel<KB>, grid, dim3(256), 0, st, d_filtered, d_img, p->d_phi, p->d_r,   \
                       p->d_theta, p->D, p->g.da, nda0, G, p->g.n_views, p->g.n_det, B, k0, flip)
    int k0 = 0;
    while (k0 < B) {
        int left = B - k0;
        if (left >= 8) { BP(8, k0); k0 += 8; }
        else if (left >= 4) { BP(4, k0); k0 += 4; }
        else if (left >= 2) { BP(2, k0); k0 += 2; }
        else { BP(1, k0); k0 += 1; }
    }
#undef BP
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

extern "C" int ipdm_fbp_forward(ipdm_fbp_plan *p, const float *d_sino, float *d_img, int32_t B, int32_t flip,
                                float gain, void *d_ws, size_t ws_bytes, void *stream)
{
    IPDM_REQUIRE(p && d_ws, "fbp_forward: bad argument");
    if (ws_bytes < ipdm_fbp_workspace_bytes(p, B)) {
        set_error("fbp_forward: workspace %zu < %zu", ws_bytes, ipdm_fbp_workspace_bytes(p, B));
        return IPDM_ERR_WORKSPACE;
    }
    int rc = ipdm_fbp_filter(p, d_sino, (float *)d_ws, B, flip, gain, stream);
    if (rc) return rc;
    return ipdm_fbp_backproject(p, (const float *)d_ws, d_img, B, flip, stream);
}

extern "C" int ipdm_fbp_index_map(ipdm_fbp_plan *p, const int32_t *d_pix, int32_t npix, double *d_u, void *stream)
{
    IPDM_REQUIRE(p && d_pix && d_u && npix > 0, "fbp_index_map: bad argument");
    dim3 grid(cdiv(npix, 128), p->g.n_views);
    hipLaunchKernelGGL(index_map_kernel, grid, dim3(128), 0, (hipStream_t)stream, d_pix, npix, d_u, p->d_phi,
                       p->d_r, p->d_theta, p->D, p->g.da, (double)p->nda[0], p->g.n_views);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}

// ------------------------------------------------------------------------------------ sharpen
__global__ void sharpen_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W, float wc,
                               float wn)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const float *s = in + (size_t)blockIdx.z * H * W;
    float acc = 0.0f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int yy = y + dy, xx = x + dx;
            float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? s[(size_t)yy * W + xx] : 0.0f;
            acc += v * ((dy == 0 && dx == 0) ? wc : wn);
        }
    out[((size_t)blockIdx.z * H + y) * W + x] = acc;
}

extern "C" int ipdm_sharpen3x3(const float *d_in, float *d_out, int32_t B, int32_t H, int32_t W, float n, void *stream)
{
    IPDM_REQUIRE(d_in && d_out && B > 0 && H > 0 && W > 0, "sharpen3x3: bad argument");
    // kernel [[-2,-2,-2],[-2,N,-2],[-2,-2,-2]]/(N-16)  (Utils/train_test_utils.py:871-874), float32 division
    const float wc = n / (n - 16.0f), wn = -2.0f / (n - 16.0f);
    dim3 grid(cdiv(W, 128), H, B);
    hipLaunchKernelGGL(sharpen_kernel, grid, dim3(128), 0, (hipStream_t)stream, d_in, d_out, H, W, wc, wn);
    IPDM_LAUNCH_CHECK();
    return IPDM_OK;
}
