/* CPU oracle: triangle-area-LUT SART reconstruction + forward projector ("ART" convertor).
 *
 * TEST INFRASTRUCTURE ONLY -- linked into oracle/libipdm_oracle.so; never part of the product.
 * PARITY UNPINNED: the reference implementation of this path is CUDA + thrust + libtorch
 * (Recon/TASART2DNSL0-Cpp/TASART2DNSL0.cu, driven by TASART2DNSL0_PyAPI.cpp) and cannot be built
 * or run in this image, and the reference holds no outputs of it.  This file restates that
 * source's arithmetic sequentially, each function citing the lines it follows; the only piece
 * checked against reference DATA is the area table (oracle/art.py::area_lut vs Recon/Simens_alut.txt,
 * equal to 1e-17).
 *
 * Where the CUDA code is not sequential the restatement takes the exact value the unordered
 * operation approximates: the atomicAdd float sums of the projector (.cu:381) are accumulated in
 * double and rounded once; thrust norms (.cu:129-133) likewise.  Texture fetches: point-filtered
 * ones are plain indexed loads with clamped coordinates; the linearly filtered area table
 * (.cu:262, cudaFilterModeLinear) is exact float bilinear interpolation (the hardware's 8-bit
 * interpolation weights are not modelled).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float dso, dsd;
    int nx, ny;
    float dx, dy, offset_x, offset_y;
    int nr;
    float dr, offset_r, angle_start;
    int na, ta_dimx, ta_dimy;
    float ta_deltax, ta_deltay;
} art_geom; /* Parameters, TASART2DNSL0.h:23-42 */

#define NFOOT 5
#define PI_F ((float)M_PI)

typedef struct {
    float src_x, src_y, uvs_x, uvs_y, uvt_x, uvt_y, beta;
} view_t;

/* host side of one view, .cu:853-861 (rotateCCW_z :147-150) */
static view_t view_setup(const art_geom *g, float beta_deg)
{
    view_t v;
    const float beta = (beta_deg - g->angle_start) * (PI_F / 180.0f);
    const float cs = cosf(beta), sn = sinf(beta);
    v.beta = beta;
    v.uvt_x = 0.0f * cs - (-1.0f) * sn;
    v.uvt_y = 0.0f * sn + (-1.0f) * cs;
    v.uvs_x = 1.0f * cs - 0.0f * sn;
    v.uvs_y = 1.0f * sn + 0.0f * cs;
    v.src_x = 0.0f * cs - g->dso * sn;
    v.src_y = 0.0f * sn + g->dso * cs;
    return v;
}

/* update_lines_kernel, .cu:270-302: the nr+1 bin-edge rays of a view as (folded angle, unit normal, offset) */
static void update_lines(const art_geom *g, const view_t *v, float *lines /* [nr+1][4] */)
{
    const float rr = g->nr * g->dr * 0.5f;
    for (int is = 0; is <= g->nr; ++is) {
        const float s0 = -rr + g->offset_r * g->dr;
        const float gamma = s0 + is * g->dr;
        const float p1x = v->src_x + g->dsd * sinf(v->beta + gamma);
        const float p1y = v->src_y + -g->dsd * cosf(v->beta + gamma);
        const float rx = p1x - v->src_x, ry = p1y - v->src_y;
        float ang = atan2f(ry, rx) * (360.0f / (2.0f * PI_F));
        if (ang < 0.0f) ang += 360.0f;
        const float A = p1y - v->src_y;
        const float B = v->src_x - p1x;
        const float C = p1x * v->src_y - v->src_x * p1y;
        const float Z = sqrtf(A * A + B * B);
        if (ang <= 45.0f) { ; }
        else if (ang <= 90.0f) ang = 90.0f - ang;
        else if (ang <= 135.0f) ang = ang - 90.0f;
        else if (ang <= 180.0f) ang = 180.0f - ang;
        else if (ang <= 225.0f) ang = ang - 180.0f;
        else if (ang <= 270.0f) ang = 270.0f - ang;
        else if (ang <= 315.0f) ang = ang - 270.0f;
        else ang = 360.0f - ang;
        lines[is * 4 + 0] = ang;
        lines[is * 4 + 1] = A / Z;
        lines[is * 4 + 2] = B / Z;
        lines[is * 4 + 3] = C / Z;
    }
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* tex2D on the area table: unnormalised coordinates, linear filter, clamp addressing (.cu:580-598) */
static float lut_fetch(const art_geom *g, const float *lut, float x, float y)
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx0 = floorf(xb), fy0 = floorf(yb);
    const float ax = xb - fx0, ay = yb - fy0;
    const int x0 = clampi((int)fx0, 0, g->ta_dimx - 1), x1 = clampi((int)fx0 + 1, 0, g->ta_dimx - 1);
    const int y0 = clampi((int)fy0, 0, g->ta_dimy - 1), y1 = clampi((int)fy0 + 1, 0, g->ta_dimy - 1);
    const float t00 = lut[y0 * g->ta_dimx + x0], t10 = lut[y0 * g->ta_dimx + x1];
    const float t01 = lut[y1 * g->ta_dimx + x0], t11 = lut[y1 * g->ta_dimx + x1];
    return (1.0f - ay) * ((1.0f - ax) * t00 + ax * t10) + ay * ((1.0f - ax) * t01 + ax * t11);
}

/* fetchAreaLut, .cu:253-268 */
static float fetch_area(const art_geom *g, const float *lut, const float *lines, int sidx, float x, float y)
{
    sidx = clampi(sidx, 0, g->nr);
    const float *L = lines + sidx * 4;
    const float ang = L[0];
    const float pos = L[1] * x + L[2] * y + L[3];
    const float ox = 1.0f / g->ta_deltax, oy = 1.0f / g->ta_deltay;
    const float value = lut_fetch(g, lut, fabsf(pos) * ox + 0.5f, ang * oy + 0.5f);
    const float vox = fabsf(g->dx * g->dy);
    return pos < 0.0f ? vox - value : value;
}

/* lut_init_foot_kernel, .cu:304-341: per pixel the source distance, first bin and NFOOT strip areas */
static void footprint(const art_geom *g, const view_t *v, const float *lut, const float *lines, int ix, int iy,
                      float *dist, int *s_bin, float *foot)
{
    const float xx = g->nx * g->dx * 0.5f, yy = g->ny * g->dy * 0.5f;
    const float x = (float)((ix + 0.5) * g->dx - xx * 1.0f + g->offset_x);
    const float y = (float)((iy + 0.5) * g->dy - yy * 1.0f + g->offset_y);
    *dist = sqrtf((x - v->src_x) * (x - v->src_x) + (y - v->src_y) * (y - v->src_y));
    const float ds = v->uvs_x * x + v->uvs_y * y;
    const float dt = v->uvt_x * x + v->uvt_y * y;
    const float gamma = atanf(ds / (dt + g->dso));
    const int sb = (int)floorf(gamma / g->dr + 0.5f * (g->nr - 1) - g->offset_r) - NFOOT / 2;
    *s_bin = sb;
    int is = sb;
    float a0 = fetch_area(g, lut, lines, is, x, y);
    ++is;
    for (int k = 0; k < NFOOT; ++k, ++is) {
        const float a1 = fetch_area(g, lut, lines, is, x, y);
        foot[k] = fabsf(a0 - a1);
        a0 = a1;
    }
}

/* One view of the forward projector: lut_fp_kernel (.cu:343-383) + apply_geodiv_kernel (:385-393).
 * vol == NULL projects a volume of ones (val > 0 in the CUDA code). */
static void forward_view(const art_geom *g, const view_t *v, const float *lut, const float *lines, const float *vol,
                         float *proj /* [nr] */, double *acc /* [nr] scratch */)
{
    memset(acc, 0, sizeof(double) * g->nr);
    for (int iy = 0; iy < g->ny; ++iy)
        for (int ix = 0; ix < g->nx; ++ix) {
            const float att = vol ? vol[iy * g->nx + ix] : 1.0f;
            if (att == 0.0f) continue;
            float dist, foot[NFOOT];
            int sb;
            footprint(g, v, lut, lines, ix, iy, &dist, &sb, foot);
            const float div = att / dist;
            for (int k = 0; k < NFOOT; ++k) {
                const int is = sb + k;
                if (is < 0 || is >= g->nr || foot[k] <= 0.0f || div == 0.0f) continue;
                acc[is] += (double)(div * foot[k]);
            }
        }
    const float geodiv = 1.0f / g->dr; /* _cmpGeoDiv, .cu:601-611 */
    for (int i = 0; i < g->nr; ++i) proj[i] = (float)acc[i] * geodiv;
}

/* Projection_torch / DoProjection, PyAPI.cpp:64-80, .cu:1335-1438 */
void art_oracle_project(const art_geom *g, const float *lut, const float *betas, const float *vol, float *proj)
{
    float *lines = (float *)malloc(sizeof(float) * 4 * (g->nr + 1));
    double *acc = (double *)malloc(sizeof(double) * g->nr);
    for (int ia = 0; ia < g->na; ++ia) {
        const view_t v = view_setup(g, betas[ia]);
        update_lines(g, &v, lines);
        forward_view(g, &v, lut, lines, vol, proj + (size_t)ia * g->nr, acc);
    }
    free(lines);
    free(acc);
}

static float norm2(const float *x, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (double)x[i] * x[i];
    return (float)sqrt(s);
}

static float texv(const art_geom *g, const float *vol, int ix, int iy)
{
    return vol[clampi(iy, 0, g->ny - 1) * g->nx + clampi(ix, 0, g->nx - 1)];
}

/* Grad_NSL0TV, .cu:483-540 */
static void grad_nsl0tv(const art_geom *g, const float *vol, float sigma, float *grad)
{
    const float mins = 0.0001f;
    for (int iy = 0; iy < g->ny; ++iy)
        for (int ix = 0; ix < g->nx; ++ix) {
            const float c = texv(g, vol, ix, iy), xp = texv(g, vol, ix + 1, iy), yp = texv(g, vol, ix, iy + 1);
            const float xm = texv(g, vol, ix - 1, iy), ym = texv(g, vol, ix, iy - 1);
            const float xmyp = texv(g, vol, ix - 1, iy + 1), xpym = texv(g, vol, ix + 1, iy - 1);
            const float Dxy = sqrtf(mins * mins + (c - xp) * (c - xp) + (c - yp) * (c - yp));
            const float Dxm = sqrtf(mins * mins + (xm - c) * (xm - c) + (xm - xmyp) * (xm - xmyp));
            const float Dym = sqrtf(mins * mins + (ym - c) * (ym - c) + (ym - xpym) * (ym - xpym));
            const float e1 = expf(Dxy / (2 * sigma)) + expf(-Dxy / (2 * sigma));
            const float e2 = expf(Dxm / (2 * sigma)) + expf(-Dxm / (2 * sigma));
            const float e3 = expf(Dym / (2 * sigma)) + expf(-Dym / (2 * sigma));
            const float Wxy = (2 / sigma) / (e1 * e1), Wxm = (2 / sigma) / (e2 * e2), Wym = (2 / sigma) / (e3 * e3);
            float t = 0;
            t += Wxy * (c - xp + c - yp) / Dxy;
            t -= Wxm * (xm - c) / Dxm;
            t -= Wym * (ym - c) / Dym;
            if (t < mins * mins) t = 0;
            grad[iy * g->nx + ix] = t;
        }
}

/* Reconstruction_torch / DoReconstruction, PyAPI.cpp:33-59, .cu:721-975.  proj = [na][nr] of one slice, out = [ny][nx]
 * (not permuted; the caller applies the PyAPI's permute).  The start volume is zero (PyAPI.cpp:41-42). */
void art_oracle_reconstruct(const art_geom *g, const float *lut, const float *betas, const float *proj, float *out,
                            int nsart, int ntv)
{
    const int np = g->nx * g->ny;
    float lamda = 0.24f, alpha = 0.1f, sigma = 0.8f;
    float *x_for = (float *)calloc(np, sizeof(float)), *x_back = (float *)calloc(np, sizeof(float));
    float *x_res = (float *)calloc(np, sizeof(float)), *grad = (float *)calloc(np, sizeof(float));
    float *lines = (float *)malloc(sizeof(float) * 4 * (g->nr + 1));
    float *cur = (float *)malloc(sizeof(float) * g->nr), *nrm = (float *)malloc(sizeof(float) * g->nr);
    double *acc = (double *)malloc(sizeof(double) * g->nr);
    const float geodiv = 1.0f / g->dr;
    for (int it = 0; it < nsart; ++it) {
        memcpy(x_back, x_for, sizeof(float) * np);
        for (int ia = 0; ia < g->na; ++ia) {
            const view_t v = view_setup(g, betas[ia]);
            update_lines(g, &v, lines);
            forward_view(g, &v, lut, lines, x_for, cur, acc);      /* _Fp_Ax(cur_proj, footinfo, -1) */
            forward_view(g, &v, lut, lines, NULL, nrm, acc);       /* _Fp_Ax(norm_proj, footinfo, 1) */
            for (int i = 0; i < g->nr; ++i) {                      /* correction_kernel, .cu:443-460 */
                const float m = proj[(size_t)ia * g->nr + i];
                cur[i] = nrm[i] > 0.0f ? geodiv * ((m - cur[i]) / nrm[i]) : 0.0f;
            }
#pragma omp parallel for schedule(static)
            for (int iy = 0; iy < g->ny; ++iy)                     /* lut_bp_kernel x2 + update_kernel, .cu:397-481 */
                for (int ix = 0; ix < g->nx; ++ix) {
                    float dist, foot[NFOOT];
                    int sb;
                    footprint(g, &v, lut, lines, ix, iy, &dist, &sb, foot);
                    const float div = 1.0f / dist;
                    float bp = 0.0f, nb = 0.0f;
                    for (int k = 0; k < NFOOT; ++k) {
                        const int is = clampi(sb + k, 0, g->nr - 1);   /* clamp-addressed point fetch */
                        bp += cur[is] * div * foot[k];
                        nb += geodiv * div * foot[k];
                    }
                    const float upd = nb > 0.0f ? lamda * (bp / nb) : 0.0f;
                    x_for[iy * g->nx + ix] = fmaxf(x_for[iy * g->nx + ix] + upd, 0.0f);
                }
        }
        for (int i = 0; i < np; ++i) x_back[i] = -1.0f * x_for[i] + x_back[i];
        const float dp = norm2(x_back, np);
        memcpy(x_back, x_for, sizeof(float) * np);
        memcpy(x_res, x_for, sizeof(float) * np);
        sigma = sigma * 0.90f;
        sigma = sigma > 0.1f ? sigma : 0.1f;
        const float dtvg = alpha * dp;
        for (int itv = 0; itv < ntv; ++itv) {
            grad_nsl0tv(g, x_for, sigma, grad);
            for (int i = 0; i < np; ++i)                            /* nonnegative, .cu:543-558 */
                if (x_for[i] < 0) x_for[i] = 0;
            const float normg = norm2(grad, np);
            const float a = -1.0f * dtvg / normg;
            for (int i = 0; i < np; ++i) x_for[i] = a * grad[i] + x_for[i];
        }
        for (int i = 0; i < np; ++i) x_back[i] = -1.0f * x_for[i] + x_back[i];
        const float dg = norm2(x_back, np);
        if (dg > (0.995 * dp)) alpha = alpha * 0.96;
        lamda = lamda * 0.95;
    }
    memcpy(out, x_res, sizeof(float) * np);
    free(x_for); free(x_back); free(x_res); free(grad); free(lines); free(cur); free(nrm); free(acc);
}
