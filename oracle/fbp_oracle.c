/* CPU oracle for the FBP domain convertor -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of Recon/FBP_kernel.py (reference, read-only):
 *   ramp filter      conv_pj  :125-131  (np.convolve(h, row)[N-1:2N-1])
 *   back-projection  fbp_cpu  :166-184  (sequential semantics: views in order 0..M-1,
 *                                        float64 geometry, float32 image accumulated through
 *                                        a float64 add per view -- exactly what the numpy
 *                                        scalar expression I[k,i,j] = I[k,i,j] + (...)/L**2 does)
 * Never linked into the product library.  Built by oracle/Makefile (gcc -O2 -fopenmp,
 * -ffp-contract=off so that no FMA contraction changes the rounding).
 */
#include <math.h>
#include <stddef.h>

/* out[k,t,n] = sum_j pj[k,t,j] * h[n + N-1 - j];  float64 accumulation, rounded once. */
void ipdm_oracle_ramp(const float *pj, const float *h, float *out, int BS, int M, int N)
{
    long rows = (long)BS * M;
#pragma omp parallel for schedule(static)
    for (long row = 0; row < rows; ++row) {
        const float *p = pj + row * (long)N;
        float *o = out + row * (long)N;
        for (int n = 0; n < N; ++n) {
            double acc = 0.0;
            for (int j = 0; j < N; ++j)
                acc += (double)p[j] * (double)h[n + N - 1 - j];
            o[n] = (float)acc;
        }
    }
}

/* Same contraction, float32 accumulate in j order (numba/np.convolve-like rounding). */
void ipdm_oracle_ramp_f32(const float *pj, const float *h, float *out, int BS, int M, int N)
{
    long rows = (long)BS * M;
#pragma omp parallel for schedule(static)
    for (long row = 0; row < rows; ++row) {
        const float *p = pj + row * (long)N;
        float *o = out + row * (long)N;
        for (int n = 0; n < N; ++n) {
            float acc = 0.0f;
            for (int j = 0; j < N; ++j)
                acc += p[j] * h[n + N - 1 - j];
            o[n] = acc;
        }
    }
}

/* Back-projection of the pixels listed in pix[] (flat i*gridN+j indices; npix of them) or of
 * all pixels when pix == NULL.  I is [BS, gridN, gridN] float32, updated in place.
 * umap (optional, may be NULL): float64 [M, npix] detector coordinate u(t,pixel) =
 * (alpha - nda0)/da + 0.5 -- the "FBP index map" of BASELINE.json's north_star. */
void ipdm_oracle_backproject(float *I, int BS, const float *pj, const double *phi, const double *r,
                             double D, int gridN, int M, int N, const double *theta, double da,
                             float nda0, const int *pix, int npix, double *umap)
{
    const double half_pi = 3.141592653589793 / 2;
    int total = pix ? npix : gridN * gridN;
#pragma omp parallel for schedule(dynamic, 256)
    for (int q = 0; q < total; ++q) {
        int p = pix ? pix[q] : q;
        double rr = r[p], ph = phi[p];
        for (int t = 0; t < M; ++t) {
            double beta = theta[t] - half_pi;
            double th = half_pi + beta + ph;
            double alpha = atan(rr * sin(th) / (D + rr * cos(th)));
            double u = (alpha - (double)nda0) / da + 0.5;
            if (umap) umap[(size_t)t * total + q] = u;
            double curdet = floor(u);
            if (0 < curdet && curdet < N) {
                double lam = u - curdet;
                double L = rr * sin(th) / sin(alpha);
                int c = (int)curdet;
                for (int k = 0; k < BS; ++k) {
                    const float *row = pj + ((size_t)k * M + t) * N;
                    float *dst = I + (size_t)k * gridN * gridN + p;
                    double inc = ((1 - lam) * (double)row[c - 1] + lam * (double)row[c]) / (L * L);
                    *dst = (float)((double)*dst + inc);
                }
            }
        }
    }
}

/* ---- float64 ARBITER forms (tests only): the same two contractions evaluated in double precision on double data, so that
 * two float32 evaluations (this oracle's and the HIP library's) can each be measured against the value they both
 * approximate.  Constants (taps h, nda0) stay the float32 values the reference builds: they define the function. */
void ipdm_oracle_ramp_f64(const double *pj, const float *h, double *out, int BS, int M, int N)
{
    long rows = (long)BS * M;
#pragma omp parallel for schedule(static)
    for (long row = 0; row < rows; ++row) {
        const double *p = pj + row * (long)N;
        double *o = out + row * (long)N;
        for (int n = 0; n < N; ++n) {
            long double acc = 0.0L;
            for (int j = 0; j < N; ++j)
                acc += (long double)(p[j] * (double)h[n + N - 1 - j]);
            o[n] = (double)acc;
        }
    }
}

void ipdm_oracle_backproject_f64(double *I, int BS, const double *pj, const double *phi, const double *r,
                                 double D, int gridN, int M, int N, const double *theta, double da, float nda0)
{
    const double half_pi = 3.141592653589793 / 2;
    int total = gridN * gridN;
#pragma omp parallel for schedule(dynamic, 256)
    for (int p = 0; p < total; ++p) {
        double rr = r[p], ph = phi[p];
        for (int t = 0; t < M; ++t) {
            double beta = theta[t] - half_pi;
            double th = half_pi + beta + ph;
            double alpha = atan(rr * sin(th) / (D + rr * cos(th)));
            double u = (alpha - (double)nda0) / da + 0.5;
            double curdet = floor(u);
            if (0 < curdet && curdet < N) {
                double lam = u - curdet;
                double L = rr * sin(th) / sin(alpha);
                int c = (int)curdet;
                for (int k = 0; k < BS; ++k) {
                    const double *row = pj + ((size_t)k * M + t) * N;
                    I[(size_t)k * total + p] += ((1 - lam) * row[c - 1] + lam * row[c]) / (L * L);
                }
            }
        }
    }
}
