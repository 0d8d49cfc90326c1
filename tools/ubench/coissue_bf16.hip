// Micro-benchmark: what issues beside v_mfma_f32_32x32x16_bf16 on one SIMD of gfx950 (companion of coissue.hip).
//   mode 0: MFMA-only wave;  mode 1: same wave + K ops per MFMA;  mode 2: partner wave running an op stream
// build: hipcc -O3 --offload-arch=gfx950 coissue_bf16.hip -o coissue_bf16.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OP>
__device__ __forceinline__ void one_op(float &x, float y, float *lds, int lane, f32x4 &ld)
{
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (OP == 3) asm volatile("ds_write_b32 %0, %1" ::"v"(lane * 4), "v"(x) : "memory");
    if (OP == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(lane * 16) : "memory");
    if (OP == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
}

template <int MODE, int OP, int K>
__global__ void __launch_bounds__(512) kern(float *out, unsigned long long *cyc, int iters)
{
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x = 1.0f + lane * 1e-3f, y = 0.999f;
    f32x4 ld = {0, 0, 0, 0};
    float xs[8];
    for (int k = 0; k < 8; ++k) xs[k] = x + k;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE != 2 || wave < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
                if (MODE == 1) {
#pragma unroll
                    for (int k = 0; k < K; ++k) one_op<OP>(xs[k & 7], y, lds, lane, ld);
                }
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) one_op<OP>(xs[k & 7], y, lds, lane, ld);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = x + ld[0] + ld[3];
    for (int k = 0; k < 8; ++k) s += xs[k];
    for (int q = 0; q < 4; ++q) s += acc[q][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int OP, int K>
void run(const char *name, int iters)
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    (void)hipMemset(cyc, 0, 256 * 8 * 8);
    const int threads = MODE == 2 ? 512 : 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<MODE, OP, K>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kern<MODE, OP, K>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256 * 8];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double m = 0, o = 0;
    for (int b = 0; b < 256; ++b) { for (int w = 0; w < 4; ++w) m += h[b * 8 + w]; for (int w = 4; w < 8; ++w) o += h[b * 8 + w]; }
    m /= 1024; o /= 1024;
    const double tf = 256.0 * 4 * iters * 4 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    if (MODE == 2) printf("%-44s %6.1f cyc/MFMA  other wave %6.1f cyc/op  (%.0f vs %.0f ticks) %.0f TF/s %.2f GHz\n", name, m / (iters * 4.0), o / (iters * 16.0), o, m, tf, m / (ms * 1e-3) / 1e9);
    else printf("%-44s %6.1f cyc/MFMA   %.0f TF/s bf16  clock %.2f GHz\n", name, m / (iters * 4.0), tf, m / (ms * 1e-3) / 1e9);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    const int N = 20000;
    run<0, 0, 0>("bf16 mfma only", N);
    run<1, 0, 2>("same wave +2 v_fma/MFMA", N);
    run<1, 0, 4>("same wave +4 v_fma/MFMA", N);
    run<1, 4, 1>("same wave +1 ds_read_b128/MFMA", N);
    run<1, 4, 2>("same wave +2 ds_read_b128/MFMA", N);
    run<1, 3, 2>("same wave +2 ds_write_b32/MFMA", N);
    run<2, 0, 0>("partner: v_fma x8 independent", N);
    run<2, 1, 0>("partner: v_exp x8 independent", N);
    run<2, 5, 0>("partner: v_cvt_pk_bf16 x8 independent", N);
    run<2, 3, 0>("partner: ds_write_b32", N);
    run<2, 4, 0>("partner: ds_read_b128", N);
    return 0;
}
