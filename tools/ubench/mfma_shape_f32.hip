// Micro-benchmark (round 4): does the clock an MI355X holds under an f32-MFMA load depend on the instruction SHAPE, as it does
// for bf16 (MI355X_MICROARCH.md, DVFS give-back item 7)?  Bare dependent-chain loops on pseudo-random operands, one and two
// waves per SIMD: v_mfma_f32_32x32x2_f32 (8 accumulators of 16 registers) against v_mfma_f32_16x16x4_f32 (32 accumulators of 4
// registers -- the same 128 accumulator registers and the same multiply-adds per loop trip).
// build: hipcc -O3 --offload-arch=gfx950 mfma_shape_f32.hip -o mfma_shape_f32.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float rnd(unsigned s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (float)(s & 0xffff) / 65536.0f - 0.5f; }

template <int SHAPE>
__global__ void __launch_bounds__(512) kern(float *out, unsigned long long *cyc, unsigned long long *rt, int iters)
{
    const int lane = threadIdx.x & 63;
    float x[8], y[8];
    for (int k = 0; k < 8; ++k) { x[k] = rnd(threadIdx.x * 977 + blockIdx.x * 131 + k * 7 + 1); y[k] = rnd(threadIdx.x * 613 + blockIdx.x * 17 + k * 29 + 5); }
    f32x16 a32[8];
    f32x4 a16[32];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) a32[a][r] = 0.f;
    for (int a = 0; a < 32; ++a) for (int r = 0; r < 4; ++r) a16[a][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 32) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a32[e]) : "v"(x[(e + q) & 7]), "v"(y[(e + 2 * q) & 7]));
        } else {
            // the same multiply-adds: 8 positions x 4 k-steps of 32x32x2 = 65536 MACs = 64 instructions of 16x16x4
#pragma unroll
            for (int e = 0; e < 32; ++e)
#pragma unroll
                for (int q = 0; q < 2; ++q) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a16[e]) : "v"(x[(e + q) & 7]), "v"(y[(e + 3 * q) & 7]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0; rt[blockIdx.x * 8 + (threadIdx.x >> 6)] = r1 - r0; }
    float s = 0;
    for (int a = 0; a < 8; ++a) s += a32[a][0] + a32[a][7];
    for (int a = 0; a < 32; ++a) s += a16[a][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SHAPE>
void run(const char *name, int threads, int iters)
{
    float *out; unsigned long long *cyc, *rt;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8); (void)hipMalloc(&rt, 256 * 8 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((kern<SHAPE>), dim3(256), dim3(threads), 0, 0, out, cyc, rt, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    static unsigned long long h[256 * 8], hr[256 * 8];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost); (void)hipMemcpy(hr, rt, sizeof hr, hipMemcpyDeviceToHost);
    double c = 0, r = 0; const int nw = threads / 64;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) { c += h[b * 8 + w]; r += hr[b * 8 + w]; }
    const double macs = (double)iters * 65536.0 * nw * 256;
    const double tf = 2.0 * macs / (best * 1e-3) / 1e12;
    printf("%-52s %7.1f TFLOP/s = %.3f of 157.3   in-kernel clock %.2f GHz   cycles per 2048 MACs and SIMD %.1f (ideal 64)   %.3f ms\n", name, tf, tf / 157.3,
           c / r * 0.1, c / nw / 256 / (iters * 32.0) * (nw / 4.0), best);
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(rt);
}

int main()
{
    const int N = 3000;
    run<32>("32x32x2, 1 wave/SIMD", 256, N);
    run<16>("16x16x4, 1 wave/SIMD", 256, N);
    run<32>("32x32x2, 2 waves/SIMD", 512, N);
    run<16>("16x16x4, 2 waves/SIMD", 512, N);
    run<32>("32x32x2, 1 wave/SIMD (again)", 256, N);
    run<16>("16x16x4, 1 wave/SIMD (again)", 256, N);
    return 0;
}
