// Micro-benchmark (round 4): two f32-MFMA waves per SIMD, each carrying its own share of "producer" work, against one
// MFMA wave per SIMD with and without a specialised partner wave.  Decides the structure of conv_wino (NOTEBOOK.md).
//
//   per "position" a wave issues 4 dependent v_mfma_f32_32x32x2_f32 on one of 8 accumulators, then its extras:
//     NV plain VALU (v_fma_f32), NT transcendentals (v_exp_f32), NR ds_read_b128, NW ds_write_b32, NG buffer/global b128 loads
//   one s_barrier per 8 positions (BAR = 1).
//   layouts:  L1  256-thread workgroups, one per CU          (1 MFMA wave per SIMD, everything in the MFMA wave)
//             L2  256-thread workgroups, two per CU          (2 MFMA waves per SIMD from DIFFERENT workgroups)
//             L3  512-thread workgroups, one per CU, waves 0-3 MFMA + NR reads only, waves 4-7 all the other extras
//             L4  512-thread workgroups, one per CU, all 8 waves as in L1 (2 MFMA waves per SIMD, ONE barrier domain)
// build: hipcc -O3 --offload-arch=gfx950 dual_mfma.hip -o dual_mfma.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Mix { int nv, nt, nr, nw, ng, bar; };

template <int NV, int NT, int NR, int NW, int NG>
__device__ __forceinline__ void extras(float (&xs)[8], float y, float *lds, int lane, const float *g, bool do_valu, bool do_mem, bool do_reads)
{
    if (do_valu) {
#pragma unroll
        for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(xs[k & 7]) : "v"(y));
#pragma unroll
        for (int k = 0; k < NT; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(xs[(k + 3) & 7]));
    }
    if (do_reads) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            f32x4 t;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(lane * 16), "n"(k * 1024) : "memory");
        }
    }
    if (do_mem) {
#pragma unroll
        for (int k = 0; k < NW; ++k) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(lane * 4), "v"(xs[k & 7]), "n"(8192 + k * 256) : "memory");
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            f32x4 t;
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(t) : "v"(lane * 16), "s"(g), "n"(k * 1024) : "memory");
        }
    }
}

template <int LAYOUT, int NV, int NT, int NR, int NW, int NG, int BAR>
__global__ void __launch_bounds__(LAYOUT >= 3 ? 512 : 256, 2) kern(float *out, const float *g, unsigned long long *cyc, int iters)
{
    __shared__ float lds[6144];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float y = 0.999f, x = 1.0f + lane * 1e-3f;
    float xs[8];
    for (int k = 0; k < 8; ++k) xs[k] = x + k;
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int i = threadIdx.x; i < 6144; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const bool mfma_wave = LAYOUT != 3 || wave < 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (mfma_wave) {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[e]) : "v"(x), "v"(y));
                extras<NV, NT, NR, NW, NG>(xs, y, lds, lane, g, LAYOUT != 3, LAYOUT != 3, true);
            } else {
                extras<NV, NT, NR, NW, NG>(xs, y, lds, lane, g, true, true, false);
            }
        }
        if (BAR) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = 0;
    for (int k = 0; k < 8; ++k) s += xs[k];
    for (int a = 0; a < 8; ++a) s += acc[a][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int LAYOUT, int NV, int NT, int NR, int NW, int NG, int BAR>
void run(const char *name, int iters)
{
    float *out, *g; unsigned long long *cyc;
    const int blocks = LAYOUT == 2 ? 512 : 256, threads = LAYOUT >= 3 ? 512 : 256;
    hipMalloc(&out, 512 * 512 * 4); hipMalloc(&g, 1 << 20); hipMalloc(&cyc, 512 * 8 * 8);
    hipMemset(g, 0, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((kern<LAYOUT, NV, NT, NR, NW, NG, BAR>), dim3(blocks), dim3(threads), 0, 0, out, g, cyc, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    static unsigned long long h[512 * 8];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const int mw_per_block = LAYOUT == 3 ? 4 : threads / 64;
    double c = 0; int n = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < mw_per_block; ++w) { c += h[b * 8 + w]; ++n; }
    c /= n;
    const double mfma_per_wave = (double)iters * 32.0;
    const double mfma_total = mfma_per_wave * mw_per_block * blocks;
    const double tflops = mfma_total * 2.0 * 32 * 32 * 2 / (best * 1e-3) / 1e12;
    const int waves_per_simd = (LAYOUT == 2 || LAYOUT == 4) ? 2 : 1;
    printf("%-58s L%d  %7.1f cyc/MFMA per wave  -> %6.1f cyc/MFMA per SIMD (ideal 64)  %6.1f TFLOP/s = %.3f of 157.3  (%.3f ms)\n", name, LAYOUT,
           c / mfma_per_wave, c / mfma_per_wave / waves_per_simd, tflops, tflops / 157.3, best);
    hipFree(out); hipFree(g); hipFree(cyc);
}

int main()
{
    const int N = 600;
    // bare MFMA streams
    run<1, 0, 0, 0, 0, 0, 0>("mfma only, 1 wave/SIMD", N);
    run<2, 0, 0, 0, 0, 0, 0>("mfma only, 2 waves/SIMD (two WGs)", N);
    // consumer-only mix of today's kernel (2 operand reads + 1 patch read per position), partner idle
    run<1, 0, 0, 3, 0, 0, 1>("3 ds_read per position, barrier/8", N);
    // the merged stream of one wave per position: 7 plain VALU + 3 transcendental, 3 LDS reads, 2 LDS writes, 2 b128 loads
    run<1, 7, 3, 3, 2, 2, 1>("merged mix, 1 wave/SIMD", N);
    run<2, 7, 3, 3, 2, 2, 1>("merged mix, 2 waves/SIMD from two WGs", N);
    run<4, 7, 3, 3, 2, 2, 1>("merged mix, 2 waves/SIMD in ONE WG (one barrier)", N);
    run<3, 7, 3, 3, 2, 2, 1>("specialised: MFMA+reads | partner wave does the rest", N);
    // heavier / lighter VALU shares
    run<2, 4, 2, 3, 2, 2, 1>("lighter VALU (4+2), two WGs", N);
    run<2, 10, 3, 3, 2, 2, 1>("heavier VALU (10+3), two WGs", N);
    run<2, 7, 3, 3, 2, 2, 0>("merged mix, two WGs, no barrier", N);
    run<1, 7, 3, 3, 2, 2, 0>("merged mix, 1 wave/SIMD, no barrier", N);
    run<2, 0, 0, 3, 2, 2, 1>("no VALU at all (3r 2w 2g), two WGs", N);
    run<2, 7, 3, 0, 0, 0, 1>("VALU only (7+3), two WGs", N);
    run<1, 7, 3, 0, 0, 0, 1>("VALU only (7+3), 1 wave/SIMD", N);
    return 0;
}
