// Micro-benchmark: what can issue beside v_mfma_f32_32x32x2_f32 on one SIMD of gfx950?
//   mode 0: MFMA-only wave (1 wave/SIMD)                     -> cycles per MFMA
//   mode 1: same wave, K extra independent VALU ops per MFMA  (op: 0 v_fma_f32, 1 v_exp_f32, 2 v_add_u32, 3 ds_write_b32, 4 v_pk_fma_f32)
//   mode 2: two waves per SIMD: waves 0-3 MFMA-only, waves 4-7 run N ops of `op` back-to-back; reports both
// build: hipcc -O3 --offload-arch=gfx950 coissue.hip -o coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ const float *lds_g;
template <int OP>
__device__ __forceinline__ void one_op(float &x, float &y, int &i, f32x2 &p, float *lds, int lane, const float *gptr = nullptr)
{
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i) : "v"(lane));
    if (OP == 3) asm volatile("ds_write_b32 %0, %1" ::"v"(lane * 4), "v"(x) : "memory");
    if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
    if (OP == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
    if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));
    if (OP == 8) { int si = __builtin_amdgcn_readfirstlane(i); asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 5\n s_add_u32 %0, %0, 7" : "+s"(si)); i = si; }
    if (OP == 9) { float tmp; asm volatile("global_load_dword %0, %1, %2" : "=v"(tmp) : "v"(lane * 4), "s"(gptr) : "memory"); }
    if (OP == 10) { float tmp; asm volatile("ds_read_b32 %0, %1" : "=v"(tmp) : "v"(lane * 4) : "memory"); }
}

template <int MODE, int OP, int K>
__global__ void __launch_bounds__(768) kern(float *out, unsigned long long *cyc, int iters)
{
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x = 1.0f + lane * 1e-3f, y = 0.999f;
    int ii = lane;
    f32x2 p = {x, y};
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) lds_g = out;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((MODE != 2 && MODE != 4) || wave < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
                if (MODE == 1) {
#pragma unroll
                    for (int k = 0; k < K; ++k) one_op<OP>(x, y, ii, p, lds, lane);
                }
            }
        }
    } else if (K == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) one_op<OP>(x, y, ii, p, lds, lane, out);
        }
    } else {
        // K independent chains
        float xs[8]; int is[8]; f32x2 ps[8];
        for (int k = 0; k < 8; ++k) { xs[k] = x + k; is[k] = ii + k; ps[k] = p; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) one_op<OP>(xs[k & 7], y, is[k & 7], ps[k & 7], lds, lane);
        }
        for (int k = 0; k < 8; ++k) { x += xs[k]; ii += is[k]; p += ps[k]; }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wave < 8) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = x + y + ii + p[0] + p[1];
    for (int a = 0; a < 4; ++a) s += acc[a][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int OP, int K>
void run(const char *name, int iters_mfma, int iters_other)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int threads = MODE == 4 ? 768 : (MODE == 2 ? 512 : 256);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kern<MODE, OP, K>), dim3(256), dim3(threads), 0, 0, out, cyc, iters_mfma);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double m = 0, o = 0;
    for (int b = 0; b < 256; ++b) { for (int w = 0; w < 4; ++w) m += h[b * 8 + w]; for (int w = 4; w < 8; ++w) o += h[b * 8 + w]; }
    m /= 1024; o /= 1024;
    if (MODE == 2 || MODE == 4) printf("%-44s mfma wave: %6.1f cyc/MFMA   other wave: %6.1f cyc/op (alone-in-time %.0f vs %.0f)\n", name, m / (iters_mfma * 4.0), o / (iters_mfma * 16.0), o, m);
    else printf("%-44s %6.1f cyc/MFMA\n", name, m / (iters_mfma * 4.0));
    hipFree(out); hipFree(cyc);
}

int main()
{
    const int N = 2000;
    run<0, 0, 0>("mfma only", N, 0);
    run<1, 0, 1>("same wave +1 v_fma/MFMA", N, 0);
    run<1, 0, 4>("same wave +4 v_fma/MFMA", N, 0);
    run<1, 0, 8>("same wave +8 v_fma/MFMA", N, 0);
    run<1, 0, 12>("same wave +12 v_fma/MFMA", N, 0);
    run<1, 1, 4>("same wave +4 v_exp/MFMA", N, 0);
    run<1, 2, 8>("same wave +8 v_add_u32/MFMA", N, 0);
    run<1, 3, 4>("same wave +4 ds_write_b32/MFMA", N, 0);
    run<1, 4, 4>("same wave +4 v_pk_fma/MFMA", N, 0);
    run<1, 5, 8>("same wave +8 v_mov/MFMA", N, 0);
    // two waves per SIMD; the other wave runs N*16 ops total in the same launch (it may finish early or late)
    run<2, 0, 0>("partner wave: v_fma stream", N, 0);
    run<2, 1, 0>("partner wave: v_exp stream", N, 0);
    run<2, 2, 0>("partner wave: v_add_u32 stream", N, 0);
    run<2, 3, 0>("partner wave: ds_write_b32 stream", N, 0);
    run<2, 4, 0>("partner wave: v_pk_fma stream", N, 0);
    run<2, 5, 0>("partner wave: v_mov stream", N, 0);
    run<2, 6, 0>("partner wave: v_rcp stream", N, 0);
    run<4, 0, 8>("TWO partner waves/SIMD: v_fma x8 indep", N, 0);
    run<4, 1, 8>("TWO partner waves/SIMD: v_exp x8 indep", N, 0);
    run<2, 8, 0>("partner wave: 4x s_add_u32 per op", N, 0);
    run<2, 9, 0>("partner wave: global_load_dword (saddr)", N, 0);
    run<2, 10, 0>("partner wave: ds_read_b32", N, 0);
    run<2, 0, 8>("partner wave: v_fma x8 independent", N, 0);
    run<2, 1, 8>("partner wave: v_exp x8 independent", N, 0);
    run<2, 2, 8>("partner wave: v_add_u32 x8 independent", N, 0);
    run<2, 4, 8>("partner wave: v_pk_fma x8 independent", N, 0);
    run<2, 6, 8>("partner wave: v_rcp x8 independent", N, 0);
    return 0;
}
