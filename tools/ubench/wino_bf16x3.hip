// Micro-benchmark (round 6): the chunk loop of a Winograd-domain convolution with the channel contraction on the bf16 matrix pipe
// through an error-free 3-way split of both operands (six products u1v1, u1v2, u2v1, u2v2, u1v3, u3v1 on v_mfma_f32_32x32x16_bf16,
// float32 accumulate), against the same loop on v_mfma_f32_32x32x2_f32 -- the structure of conv_wino2.hip: one 512-thread workgroup
// per CU, a 16-channel chunk = per wave 8 positions x (32 couts x 32 tiles), V operands from LDS, U operands from L2 straight into
// registers, one barrier per chunk, a synthetic staging block of VALU / LDS-store work per chunk.
//   mode 0: f32 (2 x ds_read_b128 + 2 x buffer_load_b128 + 8 MFMA 32x32x2 per position: two 8-channel sub-chunks)
//   mode 1: bf16 x 3 (3 x ds_read_b128 + 3 x buffer_load_b128 + 6 MFMA 32x32x16 per position)
// build: hipcc -O3 --offload-arch=gfx950 wino_bf16x3.hip -o wino_bf16x3.bin ; run: ./wino_bf16x3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV>
__global__ void __launch_bounds__(512) kern(const float *u, float *out, unsigned long long *cyc, int chunks, int u_bytes)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];            // 96 KB of V stage (both modes), written by the staging block
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, u_bytes, 0x00020000);
    f32x16 acc[8];
    for (int e = 0; e < 8; ++e)
        for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
    constexpr int NT = MODE ? 3 : 2;                 // operand loads per position
    f32x4 ua[8][NT];
    const int u_lane = lane * 16, per_pos = NT * 1024, per_wave_chunk = 8 * per_pos;
    int soff = (wave * per_wave_chunk) % (u_bytes - 8 * per_wave_chunk);
    for (int e = 0; e < 8; ++e)
        for (int t = 0; t < NT; ++t) ua[e][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, u_lane, soff + e * per_pos + t * 1024, 0));
    float xs[8];
    for (int k = 0; k < 8; ++k) xs[k] = 1.0f + lane * 1e-3f + k;
    for (int i = threadIdx.x; i < 24576; i += 512) lds[i] = 0.001f * (i & 255);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < chunks; ++c) {
        // ---- synthetic staging: NV vector instructions (one in eight transcendental) + 24 LDS stores
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if ((k & 7) == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(xs[k & 7]));
            else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(xs[k & 7]) : "v"(0.999f));
        }
#pragma unroll
        for (int k = 0; k < 24; ++k) lds[((c & 1) * 12288 + k * 512 + threadIdx.x) % 24576] = xs[k & 7];
        __syncthreads();
        const float *stage = lds + (c & 1) * 12288;
        soff += 8 * per_wave_chunk;
        if (soff > u_bytes - 8 * per_wave_chunk) soff = (wave * per_wave_chunk) % (u_bytes - 8 * per_wave_chunk);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f32x4 b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const f32x4 *>(stage + ((e * NT + t) * 64 + lane) * 4 % 12288);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[e][t][q], b[t][q], acc[e], 0, 0, 0);
            } else {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, ua[e][0]), a1 = __builtin_bit_cast(bf16x8, ua[e][1]), a2 = __builtin_bit_cast(bf16x8, ua[e][2]);
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, b[0]), b1 = __builtin_bit_cast(bf16x8, b[1]), b2 = __builtin_bit_cast(bf16x8, b[2]);
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[e], 0, 0, 0);
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[e], 0, 0, 0);
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[e], 0, 0, 0);
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[e], 0, 0, 0);
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[e], 0, 0, 0);
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[e], 0, 0, 0);
            }
            // the U of this position for the NEXT chunk, into the registers just read
#pragma unroll
            for (int t = 0; t < NT; ++t) ua[e][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, u_lane, soff + e * per_pos + t * 1024, 0));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += xs[k];
    for (int e = 0; e < 8; ++e) s += acc[e][0] + acc[e][7];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int NV>
static void run(const char *name, const float *u, int u_bytes, float *out, unsigned long long *cyc, int chunks)
{
    hipFuncSetAttribute((const void *)kern<MODE, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((kern<MODE, NV>), dim3(256), dim3(512), 98304, 0, u, out, cyc, chunks, u_bytes);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((kern<MODE, NV>), dim3(256), dim3(512), 98304, 0, u, out, cyc, chunks, u_bytes);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2048];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 2048; ++i) mean += h[i];
    mean /= 2048.0 * chunks;
    // a chunk = 512 Winograd-domain tile positions... in conv terms: 32 tiles x 128 couts x 16 positions x 16 channels x 2 flop executed
    const double gflop_exec = 256.0 * chunks * 32 * 128 * 16 * 16 * 2 / 1e9;
    printf("%-34s %8.3f ms  %7.0f s_memtime ticks per chunk  %6.1f TFLOP/s executed-equivalent (f32 MFMA peak 157.3)\n", name, ms, mean, gflop_exec / ms);
}

int main()
{
    const int u_bytes = 3 << 20;                      // 3 MB of packed U: L2-resident, as one layer's weights are
    float *u, *out;
    unsigned long long *cyc;
    hipMalloc(&u, u_bytes); hipMemset(u, 0x3c, u_bytes);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2048 * 8);
    const int chunks = 2000;
    run<0, 0>("f32, MFMA loop only", u, u_bytes, out, cyc, chunks);
    run<0, 110>("f32 + 110 VALU + 24 LDS stores", u, u_bytes, out, cyc, chunks);
    run<1, 0>("bf16x3, MFMA loop only", u, u_bytes, out, cyc, chunks);
    run<1, 110>("bf16x3 + 110 VALU + 24 LDS stores", u, u_bytes, out, cyc, chunks);
    run<1, 200>("bf16x3 + 200 VALU + 24 LDS stores", u, u_bytes, out, cyc, chunks);
    run<1, 260>("bf16x3 + 260 VALU + 24 LDS stores", u, u_bytes, out, cyc, chunks);
    return 0;
}
