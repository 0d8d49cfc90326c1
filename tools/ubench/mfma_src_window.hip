// Micro-benchmark (round 6): for how long after its ISSUE does a v_mfma_f32_32x32x16_bf16 still read its A operand?  (Result on MI355X: it does not --
// every setting below is exact: the hardware orders VALU writes, LDS returns and vector-memory returns behind an issued MFMA's source reads;
// profiles/r06_mfma_src_window.txt.  Written to test the suspected cause of conv_wino3.hip's first failure, which it rules out.)
// A wave issues a chain of N dependent MFMAs (same accumulator, A = B = ones: every MFMA adds 16 to every element), waits D cycles (s_nop),
// overwrites the A registers with zeros (four v_mov_b32), drains, and reads the accumulator: 16 N if every MFMA had read A before the
// overwrite, less by 16 per MFMA that had not (by less than 16 if it had read part of it).  The partner wave of the SIMD (w + 4) either idles
// or issues MFMA chains of its own the whole time (the matrix pipe is shared by the two waves).  All in one asm statement on named registers:
// nothing is scheduled or padded by the compiler inside it.
//   hipcc -O3 --offload-arch=gfx950 mfma_src_window.hip -o mfma_src_window.bin && ./mfma_src_window.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define NOP16 "s_nop 15\n\t"
#define DRAIN NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16
#define MFMA "v_mfma_f32_32x32x16_bf16 v[110:125], v[100:103], v[104:107], v[110:125]\n\t"

// the same with the overwrite done by an LDS READ that returns zeros into the A registers (issued D cycles behind the chain)
template <int N, int D>
__device__ __forceinline__ float one_trial_lds(unsigned lds_addr)
{
    float r;
    asm volatile(
        "v_mov_b32 v100, 0x3f803f80\n\tv_mov_b32 v101, 0x3f803f80\n\tv_mov_b32 v102, 0x3f803f80\n\tv_mov_b32 v103, 0x3f803f80\n\t"
        "v_mov_b32 v104, 0x3f803f80\n\tv_mov_b32 v105, 0x3f803f80\n\tv_mov_b32 v106, 0x3f803f80\n\tv_mov_b32 v107, 0x3f803f80\n\t"
        "v_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\tv_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\tv_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\t"
        "v_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\tv_mov_b32 v120, 0\n\tv_mov_b32 v121, 0\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\tv_mov_b32 v124, 0\n\tv_mov_b32 v125, 0\n\t"
        "s_nop 7\n\t"
        ".rept %c2\n\t" MFMA ".endr\n\t"
        ".if %c3 > 0\n\t.rept %c3\n\ts_nop 0\n\t.endr\n\t.endif\n\t"
        "ds_read_b128 v[100:103], %1\n\t"
        DRAIN
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32 %0, v117\n\t"
        : "=v"(r)
        : "v"(lds_addr), "n"(N), "n"(D)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
          "v120", "v121", "v122", "v123", "v124", "v125", "memory");
    return r;
}

// ... and by a VECTOR-MEMORY load (global_load_dwordx4 of zeros, L1-resident after the first trial) into the A registers
template <int N, int D>
__device__ __forceinline__ float one_trial_vmem(const float *zeros)
{
    float r;
    asm volatile(
        "v_mov_b32 v100, 0x3f803f80\n\tv_mov_b32 v101, 0x3f803f80\n\tv_mov_b32 v102, 0x3f803f80\n\tv_mov_b32 v103, 0x3f803f80\n\t"
        "v_mov_b32 v104, 0x3f803f80\n\tv_mov_b32 v105, 0x3f803f80\n\tv_mov_b32 v106, 0x3f803f80\n\tv_mov_b32 v107, 0x3f803f80\n\t"
        "v_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\tv_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\tv_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\t"
        "v_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\tv_mov_b32 v120, 0\n\tv_mov_b32 v121, 0\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\tv_mov_b32 v124, 0\n\tv_mov_b32 v125, 0\n\t"
        "s_nop 7\n\t"
        ".rept %c2\n\t" MFMA ".endr\n\t"
        ".if %c3 > 0\n\t.rept %c3\n\ts_nop 0\n\t.endr\n\t.endif\n\t"
        "global_load_dwordx4 v[100:103], %1, off\n\t"
        DRAIN DRAIN
        "s_waitcnt vmcnt(0)\n\t"
        "v_mov_b32 %0, v117\n\t"
        : "=v"(r)
        : "v"(zeros), "n"(N), "n"(D)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
          "v120", "v121", "v122", "v123", "v124", "v125", "memory");
    return r;
}

template <int N, int D>
__device__ __forceinline__ float one_trial()
{
    float r;
    asm volatile(
        "v_mov_b32 v100, 0x3f803f80\n\tv_mov_b32 v101, 0x3f803f80\n\tv_mov_b32 v102, 0x3f803f80\n\tv_mov_b32 v103, 0x3f803f80\n\t"
        "v_mov_b32 v104, 0x3f803f80\n\tv_mov_b32 v105, 0x3f803f80\n\tv_mov_b32 v106, 0x3f803f80\n\tv_mov_b32 v107, 0x3f803f80\n\t"
        "v_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\tv_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\tv_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\t"
        "v_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\tv_mov_b32 v120, 0\n\tv_mov_b32 v121, 0\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\tv_mov_b32 v124, 0\n\tv_mov_b32 v125, 0\n\t"
        "s_nop 7\n\t"
        ".rept %c1\n\t" MFMA ".endr\n\t"
        ".if %c2 > 0\n\t.rept %c2\n\ts_nop 0\n\t.endr\n\t.endif\n\t"
        "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
        DRAIN
        "v_mov_b32 %0, v117\n\t"
        : "=v"(r)
        : "n"(N), "n"(D)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
          "v120", "v121", "v122", "v123", "v124", "v125", "memory");
    return r;
}

template <int N, int D, int PARTNER, int LDS = 0>
__global__ void __launch_bounds__(512) kern(float *out, int iters, const float *gz)
{
    __shared__ __attribute__((aligned(16))) float zeros[512 * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 512) zeros[i] = 0.f;
    __syncthreads();
    const unsigned lds_addr = (unsigned)(size_t)(zeros + threadIdx.x * 4);      // (LDS byte address: the low 32 bits of the generic pointer)
    float lo = 1e30f, hi = -1e30f, sum = 0.f;
    if (wave < 4) {
        for (int it = 0; it < iters; ++it) {
            const float r = LDS == 2 ? one_trial_vmem<N, D>(gz + threadIdx.x * 4) : LDS ? one_trial_lds<N, D>(lds_addr) : one_trial<N, D>();
            lo = fminf(lo, r); hi = fmaxf(hi, r); sum += r;
        }
    } else if (PARTNER) {      // the other wave of the SIMD keeps the matrix pipe busy with chains of its own
        for (int it = 0; it < iters * 3; ++it) (void)one_trial<8, 0>();
    }
    if (wave < 4) {
        float *o = out + ((size_t)blockIdx.x * 4 + wave) * 64 * 3 + lane * 3;
        o[0] = lo; o[1] = hi; o[2] = sum / iters;
    }
}

template <int N, int D, int PARTNER, int LDS = 0>
static void run(float *d_out)
{
    const int iters = 2000, blocks = 256;
    static float *gz = nullptr;
    if (!gz) { hipMalloc(&gz, 512 * 16); hipMemset(gz, 0, 512 * 16); }
    hipLaunchKernelGGL((kern<N, D, PARTNER, LDS>), dim3(blocks), dim3(512), 0, 0, d_out, iters, (const float *)gz);
    hipDeviceSynchronize();
    static float h[256 * 4 * 64 * 3];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    float lo = 1e30f, hi = -1e30f;
    double mean = 0;
    long bad = 0;
    for (int i = 0; i < 256 * 4 * 64; ++i) {
        lo = fminf(lo, h[3 * i]); hi = fmaxf(hi, h[3 * i + 1]); mean += h[3 * i + 2];
        bad += h[3 * i] != 16.0f * N;
    }
    printf("%s chain of %d, overwrite %3d cycles behind the last issue, partner %s: accumulator min %6.1f max %6.1f mean %8.3f (expected %d)  lanes that ever saw less: %ld of %d\n",
           LDS == 2 ? "[vmem    ]" : LDS ? "[ds_read ]" : "[v_mov   ]", N, D, PARTNER ? "busy" : "idle", lo, hi, mean / (256 * 4 * 64), 16 * N, bad, 256 * 4 * 64);
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 4 * 64 * 3 * sizeof(float));
    run<1, 0, 0>(d); run<1, 0, 1>(d);
    run<6, 0, 0>(d); run<6, 0, 1>(d);
    run<6, 16, 0>(d); run<6, 16, 1>(d);
    run<6, 32, 1>(d); run<6, 64, 1>(d); run<6, 96, 1>(d); run<6, 128, 1>(d); run<6, 160, 1>(d); run<6, 192, 1>(d); run<6, 256, 1>(d); run<6, 384, 1>(d);
    run<2, 0, 1>(d); run<2, 32, 1>(d); run<2, 64, 1>(d);
    run<1, 0, 0, 1>(d); run<1, 0, 1, 1>(d); run<2, 0, 1, 1>(d); run<3, 0, 1, 1>(d); run<4, 0, 0, 1>(d); run<4, 0, 1, 1>(d);
    run<6, 0, 0, 1>(d); run<6, 0, 1, 1>(d); run<6, 32, 1, 1>(d); run<6, 64, 1, 1>(d); run<6, 96, 1, 1>(d); run<6, 128, 1, 1>(d); run<6, 192, 1, 1>(d);
    run<12, 0, 0, 1>(d); run<12, 0, 1, 1>(d); run<12, 128, 1, 1>(d); run<12, 256, 1, 1>(d); run<12, 384, 1, 1>(d);
    run<1, 0, 1, 2>(d); run<6, 0, 0, 2>(d); run<6, 0, 1, 2>(d); run<6, 64, 1, 2>(d); run<6, 128, 1, 2>(d); run<12, 0, 0, 2>(d); run<12, 0, 1, 2>(d); run<12, 128, 1, 2>(d); run<12, 256, 1, 2>(d);
    return 0;
}
