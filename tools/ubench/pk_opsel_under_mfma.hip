// Micro-benchmark (round 6): does `v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]` (low result = a.lo - b.HI) drop its second
// operand when the wave's own bf16 MFMAs are in flight?  In conv_wino3.hip's multiply-first wave order exactly this instruction returned `a.lo + 0` in lanes
// 48-63, in the first forward of a process only (NOTEBOOK.md round 6; tools/experiments/analyze_trace3.py).  Here: every wave issues a chain of N dependent
// MFMAs, D idle slots, then K packed adds on per-lane operands, each checked against a plain v_sub_f32 of the same operands; waves 4-7 of the workgroup
// (the SIMD partners) either do the same or issue MFMAs only.  The kernel is launched back to back from the first moment of the process (the condition the
// failure needs in the real kernel).  Prints the number of mismatching results per setting.
//   hipcc -O3 --offload-arch=gfx950 pk_opsel_under_mfma.hip -o pk_opsel_under_mfma.bin && ./pk_opsel_under_mfma.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define MFMA "v_mfma_f32_32x32x16_bf16 v[110:125], v[100:103], v[104:107], v[110:125]\n\t"
#define PKCHK \
    "v_pk_add_f32 v[134:135], v[130:131], v[132:133] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]\n\t" \
    "v_sub_f32 v136, v130, v133\n\t"            /* expected low  = a.lo - b.hi */ \
    "v_sub_f32 v137, v133, v131\n\t"            /* expected high = b.hi - a.hi */ \
    "v_cmp_neq_f32 vcc, v134, v136\n\t" \
    "v_addc_co_u32 v140, vcc, 0, v140, vcc\n\t" \
    "v_cmp_neq_f32 vcc, v135, v137\n\t" \
    "v_addc_co_u32 v141, vcc, 0, v141, vcc\n\t" \
    "v_add_f32 v130, 1.0, v130\n\t"             /* new operands for the next one */ \
    "v_add_f32 v133, 0.5, v133\n\t"

template <int N, int D, int K, bool PARTNER_MFMA_ONLY>
__global__ __launch_bounds__(512) void probe(unsigned *bad, int rounds)
{
    extern __shared__ char lds[];                              // 160 KB: one workgroup per CU, two waves per SIMD, as in the real kernel
    const int wave = threadIdx.x >> 6;
    unsigned lo_bad = 0, hi_bad = 0;
    const float fa = 1.0f + (threadIdx.x & 63) * 0.25f, fb = 3.0f + (threadIdx.x & 63) * 0.125f;
    for (int r = 0; r < rounds; ++r) {
        unsigned l, h;
        if (PARTNER_MFMA_ONLY && wave >= 4) {
            asm volatile(
                "v_mov_b32 v100, 0x3f803f80\n\tv_mov_b32 v101, 0x3f803f80\n\tv_mov_b32 v102, 0x3f803f80\n\tv_mov_b32 v103, 0x3f803f80\n\t"
                "v_mov_b32 v104, 0x3f803f80\n\tv_mov_b32 v105, 0x3f803f80\n\tv_mov_b32 v106, 0x3f803f80\n\tv_mov_b32 v107, 0x3f803f80\n\t"
                "s_nop 4\n\t"
                ".rept %c0\n\t" MFMA ".endr\n\t"
                ".rept 8\n\t" MFMA ".endr\n\t"
                "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
                :: "n"(N)
                : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118",
                  "v119", "v120", "v121", "v122", "v123", "v124", "v125");
            continue;
        }
        asm volatile(
            "v_mov_b32 v100, 0x3f803f80\n\tv_mov_b32 v101, 0x3f803f80\n\tv_mov_b32 v102, 0x3f803f80\n\tv_mov_b32 v103, 0x3f803f80\n\t"
            "v_mov_b32 v104, 0x3f803f80\n\tv_mov_b32 v105, 0x3f803f80\n\tv_mov_b32 v106, 0x3f803f80\n\tv_mov_b32 v107, 0x3f803f80\n\t"
            "v_mov_b32 v130, %2\n\tv_add_f32 v131, 7.0, v130\n\tv_mov_b32 v133, %3\n\tv_add_f32 v132, 11.0, v133\n\t"
            "v_mov_b32 v140, 0\n\tv_mov_b32 v141, 0\n\t"
            "s_nop 4\n\t"
            ".rept %c4\n\t" MFMA ".endr\n\t"
            ".if %c5 > 0\n\t.rept %c5\n\ts_nop 0\n\t.endr\n\t.endif\n\t"
            ".rept %c6\n\t" PKCHK ".endr\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v140\n\tv_mov_b32 %1, v141\n\t"
            : "=v"(l), "=v"(h)
            : "v"(fa + r), "v"(fb + r), "n"(N), "n"(D), "n"(K)
            : "vcc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118",
              "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v140", "v141");
        lo_bad += l; hi_bad += h;
    }
    if (lo_bad) atomicAdd(&bad[0], lo_bad);
    if (hi_bad) atomicAdd(&bad[1], hi_bad);
    if (lo_bad && (threadIdx.x & 63) >= 48) atomicAdd(&bad[2], lo_bad);      // ... of which in lanes 48-63
}

template <int N, int D, int K, bool P>
void run(unsigned *d_bad, const char *name)
{
    hipFuncSetAttribute((const void *)probe<N, D, K, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipMemsetAsync(d_bad, 0, 12, 0);
    for (int l = 0; l < 200; ++l) hipLaunchKernelGGL((probe<N, D, K, P>), dim3(256), dim3(512), 160 * 1024, 0, d_bad, 40);      // back to back, no host wait
    unsigned h[3];
    hipMemcpy(h, d_bad, 12, hipMemcpyDeviceToHost);
    const double total = 200.0 * 40 * 256 * 512 * K * (P ? 0.5 : 1.0);
    printf("%-58s  wrong low results %u (lanes 48-63: %u), wrong high results %u   of %.3g packed adds\n", name, h[0], h[2], h[1], total);
    fflush(stdout);
}

int main()
{
    unsigned *d_bad;
    hipMalloc(&d_bad, 12);
    run<6, 0, 16, false>(d_bad, "6 MFMAs, 0 slots, 16 packed adds, every wave");          // (the very first launches of the process)
    run<6, 0, 16, true>(d_bad, "6 MFMAs, 0 slots, 16 packed adds, partners MFMA only");
    run<1, 0, 16, true>(d_bad, "1 MFMA, 0 slots, 16 packed adds, partners MFMA only");
    run<6, 8, 16, true>(d_bad, "6 MFMAs, 8 slots, 16 packed adds, partners MFMA only");
    run<12, 0, 32, true>(d_bad, "12 MFMAs, 0 slots, 32 packed adds, partners MFMA only");
    run<0, 0, 16, false>(d_bad, "no MFMA, 16 packed adds, every wave");
    return 0;
}
