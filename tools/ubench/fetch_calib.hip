// Calibration of rocprofv3's FETCH_SIZE for the access widths conv_ws.hip uses (the guide calibrates only 16 B/lane
// reads: they report 1/2).  Each kernel streams a 1 GiB buffer (past the 256 MiB Infinity Cache) exactly once:
//   k_dword   : raw_buffer_load_b32, one dword per lane, a wave reads 256 contiguous bytes   (conv_ws input / residual loads)
//   k_b128    : raw_buffer_load_b128, 16 bytes per lane, a wave reads 1 KiB contiguous        (weights, VEC4 residual)
//   k_rows34  : dword per lane, rows of 34 floats at a 2048-byte pitch (a conv input tile row with its halo)
// build: hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib.bin ; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./fetch_calib.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_dword(const float *p, float *out, long n)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 0x7fffffff, 0x00020000);
    float s = 0;
    const long per = 256L * 64;                       // dwords per workgroup pass
    for (long base = (long)blockIdx.x * per; base < n; base += (long)gridDim.x * per) {
        const float *q = p + base;
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)q, 0, (int)(per * 4), 0x00020000);
#pragma unroll 8
        for (int i = 0; i < 64; ++i) s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (i * 256 + threadIdx.x) * 4, 0, 0));
    }
    (void)r;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_b128(const float *p, float *out, long n)
{
    float s = 0;
    const long per = 256L * 64 * 4;
    for (long base = (long)blockIdx.x * per; base < n; base += (long)gridDim.x * per) {
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)(p + base), 0, (int)(per * 4), 0x00020000);
#pragma unroll 8
        for (int i = 0; i < 64; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rr, (i * 256 + threadIdx.x) * 16, 0, 0);
            s += __builtin_bit_cast(float, v.x) + __builtin_bit_cast(float, v.w);
        }
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// 34 consecutive floats out of every 512 (pitch 2048 B): lanes 0..33 of each wave, the other lanes killed
__global__ void __launch_bounds__(256) k_rows34(const float *p, float *out, long n)
{
    float s = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rows = n / 512;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)(p + row * 512), 0, 2048, 0x00020000);
        s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane < 34 ? lane * 4 : 0x7fffffff, 0, 0));
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main()
{
    const long n = 1L << 28;      // floats = 1 GiB
    float *p, *out;
    hipMalloc(&p, n * 4);
    hipMalloc(&out, 4096);
    hipMemset(p, 0, n * 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_dword, dim3(4096), dim3(256), 0, 0, p, out, n);
        hipLaunchKernelGGL(k_b128, dim3(4096), dim3(256), 0, 0, p, out, n);
        hipLaunchKernelGGL(k_rows34, dim3(4096), dim3(256), 0, 0, p, out, n);
    }
    hipDeviceSynchronize();
    printf("bytes per launch: k_dword %ld, k_b128 %ld, k_rows34 %ld (useful) / %ld (whole 64-B sectors touched) / %ld (whole 128-B lines)\n",
           n * 4, n * 4, n / 512 * 34 * 4, n / 512 * 192, n / 512 * 256);
    return 0;
}
