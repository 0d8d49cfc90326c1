// Probe: how does a 16-byte raw buffer load behave when only part of it lies inside num_records?
// hipcc --offload-arch=gfx950 -O2 tools/ubench/oob_probe.hip -o /tmp/oob_probe && /tmp/oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float *src, float *out, int nrec_bytes)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nrec_bytes, 0x00020000);
    // lane l loads 16 bytes at byte offset nrec - 16 + 4*l  (l = 0: fully inside, l = 1..3: 1..3 dwords past the end, l = 4: fully outside)
    const int l = threadIdx.x;
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, nrec_bytes - 16 + 4 * l, 0, 0));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
    // the same through the scalar offset
    f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, 4 * l, nrec_bytes - 16, 0));
    for (int e = 0; e < 4; ++e) out[64 + l * 4 + e] = w[e];
    // negative per-lane offset
    f32x4 n = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, -4 * l, 0, 0));
    for (int e = 0; e < 4; ++e) out[128 + l * 4 + e] = n[e];
}
int main()
{
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
    float *d, *o; hipMalloc(&d, 256); hipMalloc(&o, 192 * 4);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(8), 0, 0, d, o, 128);     // buffer = first 32 floats (values 100..131)
    float r[192]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int k = 0; k < 3; ++k) {
        printf("%s\n", k == 0 ? "voffset = nrec-16+4l:" : k == 1 ? "soffset = nrec-16, voffset = 4l:" : "voffset = -4l:");
        for (int l = 0; l < 6; ++l) printf("  l=%d: %g %g %g %g\n", l, r[k * 64 + l * 4], r[k * 64 + l * 4 + 1], r[k * 64 + l * 4 + 2], r[k * 64 + l * 4 + 3]);
    }
    return 0;
}
