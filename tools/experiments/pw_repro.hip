// Reproduction harness of the shelved round-4 pointwise kernel (tools/experiments/conv_pw.hip): that kernel
// produced "a run-dependent handful of exact zeros (accumulator register 0, lanes 12-15 / 28-31)" and was shelved without a cause.
// This program compiles THAT source as it stands, links against the product library for the helpers it calls, launches the kernel
// `reps` times per case on fixed inputs and compares every launch with a plain reference kernel: how many launches differ from the
// reference, where (cout block lane, accumulator register), and whether the wrong values are exact zeros.
//   build:  make -C tools/experiments            ->  tools/experiments/pw_repro   (needs ipdm-pytorch_amd/libipdm_hip.so)
//   run:    tools/experiments/pw_repro [reps]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "conv_pw.hip"

using namespace ipdm;

namespace {
__global__ void ref_kernel(const float *x, const float *w, const float *bias, const float *res, float *out, int Cin, int Cout, int HW)
{
    const int p = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y, n = blockIdx.z;
    if (p >= HW) return;
    float acc = 0.0f;
    for (int c = 0; c < Cin; ++c) acc = fmaf(x[((size_t)n * Cin + c) * HW + p], w[(size_t)co * Cin + c], acc);      // (the MFMA's order: channels ascending)
    acc += bias[co];
    if (res) acc += res[((size_t)n * Cout + co) * HW + p];
    out[((size_t)n * Cout + co) * HW + p] = acc;
}

unsigned rng_state = 12345u;
float frand() { rng_state = rng_state * 1664525u + 1013904223u; return (float)((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }

int run_case(int B, int Cin, int Cout, int H, int W, bool with_res, int reps)
{
    const int HW = H * W;
    std::vector<float> hx((size_t)B * Cin * HW), hw((size_t)Cout * Cin), hb(Cout), hr((size_t)B * Cout * HW), packed;
    for (auto &v : hx) v = frand();
    for (auto &v : hw) v = frand() / sqrtf((float)Cin);
#ifdef PW_NO_BIAS_MFMA
    for (auto &v : hb) v = 0.0f;
#else
    for (auto &v : hb) v = frand();
#endif
    for (auto &v : hr) v = frand();
    conv_pack_weights_pw(hw.data(), Cout, Cin, packed);
    float *dx, *dw, *dwp, *db, *dr, *dout, *dref;
    hipMalloc(&dx, hx.size() * 4); hipMalloc(&dw, hw.size() * 4); hipMalloc(&dwp, packed.size() * 4); hipMalloc(&db, hb.size() * 4);
    hipMalloc(&dr, hr.size() * 4); hipMalloc(&dout, hr.size() * 4); hipMalloc(&dref, hr.size() * 4);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dwp, packed.data(), packed.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(ref_kernel, dim3((HW + 255) / 256, Cout, B), dim3(256), 0, nullptr, dx, dw, db, with_res ? dr : nullptr, dref, Cin, Cout, HW);
    std::vector<float> href(hr.size()), hout(hr.size());
    hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost);
    ConvArgs a;
    a.x1 = dx; a.x2 = nullptr; a.C1 = Cin; a.C2 = 0; a.B = B; a.Hs = H; a.Ws = W; a.H = H; a.W = W; a.upsample = 0; a.scale_y = a.scale_x = 1.f;
    a.w = dwp; a.w_wino = dwp; a.cout_pad = Cout; a.w_interleave = 4; a.bias = db; a.Cout = Cout; a.ksize = 1; a.stride = 1; a.Ho = H; a.Wo = W;
    a.act = 0; a.gn_scale = a.gn_shift = nullptr; a.res = with_res ? dr : nullptr; a.out = dout; a.tiles_x = a.tiles_y = a.co_tiles = 0;
    int bad_launches = 0, zeros = 0, wrong = 0, hist_lane[32] = {0}, hist_px[32] = {0}, hist_blk[16] = {0}, shown = 0;
    for (int rep = 0; rep < reps; ++rep) {
        hipMemset(dout, 0xff, hout.size() * 4);
        if (conv2d_pw_launch(a, nullptr) != 0) { printf("launch refused: %s\n", ipdm_last_error()); return -1; }
        hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t i = 0; i < hout.size(); ++i)
            if (hout[i] != href[i] && !(fabsf(hout[i] - href[i]) <= 1e-5f * fmaxf(1.0f, fabsf(href[i])))) {
                ++bad;
                const int p = (int)(i % HW), co = (int)(i / HW % Cout);
                if (hout[i] == 0.0f) ++zeros;
                if (shown < 12) { printf("   wrong at n %d cout %d pixel %d (pixel block of 32: %d)\n", (int)(i / ((size_t)HW * Cout)), co, p, p >> 5); ++shown; }
                ++hist_blk[(p >> 5) & 15];
                ++hist_lane[co & 31];
                ++hist_px[p & 31];
            }
        wrong += bad;
        bad_launches += bad ? 1 : 0;
    }
    printf("B%d %d->%d @%dx%d res%d: %d of %d launches differ from the reference; %d wrong values, %d of them exact zeros\n", B, Cin, Cout, H, W,
           (int)with_res, bad_launches, reps, wrong, zeros);
    if (wrong) {
        printf("   by cout & 31:");
        for (int i = 0; i < 32; ++i) printf(" %d", hist_lane[i]);
        printf("\n   by pixel & 31:");
        for (int i = 0; i < 32; ++i) printf(" %d", hist_px[i]);
        printf("\n   by (pixel >> 5) & 15:");
        for (int i = 0; i < 16; ++i) printf(" %d", hist_blk[i]);
        printf("\n");
    }
    hipFree(dx); hipFree(dw); hipFree(dwp); hipFree(db); hipFree(dr); hipFree(dout); hipFree(dref);
    return bad_launches;
}
}  // namespace

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    int total = 0;
    total += run_case(8, 256, 768, 57, 125, false, reps);      // qkv at T = 7125: the shape round 4 saw the zeros on
    total += run_case(8, 256, 256, 57, 125, true, reps);       // proj_out + residual
    total += run_case(2, 256, 768, 64, 64, false, reps);       // RPW = 4 / 2 launches
    total += run_case(1, 128, 128, 228, 500, true, reps);
    total += run_case(8, 128, 128, 96, 96, false, reps);
    printf("total launches with wrong values: %d\n", total);
    return 0;
}
