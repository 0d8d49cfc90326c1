// Internal (non-ABI) interfaces between the UNet executor and its kernels.
#pragma once
#include <vector>
#include "common.h"

namespace ipdm {

// The same statistics from the per-tile partial sums the producing convolutions left behind (ConvArgs::stats) instead of
// a pass over the activations: up to two sources (channel concat), each [B][rows][C][2] float32.
struct GnTileSrc { const float *stats = nullptr; int rows = 0, C = 0; };
struct GnTileArgs {
    GnTileSrc src[2];
    int nsrc = 1, B = 0;
    long HW = 0;
    int groups = 0;
    const float *gamma = nullptr, *beta = nullptr;
    float eps = 1e-5f;
    double *partials = nullptr;            // [B, groups, GN_SPLIT, 2]
    float *scale = nullptr, *shift = nullptr;        // [B, C1+C2]
};

struct ConvArgs {
    const float *x1, *x2;        // sources [B,C1,Hs,Ws], [B,C2,Hs,Ws] (x2 may be null)
    int C1, C2, B;
    int Hs, Ws;                  // source spatial size
    int H, W;                    // (virtual) conv-input size; != (Hs,Ws) => nearest up-sampling
    int upsample;
    float scale_y, scale_x;      // Hs/H, Ws/W in float32 (ATen's nearest rule)
    const float *w;              // packed [Cin_pad][k*k][cout_pad]
    int cout_pad;
    int w_interleave = 0;        // weight-slab layout chosen at pack time: 0 plain [cout], MB>0: cout-interleaved [32][MB] per 32*MB group
    const float *bias;           // [Cout] or null
    int Cout, ksize, stride;
    int Ho, Wo;
    int act;                     // 0 none, 1 GroupNorm, 2 GroupNorm+SiLU (prologue on the input)
    const float *gn_scale, *gn_shift;   // [B, C1+C2]
    const float *res;            // [B,Cout,Ho,Wo] or null
    float *out;
    int tiles_x, tiles_y, co_tiles;     // filled by the launcher
    // Fused GroupNorm statistics of the OUTPUT (for the GroupNorm that follows this convolution): when non-null, every
    // (tile, consumer wave) / workgroup writes one row of per-channel partial sums {sum, sum of squares} of the final
    // output values it produced: stats[((n * stats_rows + row) * Cout + c) * 2 + {0,1}] (float32 partials over <= a few
    // hundred pixels; combined in float64 by gn_tiles_launch).  stats_rows = conv_stats_rows(args).
    float *stats = nullptr;
    int stats_rows = 0;
    // K split of layers with too few output tiles to fill the chip (batch 1, low resolutions): conv_split(args) slices of
    // the input-channel range are summed by separate workgroups into split_ws [ksplit][B,Cout,Ho,Wo] (caller-provided,
    // conv_split_ws_bytes(args) bytes) and a combine pass adds them in a fixed order (+ bias, residual, statistics).
    // ksplit is set by the launcher; callers only provide split_ws (null: never split).
    float *split_ws = nullptr;
    int ksplit = 1;
    // The convolution of a 2x nearest up-sampled image (Upsample: F.interpolate + 3x3 conv) as four 2x2-tap convolutions on
    // the source grid, one per output parity, over weights whose coinciding taps were added up at pack time
    // (conv_pack_weights_up2): 4 instead of 9 multiply-adds per output.  The caller passes the ordinary arguments plus
    // w_up2; when conv_up2_eligible(args) the launcher takes this path and writes `out` PARITY-PLANAR:
    // [n][cout][row & 1][col & 1][Ho/2][Wo/2] (same channel stride as NCHW).  Readers set x1_planar (conv_planar_ok).
    const float *w_up2 = nullptr;
    // ... and, for the layers conv_wup2_shape_ok accepts, the parity filters in the Winograd F(2x2,2x2) domain
    // (conv_pack_weights_wup2): 9 instead of 16 multiply-adds per 2x2 outputs of a parity, same parity-planar output and
    // statistics rows (conv_wup2.hip; conv_wup2_eligible).
    const float *w_wup2 = nullptr;
    // The same layer's weights in the Winograd F(2x2,3x3) domain (conv_pack_weights_wino: U = G g G^T); when
    // conv_wino_eligible(args) the launcher evaluates the convolution there (conv_wino.hip): 16 instead of 36
    // multiply-adds per 2x2 outputs, same NCHW output and statistics rows as the direct kernel.
    const float *w_wino = nullptr;
    // The 1x1 shortcut of a ResidualBlock whose channel count changes (Model/model.py:116-130), folded into the block's
    // SECOND 3x3 convolution on the narrow levels (conv_direct.hip, conv_direct_skip_ok): the block input (sk_x1 [, sk_x2],
    // same spatial size as this layer's input, sk_x1 possibly parity-planar) and the shortcut's packed weights
    // [Cin][1][sk_cout_pad]; `bias` then carries both biases and `res` stays null.
    const float *sk_x1 = nullptr, *sk_x2 = nullptr, *sk_w = nullptr;
    int sk_C1 = 0, sk_C2 = 0, sk_cout_pad = 0, sk_planar = 0;
    int x1_planar = 0;                  // x1 is stored parity-planar (the output of an up2 convolution)
    int up2 = 0;                        // set by the launcher
    int dbg = 0;                        // IPDM_CONV_DBG bit mask (kernel experiments only; 0 on the product path)
    unsigned long long *dbg_buf = nullptr;   // dbg & 8: per-workgroup cycle stamps [grid][4]
};

int conv2d_launch(const ConvArgs &a, hipStream_t st);
int conv_kernel_code(const ConvArgs &a);                  // which kernel conv2d_launch would take (codes: include/ipdm_hip.h)
int conv2d_ws_launch(const ConvArgs &a, hipStream_t st);   // persistent wave-specialised variant (conv_ws.hip)
bool conv_direct_eligible(const ConvArgs &a);             // narrow layers: direct packed-f32 VALU kernel (conv_direct.hip)
int conv2d_direct_launch(const ConvArgs &a, hipStream_t st);
bool conv_nm_eligible(const ConvArgs &a);                 // ... of those, the stride-1 layers that run on the 16-cout MFMA (conv_nm.hip)
int conv2d_nm_launch(const ConvArgs &a, hipStream_t st);
int conv_k_chunk();   // concat inputs must split at a multiple of this many channels (3x3 kernels)
int conv_ws_k_chunk(int ks, int interleave);   // K chunk of the kernel a (ks, weight layout) pair runs on: also its concat alignment

// per-launch HIP-event timing of kernel classes (bench.py roofline): 0 = conv 3x3 s1 wide tile (the
// dominant kernel), 1 = every other conv variant, 2 = attention
// 3 = the Winograd-domain form of class 0's layers, recorded with its EXECUTED flops (16/36 of the 3x3 count)
// 4 = the narrow direct convolutions (conv_direct.hip), bandwidth-bound: recorded with their algorithmic HBM BYTES
// 5 = the 128-cout-tile Winograd kernel (conv_wino2.hip; class 3 keeps the 64-cout-tile kernel), EXECUTED flops
// 6 = the narrow direct convolutions that READ a wide tensor (>= 64 input channels): f32-VALU-bound, recorded with their flops
// 7 = the wide Upsample layers in the Winograd F(2x2,2x2) domain of their parity form (conv_wup2.hip), EXECUTED flops (9 products per source pixel)
constexpr int PROF_CLASSES = 8;
bool prof_enabled();
void prof_before(int cls, hipStream_t st);
void prof_after(int cls, double flops, hipStream_t st);
// which weight layout / kernel family a convolution of this shape uses (0 = plain layout, legacy kernels of conv.hip)
int conv_weight_interleave(int Cout, int ks, int stride);
void conv_pack_weights(const float *w, int Cout, int Cin, int ks, int interleave, std::vector<float> &packed, int &cin_pad, int &cout_pad);

// GroupNorm statistics over (possibly concatenated) NCHW sources -> per-(sample,channel) affine
// scale/shift: y = x*scale + shift == gamma*(x-mean)*rstd + beta.
struct GnArgs {
    const float *x1, *x2;
    int C1, C2, B;
    long HW;
    int groups;
    const float *gamma, *beta;   // [C1+C2]
    float eps;
    double *partials;            // [B, groups, GN_SPLIT, 2]
    float *scale, *shift;        // [B, C1+C2]
    int split = 0;               // workgroups per (sample, group), chosen by the launcher (<= GN_SPLIT)
};
constexpr int GN_SPLIT = 64;
int gn_tiles_launch(const GnTileArgs &a, hipStream_t st);
// rows of ConvArgs::stats per sample the kernel chosen for this convolution writes (0: that kernel has no fused statistics)
int conv_stats_rows(const ConvArgs &a);
int conv_split(const ConvArgs &a);                 // K slices the launcher would use given a split workspace (1: no split)
size_t conv_split_ws_bytes(const ConvArgs &a);     // 0 when conv_split(a) == 1
constexpr int SPLIT_PIX = 2048;                    // pixels per workgroup (= per statistics row) of the combine pass
int conv_ws_stats_rows(const ConvArgs &a);
bool conv_up2_eligible(const ConvArgs &a);         // shape fields + w_up2 + w_interleave decide (dry runs included)
bool conv_wino_eligible(const ConvArgs &a);        // shape fields + w_wino decide
bool conv_wup2_shape_ok(int Cout, int Cin);        // worth packing the F(2x2,2x2) image of an Upsample layer
bool conv_wup2_eligible(const ConvArgs &a);        // conv_up2_eligible + w_wup2 + that shape rule (the layer alone)
int conv2d_wup2_launch(const ConvArgs &orig, hipStream_t st, int prof_cls);   // conv_wup2.hip; called by conv2d_ws_launch
// [Cin/8][Cout/128][a][b][cout quarter][position 9][k parity][cout 32][k step]
void conv_pack_weights_wup2(const float *w, int Cout, int Cin, std::vector<float> &packed);
bool conv_wino_shape_ok(int Cout, int Cin, int ks, int stride, int interleave);   // worth packing U for this layer
int conv2d_wino_launch(const ConvArgs &a, hipStream_t st);
int conv_wino_split(const ConvArgs &a);            // K slices conv_wino2 would cut a K-split layer into (0: it cannot)
bool conv_wino2_eligible(const ConvArgs &a);       // ... of those, the layers the round-4 kernel takes (whole 128-cout tiles, 16-channel chunks)
int conv2d_wino2_launch(const ConvArgs &prepared, hipStream_t st);   // conv_wino2.hip; called by conv2d_wino_launch
bool conv_wino3_eligible(const ConvArgs &prepared);                    // conv_wino3.hip: option conv_bf16x3 + conv_wino2's whole layers
int conv2d_wino3_launch(const ConvArgs &prepared, hipStream_t st);   // ... called by conv2d_wino_launch in conv_wino2's place
// [Cin/8][Cout/64][xi 16][cout half][k parity][cout 32][k step] = the LDS image of one (chunk, cout tile)
void conv_pack_weights_wino(const float *w, int Cout, int Cin, std::vector<float> &packed);
bool conv_ws_planar_ok(const ConvArgs &a);
bool conv_direct_skip_ok(const ConvArgs &a);       // this 3x3 layer can carry the block's 1x1 shortcut as extra K chunks
bool conv_direct_up2_eligible(const ConvArgs &a);  // narrow Upsample layers: the same parity form inside conv_direct (NCHW output)
bool conv_planar_ok(const ConvArgs &a);            // the kernel this convolution runs on can read x1 parity-planar
// [4 parities][Cin_pad][2x2][cout_pad], each parity packed like a ks = 2 convolution of the same interleave
void conv_pack_weights_up2(const float *w, int Cout, int Cin, int interleave, std::vector<float> &packed);
// parity-planar [B*C][2][2][H/2][W/2] -> NCHW [B*C][H][W]
int planar_to_linear_launch(const float *src, float *dst, long planes, int H, int W, hipStream_t st);
int conv_ws_split(const ConvArgs &a);
bool conv_pw_eligible(const ConvArgs &a);          // wide 1x1 layers: the barrier-free pointwise kernel (conv_pw.hip) takes THIS launch
bool conv_pw_layer_ok(const ConvArgs &a);          // ... could take the layer (shape rule)
bool conv_pw_stats_layer(const ConvArgs &a);       // ... takes it when the layer leaves fused statistics (a rule of the layer alone)
int conv_pw_stats_rows(const ConvArgs &a);
int conv2d_pw_launch(const ConvArgs &a, hipStream_t st);
int conv_direct_stats_rows(const ConvArgs &a);
size_t gn_partials_bytes(int B, int groups);
int gn_stats_launch(const GnArgs &a, hipStream_t st);

// scratch: attention_scratch_floats() floats (the partial outputs of the key-slice split; null: never split)
int attention_launch(const float *qkv, float *out, int B, int heads, int d, int T, hipStream_t st, float *scratch = nullptr);
size_t attention_scratch_floats(int B, int heads, int d, int T);

// time embedding: emb = Linear(SiLU(Linear(sinusoid(t)))) ; out = SiLU(emb)  (Model/model.py:14-32,218-222,105-108)
int temb_launch(const float *freqs, int mc, int t, const float *w0, const float *b0, const float *w2, const float *b2,
                float *tmp, float *silu_emb, hipStream_t st);
// y[r] = base[r] + dot(W[r,:], v) + b[r]  for r < rows, K columns
int gemv_bias_launch(const float *W, const float *b, const float *base, const float *v, float *y, int rows, int K,
                     hipStream_t st);

}  // namespace ipdm
