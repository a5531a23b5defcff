// conv_common.h -- shared between the implicit-GEMM translation units (conv_igemm.hip: tile / halo kernels and the dispatcher;
// conv_resident.hip: the register-resident-weight kernels).
#pragma once
#include "common.h"
#include <utility>

namespace vd3d_conv {

struct ConvArgs {
    const char* in;
    const char* weight;
    const char* wfrag = nullptr;      // optional MFMA register image of the weights (vd3d_conv_params.weight_frag)
    const float* scale;
    const float* shift;
    const char* residual;
    char* out;
    int B, H, W, Cin;
    int in_pix_stride, in_row_stride;
    int64_t in_batch_stride;
    uint32_t in_bytes, w_bytes;
    int Ho, Wo, Cout;
    int out_pix_stride, res_pix_stride;
    int kh, kw, stride, pad, dil;
    int Kpad, relu, out_f32;
    int M, tiles_m, tiles_n, ntaps, nk;
    FastDiv fd_howo, fd_wo;           // m / (Ho * Wo), rem / Wo of the tile kernels' pixel-row decode
    FastDiv fd_tiles_m;               // tile / tiles_m (persistent fused-head kernel)
    int vec_epilogue, wide_store, chunk_major;
    int group_m = 0;                  // > 0: DMA tile kernels walk group_m pixel tiles x all N tiles per XCD run (huge 1x1 GEMMs)
    int strip_lines = 0;              // 16x16x32 strip tiles: 16-bit output re-laid through LDS (whole pixel runs per store instruction)
    int line_store = 0;               // 16-bit output re-laid through LDS so that store instructions cover whole 128-byte lines
    // split-K (low-parallelism shapes, e.g. batch 1): workgroup (tile, split) of a KS = 1 instantiation walks K slices
    // [split * ks_per, + ks_per) and parks its raw fp32 accumulators in ks_ws in register order; splitk_reduce_kernel adds the ks_n
    // partials in split order (deterministic) and runs the epilogue.  Only those two kernels read these fields.
    float* ks_ws = nullptr;
    int64_t ks_ws_bytes = 0;
    int ks_phase = 0, ks_per = 0, ks_n = 1;
    // fused KM3D head (vd3d_km3d_head_fused): per 256-channel N tile h, a second GEMM [256 px x 256] x [256 x n_h] in the
    // epilogue; h_w2 = packed [heads][32][256] bf16, h_b2 = [heads][32] fp32, h_out[h] = fp32 [M][h_n[h]]
    const char* h_w2 = nullptr;
    const float* h_b2 = nullptr;
    float* h_out[9] = {};
    int h_n[9] = {};
};

constexpr uint32_t kOOB = 0x80000000u;  // byte offset guaranteed >= num_records (host enforces in_bytes < 2^31)

// entry points of conv_resident.hip (bf16 3x3 / stride 1 / pad 1 kernels whose weights live in registers)
bool regw_shape_ok(const ConvArgs& a);                       // Cin 128 | 256, Cout a multiple of the channel slice, weight_frag given
// fmt = VD3D_BF16 | VD3D_F16 (the 16-bit storage format of activations and weights)
int launch_regw(ConvArgs& a, hipStream_t stream, int fmt, int ring = 4, int abl = 0);     // (ring / abl != defaults: tuning build only)
int launch_resident64(ConvArgs& a, hipStream_t stream, int fmt);      // Cin = Cout = 64
bool ksplit_shape_ok(const ConvArgs& a);                               // Cin 256 3x3 / s1 / p1, Cout % 64 == 0, weight_frag given
int launch_ksplit(ConvArgs& a, hipStream_t stream, int fmt);          // 8-wave K-split resident-weight kernel (layer3)
bool small_shape_ok(const ConvArgs& a);                                // 3x3, stride 1 | 2, Cin 16 | 32 | 64, Cout <= 32, no residual
int launch_small(ConvArgs& a, hipStream_t stream, int fmt);           // small-channel streaming kernel
bool narrow_shape_ok(const ConvArgs& a);                               // 3x3 / s1 / p1, Cin % 64 == 0 (>= 128), Cout <= 32, weight_frag given
int launch_narrow(ConvArgs& a, hipStream_t stream, int fmt);          // narrow-output streaming kernel (chunked small-channel kernel)
bool pair_shape_ok(const ConvArgs& a, const ConvArgs& b);               // conv A 3x3 / s1 16 -> 16, conv B 3x3 / s2 16 -> <= 32, 16-bit
int launch_pair(ConvArgs& a, ConvArgs& b, hipStream_t stream, int fmt);   // DLA level0 + level1 in one launch
// a whole ResNet Bottleneck of the 64-wide stage in one launch (conv_bottleneck.hip): c1 1x1 (256 | 64 -> 64), c2 3x3 (64 -> 64), c3 1x1 (64 -> 256),
// cd = the 1x1 downsample conv (64 -> 256) of the stage's first block or nullptr (identity block: c3.residual == c1.in)
bool bottleneck_shape_ok(const ConvArgs& c1, const ConvArgs& c2, const ConvArgs& c3, const ConvArgs* cd);
int launch_bottleneck(ConvArgs& c1, ConvArgs& c2, ConvArgs& c3, ConvArgs* cd, hipStream_t stream, int fmt);
bool km3d_head_shape_ok(const ConvArgs& a);                            // 3x3 / s1 / p1, Cin % 64 == 0, Cout = 256 x branches (h_* fields set)
int launch_km3d_head(ConvArgs& a, hipStream_t stream, int fmt);       // persistent fused KM3D head (km3d_head_conv.hip)
bool pw_shape_ok(const ConvArgs& a);                                   // 1x1 / stride 1, Cin 64 | 128 | 256, Cout % 256 == 0, weight_frag given
int launch_pw(ConvArgs& a, hipStream_t stream, int fmt);              // point-wise expansion streaming kernel (no LDS)

}  // namespace vd3d_conv
