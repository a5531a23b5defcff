/*
 * vd3d.h -- C-ABI of libvd3d_hip.so: the MI355X (gfx950) native operator set behind the visualDet3D
 * detector forward path (YOLOStereo3D / GroundAware-Mono3D / KM3D).
 *
 * Boundary rules:
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless the name says host;
 *   - activations are NHWC ("channels-last"), element type selected by `dtype` (VD3D_BF16 / VD3D_F32);
 *   - every entry point takes the hipStream_t to launch on (as void*) and returns 0 or a negative
 *     VD3D_E* code -- never exit()/printf like the reference's extensions do
 *     (lib/ops/iou3d/src/iou3d.cpp:13-21, lib/ops/dcn/src/cuda/deform_conv_cuda_kernel.cu:272-276);
 *   - nothing here allocates: callers own every buffer (the reference allocates temporaries inside the
 *     extension: deform_conv_cuda.cpp:527,533; iou3d.cpp:87,98).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference root,
 * visualDet3D/networks/...).
 */
#ifndef VD3D_H
#define VD3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VD3D_BF16 0
#define VD3D_F16 2   /* IEEE half storage (same kernels and MFMA rate as bf16; conv, elementwise, DCN and the KM3D head) */
#define VD3D_F32 1

#define VD3D_OK 0
#define VD3D_EINVAL (-1)   /* bad argument (shape / alignment / dtype) */
#define VD3D_ELAUNCH (-2)  /* HIP launch error (hipGetLastError) */
#define VD3D_ERANGE (-3)   /* tensor too large for 32-bit element offsets */

/* ABI version, bumped on any signature change. */
int vd3d_abi_version(void);
/* Human-readable text of the last HIP error seen by this thread (host pointer, never NULL). */
const char* vd3d_last_error(void);
/* sha256 (first 16 hex digits) of the sources this library was built from (visualdet3d_amd/build.py source_hash): lets a measurement
 * record say which kernels it measured -- bench.py marks profiles/*_pmc_traffic.json stale when the convolution sources moved since. */
const char* vd3d_source_hash(void);

/* ------------------------------------------------------------------------------------------------
 * Fused implicit-GEMM convolution:  out = act( conv(in, W) * scale + shift (+ residual) )
 * Replaces nn.Conv2d + BatchNorm2d(eval) + ReLU (+ residual add) chains:
 *   backbones/resnet.py:23-52,55-91 (BasicBlock / Bottleneck), lib/blocks.py:24-43 (ConvBnReLU),
 *   lib/ghost_module.py:27-32 (primary conv), heads/detection_3d_head.py:54-79,508-530 (head towers),
 *   lib/PSM_cost_volume.py:29-33 (1x1 down-sample).
 * GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = kh*kw*Cin (tap-major, channel-minor).
 * MFMA: v_mfma_f32_32x32x16_bf16 (VD3D_BF16) or v_mfma_f32_32x32x2_f32 (VD3D_F32), fp32 accumulate.
 */
typedef struct vd3d_conv_params {
    const void* in;        /* [B][H][W][>=Cin] activations; Cin contiguous elements per (pixel, tap)   */
    const void* weight;    /* packed [CoutPad][Kpad], K index = (ky*kw + kx)*Cin + c, zero padded      */
    const float* scale;    /* [Cout] or NULL (== 1)                                                    */
    const float* shift;    /* [Cout] or NULL (== 0)                                                    */
    const void* residual;  /* [M][>=Cout] same dtype as `in`, or NULL                                  */
    void* out;             /* [M][>=Cout]                                                              */
    int32_t B, H, W, Cin;  /* input geometry (H, W = bounds for the zero padding test)                */
    int32_t in_pix_stride; /* elements between horizontally adjacent input pixels                      */
    int32_t in_row_stride; /* elements between input rows                                              */
    int64_t in_batch_stride;
    int64_t in_bytes;      /* size in bytes of the allocation `in` points into, from `in` (bounds)    */
    int32_t Ho, Wo, Cout;
    int32_t out_pix_stride; /* elements between consecutive output pixels (dense [M] pixel order)      */
    int32_t res_pix_stride;
    int32_t kh, kw, stride, pad, dil;
    int32_t Kpad;          /* padded K of the packed weight (multiple of 128 bytes / element size)    */
    int32_t CoutPad;       /* rows of the packed weight (multiple of 128)                              */
    int32_t relu;          /* 1: ReLU epilogue                                                         */
    int32_t dtype;         /* VD3D_BF16 | VD3D_F32: element type of in / weight / residual             */
    int32_t out_f32;       /* 1: write `out` as fp32 even when dtype is bf16 (final head convs)        */
    /* ABI >= 2.  Optional second copy of the SAME weights as an MFMA register image, used by the kernels that keep their
     * weights resident in registers (3x3 / stride 1 / pad 1, bf16, Cin = 64 | 128 | 256, Cout % 32 == 0); NULL: those
     * kernels are not selected.  Layout: [Cout/32][Cin/64][36][64 lanes][8 elements]: block (nb, kc, f = tap*4 + ks) is the
     * 1 KiB A-operand fragment of output channels 32*nb .. +31 for k = tap*Cin + 64*kc + 16*ks .. +15: lane l holds channel
     * 32*nb + (l & 31), k = tap*Cin + 64*kc + (2*ks + (l >> 5))*8 .. +7 -- so the once-per-launch weight load of a wave is 36
     * (or 72) fully coalesced 1 KiB reads instead of 64 scattered 16-byte reads per instruction.
     * 1x1 / stride 1 / pad 0 convolutions with Cin = 64 | 128 | 256 and Cout % 256 == 0 (point-wise streaming kernel) take the same
     * image with ONE tap: [Cout/32][Cin/64][4][64 lanes][8 elements].
     * ABI >= 4: the narrow-output streaming kernel (3x3 / stride 1 / pad 1, Cout = 32 rows incl. zero filters, ANY Cin % 64 == 0 >= 128:
     * the DCN offset convs) streams the same image through LDS one 64-channel block [36][64 lanes][8] at a time. */
    const void* weight_frag;
    /* ABI >= 5.  Optional scratch for the split-K path of low-parallelism shapes (a batch-1 call: M = B*Ho*Wo gives fewer 128-row
     * tiles than the device has CUs while K = kh*kw*Cin is thousands deep): tiles x splits workgroups park fp32 partial tiles here and
     * a second launch adds them in split order and runs the epilogue -- deterministic, no atomics.  NULL (or fewer bytes than
     * vd3d_conv2d_workspace_bytes asks for): the convolution runs unsplit.  16-byte aligned device memory, contents undefined on
     * entry and exit; the library never allocates.  16-bit formats only: in VD3D_F32 (validation mode) the summation order must not
     * depend on the batch size, the field is ignored and vd3d_conv2d_workspace_bytes returns 0. */
    void* splitk_ws;
    int64_t splitk_ws_bytes;
} vd3d_conv_params;

int vd3d_conv2d_igemm(const vd3d_conv_params* p, void* stream);
/* Bytes of `splitk_ws` with which vd3d_conv2d_igemm(p) would split K (0: this shape does not split; -1: invalid parameters).
 * Depends on the shape fields, dtype and weight_frag only. */
int64_t vd3d_conv2d_workspace_bytes(const vd3d_conv_params* p);
/* ids of the tiles the dispatch heuristic can select -> ids[0..cap); returns how many there are (every one of them is forced
 * against the oracle by tests/test_conv_tiles_gpu.py through the test hook of csrc/test_hooks.h, which is not part of this ABI). */
int vd3d_conv2d_production_tiles(int32_t* ids, int cap);

/* Test-time image pipeline for ONE frame, fed from uint8 (data/pipeline/stereo_augmentator.py: ConvertToFloat :30-36,
 * CropTop :214-249, Resize :62-134 = cv2.resize INTER_LINEAR on float32 + crop / zero-pad to the network width, Normalize
 * :39-59) fused with the collate_fn's HWC -> CHW float cast (data/kitti/dataset/stereo_dataset.py:141-157).
 *   src_hwc: [Hs][Ws][3] uint8 (host-visible or device memory reachable by the GPU: pass a device pointer);
 *   (Hr, Wr): size after Resize (the host rounds like stereo_augmentator.py:78-80); (H, W): network input (Hr >= H);
 *   out_nchw: fp32 [3][H][W] (the reference's network input) or NULL; out_packed: bf16 [H+6][W+8][4] zero-bordered NHWC4
 *   (what vd3d_stem_conv_pool / the stem conv read) or NULL; mean3 / std3: host pointers to 3 floats. */
int vd3d_preprocess_image(const uint8_t* src_hwc, int Hs, int Ws, int crop_top, int Hr, int Wr, float* out_nchw,
                          void* out_packed, int H, int W, const float* mean3, const float* std3, void* stream);

/* Stem input packing: NCHW fp32 image -> zero-bordered NHWC4 (3 channels + 1 zero) so that one kernel row of
 * the 7x7/s2 stem (backbones/resnet.py:118, conv1) is 8 px * 4 ch = 32 contiguous elements.
 * out: [B][H+2*pad_y][W+pad_l+pad_r][4] of `dtype`; borders are written as zeros. */
int vd3d_pack_image_nhwc4(const float* in_nchw, void* out, int B, int H, int W,
                          int pad_y, int pad_l, int pad_r, int dtype, void* stream);

/* 7x7 / stride 1 / pad 3 convolution of a 3-channel fp32 NCHW image to Cout <= 16 channels + folded BN + ReLU, 16-bit NHWC
 * output: the DLA base layer (backbones/dla.py:116-117) without a packed copy of the image (the halo is converted in LDS).
 * weight_frag: [7][64][8] elements of `dtype` (VD3D_BF16 | VD3D_F16), the MFMA operand image
 *   weight_frag[ky][l][e] = w[o = l & 15][c = e & 3][ky][kx = 2 * (l >> 4) + (e >> 2)]   (0 where c = 3, kx = 7 or o >= Cout).
 * out: [B][H][W][out_pix_stride] (out_pix_stride % 4 == 0, 8-byte aligned); scale / shift: fp32[Cout] or NULL. */
int vd3d_image_conv7x7(const float* img_nchw, const void* weight_frag, const float* scale, const float* shift, void* out,
                       int B, int H, int W, int Cout, int out_pix_stride, int relu, int dtype, void* stream);

/* General form of the image packer: `cpad` = 4 or 8 channels per pixel (3 real + zeros), independent borders.
 * Used by DLA's 7x7 stride-1 base layer (backbones/dla.py:247-251): with 8-channel bf16 pixels every pixel is 16 bytes,
 * so a kernel row (8 px x 8 ch = 64 elements) is a 16-byte aligned run for any output column. */
int vd3d_pack_image_nhwc(const float* in_nchw, void* out, int B, int H, int W, int pad_y0, int pad_y1, int pad_l, int pad_r,
                         int cpad, int dtype, void* stream);

/* nn.MaxPool2d(2, stride 2) on NHWC (backbones/dla.py:213, Tree.downsample). */
int vd3d_maxpool2x2(const void* in, void* out, int B, int H, int W, int C, int in_pix_stride, int out_pix_stride,
                    int dtype, void* stream);
/* Fused ResNet stem (backbones/resnet.py:118-121,187-190): 7x7/s2/p3 conv + eval BatchNorm + ReLU + MaxPool2d(3, 2, 1) in one
 * pass, bf16.  packed: output of vd3d_pack_image_nhwc4 with pads (3, 3, 5) = [B][H+6][W+8][4] bf16; weight: the stem's packed
 * panel [CoutPad][Kpad] bf16 with k = ky*32 + kx*4 + c (kx = 7 and c = 3 zero); scale/shift: fp32[64] folded BN (may be NULL);
 * out: [B][H/4][W/4][out_pix_stride >= 64] bf16.  Needs 64 output channels, H/4 % 8 == 0, W/4 % 16 == 0. */
int vd3d_stem_conv_pool(const void* packed, const void* weight, const float* scale, const float* shift, void* out,
                        int B, int H, int W, int Kpad, int out_pix_stride, void* stream);
/* The same fused stem reading the fp32 NCHW image(s) -- the reference's network input -- directly: batch entries [0, B0) come from img0
 * ([B0][3][H][W]), [B0, B0 + B1) from img1 (stereo: left, right; B1 = 0: one tensor).  No packed copy, no vd3d_pack_image_nhwc4 launch;
 * results are bit-identical to packing first (the bf16 rounding of the image is the same single rounding). */
int vd3d_stem_conv_pool_f32(const float* img0, int B0, const float* img1, int B1, const void* weight, const float* scale,
                            const float* shift, void* out, int H, int W, int Kpad, int out_pix_stride, void* stream);

/* Depth-wise nn.ConvTranspose2d(C, C, 2f, stride f, padding f/2, groups C, bias False) on NHWC (backbones/dla_utils.py:69-71)
 * fused with the `+ layers[i-1]` that follows it (:83); weight [(2f)^2][C] fp32; add may be NULL. */
int vd3d_dwconv_transpose(const void* in, const float* weight, const void* add, void* out, int B, int H, int W, int C,
                          int f, int in_pix_stride, int add_pix_stride, int out_pix_stride, int dtype, void* stream);

/* nn.MaxPool2d(3, stride 2, pad 1) on NHWC (backbones/resnet.py:121,193). */
int vd3d_maxpool3x3s2(const void* in, void* out, int B, int H, int W, int C,
                      int in_pix_stride, int out_pix_stride, int dtype, void* stream);
/* nn.AvgPool2d(2) on NHWC (detectors/yolostereo3d_core.py:25,35). */
int vd3d_avgpool2x2(const void* in, void* out, int B, int H, int W, int C,
                    int in_pix_stride, int out_pix_stride, int dtype, void* stream);
/* Depth-wise 3x3 (pad 1) + folded BN + ReLU on NHWC: lib/ghost_module.py:34-38 (cheap_operation).
 * weight: [9][C] fp32 tap-major; scale/shift: [C] fp32. */
int vd3d_dwconv3x3(const void* in, const float* weight, const float* scale, const float* shift, void* out,
                   int B, int H, int W, int C, int in_pix_stride, int out_pix_stride, int relu,
                   int dtype, void* stream);
/* Strided NHWC channel-slice copy (torch.cat replacement when a producer cannot write in place). */
int vd3d_copy_channels(const void* in, void* out, int64_t n_pix, int C, int in_pix_stride, int out_pix_stride,
                       int dtype, void* stream);
/* NHWC (dtype) -> NCHW fp32 and back: only at the module boundary (state taps, tests). */
int vd3d_nhwc_to_nchw_f32(const void* in, float* out, int B, int H, int W, int C, int in_pix_stride,
                          int dtype, void* stream);
int vd3d_nchw_f32_to_nhwc(const float* in, void* out, int B, int H, int W, int C, int out_pix_stride,
                          int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stereo cost volumes.
 * PSMCosineModule.forward (lib/PSM_cost_volume.py:81-96):
 *   cost[b,y,x,d] = (1/C) * sum_c L[b,y,x,c] * R[b,y,x-d,c]   for x >= d, else 0;   d in [0, D)
 * left/right: NHWC [B][H][W][C]; cost: NHWC [B][H][W][D] written with out_pix_stride (concat slice). */
int vd3d_psm_cosine(const void* left, const void* right, void* cost, int B, int H, int W, int C, int D,
                    int in_pix_stride, int out_pix_stride, int dtype, void* stream);
/* CostVolume.forward concat-volume build (lib/PSM_cost_volume.py:49-64), channels-last:
 *   vol[b,d,y,x,0:F]  = L[b,y,x,:] (x >= d else 0);  vol[b,d,y,x,F:2F] = R[b,y,x-d,:] (x >= d else 0). */
int vd3d_costvol_build(const void* left, const void* right, void* vol, int B, int H, int W, int F, int D,
                       int in_pix_stride, int dtype, void* stream);
/* CostVolume.forward after its 1x1 down-sample, in ONE launch (lib/PSM_cost_volume.py:45-68): concat volume (never materialised) ->
 * Conv3d(2F -> F) + BN3d + ReLU -> Conv3d(F -> F) + BN3d + ReLU -> reshape to NHWC [B][H][W][F*D] (channel = f*D + d, written with
 * out_pix_stride).  bf16 only, F = 8 (PSM_features of every shipped config), D <= 24; the intermediate volume is rounded to bf16
 * exactly where vd3d_conv3d_3x3x3 rounds it.  left / right: NHWC [B][H][W][F] (in_pix_stride); w1 [27][2F][F], w2 [27][F][F] fp32. */
int vd3d_cost_volume_fused(const void* left, const void* right, const float* w1, const float* scale1, const float* shift1,
                           const float* w2, const float* scale2, const float* shift2, void* out, int B, int H, int W, int F, int D,
                           int in_pix_stride, int out_pix_stride, int dtype, void* stream);
/* Conv3d(3x3x3, pad 1) + folded BN3d + ReLU, channels-last [B][D][H][W][Cin] (lib/PSM_cost_volume.py:34-41).
 * weight: [27][Cin][Cout] fp32.  If out_fd_major != 0 the result is written as NHWC [B][H][W][Cout*D] with
 * channel = f*D + d (the reshape at PSM_cost_volume.py:66-67) using out_pix_stride. */
int vd3d_conv3d_3x3x3(const void* in, const float* weight, const float* scale, const float* shift, void* out,
                      int B, int D, int H, int W, int Cin, int Cout, int relu, int out_fd_major,
                      int out_pix_stride, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Anchor head post-processing (heads/detection_3d_head.py:341-400 get_bboxes, :218-263 _decode,
 * networks/utils/utils.py:186-196 ClipBoxes, heads/anchors.py:99-111 ground filter, torchvision nms).
 * One launch covers B samples; per-sample results are identical to the reference's batch-1 path.
 *   cls   [B][N][n_cls+1] fp32 logits (last = alpha bin), reg [B][N][12] fp32
 *   anchors [N][4] fp32; prior_mean_std [A][types][6][2] fp32 (A = anchors per cell, N = cells*A)
 *   P2 [B][3][4] fp32
 * Outputs (padded to max_det per sample, decreasing score == torchvision nms order):
 *   out_scores [B][max_det], out_boxes [B][max_det][11], out_labels [B][max_det] int32,
 *   out_anchor [B][max_det] int32 (flat anchor index), out_count [B] int32
 * workspace: see vd3d_head_workspace_bytes.  max_cand: a power of two.  If more than max_cand anchors pass the score threshold in a
 * sample, out_count[b] = -1 (more than max_det survivors: -2) and the caller repeats the call for that sample with a larger capacity -- the
 * reference's candidate list has no cap (detection_3d_head.py:341-400).  Capacities <= 8192 keep the per-frame lists in LDS; larger ones
 * (up to every anchor of the frame) run the same steps on lists in `workspace`, one workgroup per frame: correct at any count, not fast. */
typedef struct vd3d_head_params {
    const float* cls;
    const float* reg;
    const float* anchors;
    const float* prior_mean_std;
    const float* P2;
    int32_t B, N, A, n_cls, n_types;
    int32_t img_h, img_w;      /* ClipBoxes bounds; img_w <= 0: no clipping (the reference's `img_batch is None`) */
    float score_thr, nms_iou_thr;
    float filter_y_min, filter_y_max, filter_x_max;
    int32_t use_filter;
    int32_t max_cand, max_det;
    void* workspace;
    float* out_scores;
    float* out_boxes;
    int32_t* out_labels;
    int32_t* out_anchor;
    int32_t* out_count;
} vd3d_head_params;
int64_t vd3d_head_workspace_bytes(int B, int max_cand);
int vd3d_head_postprocess(const vd3d_head_params* p, void* stream);
/* The two stages of vd3d_head_postprocess as separate entries (ABI >= 3).  vd3d_head_select: ground filter + sigmoid + threshold ->
 * candidate lists in `workspace`; reads only `cls` (reg / out_* may be NULL), so it can run on the stream of the cls tower while the reg
 * tower is still computing.  vd3d_head_nms: decode + clip + z-prior filter + NMS over those lists; same parameters, stream-ordered
 * after the select.  vd3d_head_postprocess == select then nms on one stream. */
int vd3d_head_select(const vd3d_head_params* p, void* stream);
int vd3d_head_nms(const vd3d_head_params* p, void* stream);

/* Multi-GPU result record (the only cross-rank traffic is the gather of the padded detections): packs the
 * padded outputs of vd3d_head_postprocess / vd3d_km3d_decode of B frames into ONE contiguous fp32 block
 *   pack[B][k + 1][13]:  rows 0..k-1 = (score, 11 box fields, label) of detection r (zeros past the frame's count),
 *                        row k = (count, 0 ...) -- the count rides in the block, negative overflow markers included,
 * so a step needs a single all_gather (and a single device->host copy) and the pack is one launch that can be captured
 * in the step's hipGraph.  scores [B][K], boxes [B][K][11], labels [B][K] int32, count [B] int32, k <= K. */
int vd3d_pack_detections(const float* scores, const float* boxes, const int32_t* labels, const int32_t* count, int B, int K,
                         int k, float* pack, void* stream);

/* torchvision.ops.nms replacement (call sites heads/detection_3d_head.py:386, heads/km3d_head.py:303):
 * boxes [n][4] fp32, scores [n] fp32 -> keep [n] int32 (indices in decreasing-score order), count [1]. */
int vd3d_nms(const float* boxes, const float* scores, int n, float iou_thr, int32_t* keep, int32_t* count,
             void* workspace, void* stream);
int64_t vd3d_nms_workspace_bytes(int n);

/* ------------------------------------------------------------------------------------------------
 * iou3d (lib/ops/iou3d/src/iou3d.cpp:174-179 pybind surface; kernels iou3d_kernel.cu:223-348).
 * boxes are BEV (x1,y1,x2,y2,ry) fp32.  nms variants take boxes pre-sorted by score and write the kept
 * indices + count on the DEVICE (the reference copies the mask to the host and scans there). */
int vd3d_boxes_overlap_bev(const float* boxes_a, int na, const float* boxes_b, int nb, float* out, void* stream);
int vd3d_boxes_iou_bev(const float* boxes_a, int na, const float* boxes_b, int nb, float* out, void* stream);
int vd3d_nms_bev(const float* boxes, int n, float thr, int normal, int32_t* keep, int32_t* count,
                 void* workspace, void* stream);
int64_t vd3d_nms_bev_workspace_bytes(int n);

/* ------------------------------------------------------------------------------------------------
 * Deformable convolution forward, v1 (mask == NULL) and v2 / modulated
 * (lib/ops/dcn/src/deform_conv_ext.cpp:149-163; host deform_conv_cuda.cpp:152-260,491-570; kernels
 * deform_conv_cuda_kernel.cu:190-243, :570-633).  The bilinear-sampled columns are never materialised in HBM.
 *
 * vd3d_deform_conv_forward: the reference extension's tensors, NCHW fp32 contiguous:
 *   input [B][C][H][W], offset [B][dg*2*kh*kw][Ho][Wo] (y,x interleaved per tap), mask [B][dg*kh*kw][Ho][Wo] or NULL,
 *   weight [O][C/g][kh][kw], bias [O] or NULL, output [B][O][Ho][Wo];  workspace: vd3d_deform_conv_workspace_bytes. */
int64_t vd3d_deform_conv_workspace_bytes(int O, int C, int groups, int kh, int kw);
int vd3d_deform_conv_forward(const float* input, const float* weight, const float* bias, const float* offset,
                             const float* mask, float* output, void* workspace, int B, int C, int H, int W, int O,
                             int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                             int dil_h, int dil_w, int groups, int deformable_groups, void* stream);

/* General form used by the engine: any layout through element strides {batch, channel, y, x}, bf16 or fp32
 * activations, pre-packed weight (vd3d_dcn_pack_weight: [O][Kpad], K = tap*C/g + c), fused epilogue
 * y = relu?((acc + bias) * scale + shift), optional in-kernel sigmoid of the modulation logits. */
typedef struct vd3d_dcn_params {
    const void* in; const void* weight; const float* bias; const float* scale; const float* shift;
    const float* offset; const float* mask; void* out;
    int32_t B, C, H, W, O, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
    int32_t groups, deformable_groups, Kpad, dtype, mask_sigmoid, relu;
    int64_t in_strides[4], offset_strides[4], mask_strides[4], out_strides[4];
} vd3d_dcn_params;
int vd3d_dcn_pack_weight(const float* w_oihw, void* packed, int O, int Cg, int kh, int kw, int Kpad, int dtype, void* stream);
int vd3d_deform_conv(const vd3d_dcn_params* p, void* stream);
/* Sampling half of a deformable convolution with MANY output channels (the 2176 -> 2176 DCNv2 of the stereo base head): writes the
 * bilinear-sampled, modulated columns ONCE, in the activation dtype (the rounding point of the fused kernel), as
 * columns[B][Ho][Wo][kh*kw][C]; the contraction then runs as a 1x1 vd3d_conv2d_igemm over K = kh*kw*C (weight packed tap-major,
 * bias / BN / ReLU in its epilogue).  The reference materialises the same matrix in fp32 (deform_conv_cuda.cpp:531-569).
 * Channel-contiguous (NHWC) input, groups = deformable_groups = 1; weight / out / bias / scale / shift of `p` are ignored. */
int vd3d_deform_columns(const vd3d_dcn_params* p, void* columns, void* stream);

/* LookGround sampling (lib/look_ground.py:24-69): builds [x ; prior disparity] and bilinear-samples it
 * (grid_sample, border padding, align_corners=True) at (x, y + y_shift), y_shift = geometric prior + 0.1*tanh(disp).
 *   x [B][H][W][C] NHWC, disp [B][H][W] fp32 = raw output of the disp_create conv (tanh applied here),
 *   P2 [B][3][4] full-resolution calibration (the /16 of look_ground.py:31 is applied here)
 *   out [B][H][W][round_up(C+1, 16 bytes)]: channels [0,C) = sampled x, channel C = sampled prior, rest 0. */
int vd3d_look_ground_sample(const void* x, const float* disp, const float* P2s, void* out, int B, int H, int W,
                            int C, int in_pix_stride, int out_pix_stride, float baseline, float elevation,
                            int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Hill-climbing yaw post-optimisation for a padded batch of detections (heads/detection_3d_head.py:294-308
 * _post_process; lib/fast_utils/hill_climbing.py:7-122 post_opt/hill_climb/test_projection; fast_utils/bbox3d.py:19-82;
 * fast_utils/bbox2d.py:39-66; utils/utils.py:30-45).  fp64 per box, one lane per detection, no host round trip.
 *   boxes  [B][cap][11] fp32 (x1,y1,x2,y2,cx,cy,z,w,h,l,alpha): alpha is rewritten in place for the boxes with
 *          z > min_depth (reference: 3) and label == target_label (reference: 0);
 *   labels [B][cap] int32; counts [B] int32 valid detections per sample (NULL: all cap; negative: none);
 *   P2s    [B][3][4] fp32; clamp_w/clamp_h: the projection clamp the reference hard-codes to 1280 x 288
 *          (hill_climbing.py:111-113). */
int vd3d_post_opt(float* boxes, const int32_t* labels, const int32_t* counts, const float* P2s, int B, int cap,
                  float clamp_w, float clamp_h, float min_depth, int target_label, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rotated-rectangle IoU of the KITTI AP evaluator (evaluator/kitti/rotate_iou.py:261-328 rotate_iou_gpu_eval, a numba.cuda
 * kernel in the reference; callers evaluator/kitti/eval.py:124,173).  boxes [N][5], query_boxes [K][5] = (cx, cy, dx, dy, angle)
 * fp32; iou [N][K] fp32.  criterion -1: inter / union, 0: inter / area(query), 1: inter / area(box), other: the intersection
 * area (rotate_iou.py:245-258; note the kernel passes the QUERY box first, :290-291). */
int vd3d_rotate_iou_eval(const float* boxes, const float* query_boxes, int N, int K, int criterion, float* iou, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Post-path geometry of pipelines/evaluators.py:112-129 (test_one) for a padded batch of detections, one launch:
 * BackProjection (networks/utils/utils.py:262-278), theta = alpha2theta_3d (utils/utils.py:47-62), 2D-box shift + rescale to
 * the original image (evaluators.py:118-127), bottom-centre y (data/kitti/utils.py:180-182).
 *   boxes [B][cap][11] fp32 (x1,y1,x2,y2,cx,cy,z,w,h,l,alpha); counts [B] int32 or NULL; P2s [B][3][4] (network-input
 *   calibration); xform [B][4] = shift_left, shift_top, scale_x, scale_y (computed by the host in fp64 as evaluators.py:118-122);
 *   out [B][cap][12] fp32 = x1,y1,x2,y2 (original image), x3d, y_bottom, z, w, h, l, alpha, theta (rows >= count are zero). */
int vd3d_kitti_postpath(const float* boxes, const int32_t* counts, const float* P2s, const float* xform, float* out,
                        int B, int cap, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Two chained 3x3 convolutions in one launch: `a` (stride 1, pad 1, 16 -> 16 channels) + folded BN + ReLU, then `b` (stride 2, pad 1,
 * 16 -> <= 32 channels) + folded BN + ReLU -- DLA level0 -> level1 (backbones/dla.py:118-121, :142-148 `_make_conv_level`; called
 * from DLA.forward :150-158).  Both are the vd3d_conv2d_igemm parameter blocks of the two convolutions (16-bit formats, packed
 * weights, scale / shift, relu flags); `a->out` and `b->in` are ignored: the intermediate tensor (the rounding point of the unfused
 * path) lives in LDS only.  b's geometry must continue a's (b->B/H/W == a->B/Ho/Wo).  Results are bit-identical to the two calls. */
int vd3d_conv2d_pair(const vd3d_conv_params* a, const vd3d_conv_params* b, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ABI >= 6.  A whole ResNet Bottleneck of the 64-wide stage in one launch (backbones/resnet.py:55-91 `Bottleneck.forward`: conv1 1x1 + bn1 + relu ->
 * conv2 3x3 + bn2 + relu -> conv3 1x1 + bn3, + identity | downsample(x), relu; ResNet-50 / 101 / 152 layer1).  c1 / c2 / c3 / ds are the
 * vd3d_conv2d_igemm parameter blocks of the three convolutions and of the stage's first block's 1x1 downsample conv (NULL for an identity block,
 * whose c3->residual must be c1->in): 16-bit formats, conv1 256 | 64 -> 64, conv2 64 -> 64, conv3 (and ds) 64 -> 256, stride 1.  c1->in is the
 * block's input, c3->out its output; every other in / out pointer is ignored -- both 64-channel intermediates live in LDS only (rounded to the storage
 * type exactly where the separate launches round them).  Other shapes: VD3D_EINVAL (the caller launches the convolutions separately). */
int vd3d_conv2d_bottleneck(const vd3d_conv_params* c1, const vd3d_conv_params* c2, const vd3d_conv_params* c3, const vd3d_conv_params* ds,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * KM3D head, fused (heads/km3d_head.py:132-153,353-357: nine branches conv3x3(64 -> 256) + ReLU + conv1x1(256 -> n_h)).  The nine
 * 3x3 convs run as ONE implicit GEMM with Cout = 256 x heads (`p`: the vd3d_conv2d_igemm parameters of that conv, bf16,
 * shift = the concatenated first-conv biases, relu implied, `p->out` ignored); each 256-pixel x 256-channel tile applies bias +
 * ReLU, rounds to bf16 and multiplies by its head's 1x1 weights in the epilogue, so the 9 x 256-channel intermediate never
 * reaches HBM.  w2_packed: [heads][32][256] bf16 (rows >= n_out[h] zero), b2: [heads][32] fp32, outs[h]: fp32 [B*H*W][n_out[h]]
 * (host array of device pointers), n_out[h] <= 32, heads <= 9. */
int vd3d_km3d_head_fused(const vd3d_conv_params* p, const void* w2_packed, const float* b2, void* const* outs,
                         const int32_t* n_out, int n_heads, void* stream);

/* ------------------------------------------------------------------------------------------------
 * KM3D / RTM3D keypoint-head decoding (heads/km3d_head.py:155-314 _decode + get_bboxes; networks/utils/rtm3d_utils.py
 * _nms :122-127, _topk :201-216, _topk_channel :219-228, gen_position :314-455; torchvision nms) for a whole batch.
 * All maps are fp32 NHWC logits / regressions [B][H][W][n] (n: hm n_cls, wh 2, hps 18, rot 8, dim 3, prob 1, reg 2,
 * hm_hp 9, hp_offset 2); P2 [B][3][4]; kconst = the head's `const` buffer, 32 floats ([16][2]).
 * Outputs padded to K per sample in decreasing-score order: scores [B][K], boxes [B][K][11]
 * (x1,y1,x2,y2,cx,cy,z,w,h,l,alpha), cls [B][K] int32, count [B] int32 (-1: a heat-map channel has more local maxima above its threshold than
 * max_peaks -- the caller repeats the sample with a larger capacity: the reference's top-K has no cap, rtm3d_utils.py:201-228; capacities <= 8192
 * sort in LDS, larger ones -- up to every pixel of the map -- in `workspace`).
 * Limits (checked before anything is enqueued): n_joints == 9, K <= 128, n_cls * K <= 512, max_peaks a power of two in [K, 2^24]
 * (VD3D_EINVAL); n_cls <= 9 heat-map channels per map -- the tiled peaks kernel keeps all channels of a map per workgroup
 * (VD3D_ERANGE). */
typedef struct vd3d_km3d_params {
    const float *hm, *wh, *hps, *rot, *dim, *prob, *reg, *hm_hp, *hp_offset, *P2, *kconst;
    int32_t B, H, W, n_cls, n_joints, K, max_peaks, img_h, img_w;
    float score_thr, nms_iou_thr;
    void* workspace;
    float* out_scores; float* out_boxes; int32_t* out_cls; int32_t* out_count;
} vd3d_km3d_params;
int64_t vd3d_km3d_workspace_bytes(int B, int n_cls, int n_joints, int max_peaks, int K);
int vd3d_km3d_decode(const vd3d_km3d_params* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VD3D_H */
