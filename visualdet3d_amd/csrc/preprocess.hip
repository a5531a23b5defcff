// preprocess.hip -- test-time image pipeline on the device, fed from uint8 (gfx950).
//
// Replaces, for one image, ConvertToFloat -> CropTop -> Resize -> Normalize of data/pipeline/stereo_augmentator.py
// (:30-36, :214-249, :62-134, :39-59; config/Stereo3D_example:102-107) and the HWC -> CHW transpose + float cast of the
// collate_fn (data/kitti/dataset/stereo_dataset.py:141-157).  The host pipeline moves 4 bytes per sample and channel through
// cv2 / numpy and then over PCIe; here the uint8 frame (1.4 MB) is the only upload and the kernel writes either the
// reference's network input (fp32 NCHW) or directly the zero-bordered NHWC4 bf16 image the fused stem kernel consumes.
//
// Resize = cv2.resize(float32 image, (Wr, Hr)), INTER_LINEAR: source coordinate fx = (float)((dx + 0.5) * scale - 0.5) with
// scale = 1 / ((double)Wr / Ws), sx = floor(fx), weight fx - sx; sx < 0 -> (0, weight 0); sx >= Ws - 1 -> (Ws - 1, weight
// 0); horizontal pass S[sx]*(1-fx) + S[sx+1]*fx, then vertical pass, all fp32 (OpenCV resize.cpp, HResizeLinear /
// VResizeLinear).  cv2 is a third-party dependency that is not installed here: parity against cv2 itself is UNPINNED; the
// restatement in oracle/preprocess_ref.py follows the algorithm above.  Columns >= Wr (image narrower than the network
// input) are zero BEFORE Normalize like np.pad in Resize (:103-112), columns >= W are cropped (:95-101).
#pragma clang fp contract(off)
#include "common.h"

namespace {

struct PreArgs {
    const uint8_t* src;
    float* out_nchw;
    short* out_packed;
    int Hs, Ws, crop_top, Hr, Wr, H, W;
    double scale_x, scale_y;
    float mean[3], stdv[3];
};

__device__ inline void lin_coord(int d, double scale, int n, int& s0, float& w) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
    s0 = s;
    w = f;
}

__global__ void preprocess_kernel(const PreArgs p) {
    // thread grid over the PACKED extent (H + 6) x (W + 8); the interior is the normalised image, the border is zero
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    const int py = blockIdx.y;
    const int Wp = p.W + 8, Hp = p.H + 6;
    if (px >= Wp || py >= Hp) return;
    const int x = px - 3, y = py - 3;
    const bool inside = x >= 0 && x < p.W && y >= 0 && y < p.H;
    float v[3] = {0.f, 0.f, 0.f};
    if (inside) {
        float r[3] = {0.f, 0.f, 0.f};
        if (x < p.Wr && y < p.Hr) {
            int sx, sy;
            float fx, fy;
            lin_coord(x, p.scale_x, p.Ws, sx, fx);
            lin_coord(y, p.scale_y, p.Hs - p.crop_top, sy, fy);
            const int sx1 = sx + 1 < p.Ws ? sx + 1 : sx;
            const int sy1 = sy + 1 < p.Hs - p.crop_top ? sy + 1 : sy;
            const uint8_t* r0 = p.src + ((size_t)(sy + p.crop_top) * p.Ws) * 3;
            const uint8_t* r1 = p.src + ((size_t)(sy1 + p.crop_top) * p.Ws) * 3;
            const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float h0 = (float)r0[sx * 3 + c] * a0 + (float)r0[sx1 * 3 + c] * a1;
                const float h1 = (float)r1[sx * 3 + c] * a0 + (float)r1[sx1 * 3 + c] * a1;
                r[c] = h0 * b0 + h1 * b1;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = ((r[c] / 255.0f) - p.mean[c]) / p.stdv[c];
        if (p.out_nchw) {
#pragma unroll
            for (int c = 0; c < 3; ++c) p.out_nchw[((size_t)c * p.H + y) * p.W + x] = v[c];
        }
    }
    if (p.out_packed) {
        i32x2 o;
        o[0] = (int)((uint32_t)(uint16_t)f2bf(v[0]) | ((uint32_t)(uint16_t)f2bf(v[1]) << 16));
        o[1] = (int)((uint32_t)(uint16_t)f2bf(v[2]));
        *(i32x2*)(p.out_packed + ((size_t)py * Wp + px) * 4) = o;
    }
}

}  // namespace

extern "C" int vd3d_preprocess_image(const uint8_t* src_hwc, int Hs, int Ws, int crop_top, int Hr, int Wr, float* out_nchw,
                                     void* out_packed, int H, int W, const float* mean3, const float* std3, void* stream) {
    if (!src_hwc || (!out_nchw && !out_packed) || !mean3 || !std3) { vd3d_set_error("preprocess_image: null pointer"); return VD3D_EINVAL; }
    if (Hs <= 0 || Ws <= 0 || crop_top < 0 || crop_top >= Hs || Hr <= 0 || Wr <= 0 || H <= 0 || W <= 0 || Hr < H) {
        vd3d_set_error("preprocess_image: bad sizes (the resized height must cover the network input height)");
        return VD3D_EINVAL;
    }
    PreArgs a;
    a.src = src_hwc; a.out_nchw = out_nchw; a.out_packed = (short*)out_packed;
    a.Hs = Hs; a.Ws = Ws; a.crop_top = crop_top; a.Hr = Hr; a.Wr = Wr; a.H = H; a.W = W;
    a.scale_x = 1.0 / ((double)Wr / (double)Ws);
    a.scale_y = 1.0 / ((double)Hr / (double)(Hs - crop_top));
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }
    dim3 grid((W + 8 + 255) / 256, H + 6);
    hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return vd3d_check_launch("preprocess_image");
}
