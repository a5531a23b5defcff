// postpath.hip -- geometry between the detector output and the KITTI result file, batched on the device (gfx950).
//
// Replaces the per-frame tensor ops of pipelines/evaluators.py:112-129 (test_one): BackProjection
// (networks/utils/utils.py:262-278), theta = alpha2theta_3d (utils/utils.py:47-62; the only output of BBox3dProjector the
// writer uses, networks/utils/utils.py:231-233), the 2D-box shift + rescale to the original image (evaluators.py:118-127)
// and the bottom-centre y of data/kitti/utils.py:180-182.  fp32, reference operation order, contraction off.  One lane per
// detection of the padded batch; the result rows are what write_result_to_file prints.
#pragma clang fp contract(off)
#include "common.h"

namespace {

__global__ void kitti_postpath_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ counts,
                                      const float* __restrict__ P2s, const float* __restrict__ xform,
                                      float* __restrict__ out, int cap) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = counts ? counts[b] : cap;
    if (i >= cap) return;
    float* o = out + ((size_t)b * cap + i) * 12;
    if (i >= n) {
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = 0.f;
        return;
    }
    const float* bx = boxes + ((size_t)b * cap + i) * 11;
    const float* P = P2s + b * 12;
    const float fx = P[0], cx = P[2], tx = P[3], fy = P[5], cy = P[6], ty = P[7];
    const float* xf = xform + b * 4;                       // shift_left, shift_top, scale_x, scale_y (host, fp64 -> fp32)
    const float z = bx[6];
    const float x3d = (bx[4] * z - cx * z - tx) / fx;      // BackProjection
    const float y3d = (bx[5] * z - cy * z - ty) / fy;
    const float theta = bx[10] + atan2f(x3d + tx / fx, z); // alpha2theta_3d
    o[0] = (bx[0] + xf[0]) * xf[2];
    o[1] = (bx[1] + xf[1]) * xf[3];
    o[2] = (bx[2] + xf[0]) * xf[2];
    o[3] = (bx[3] + xf[1]) * xf[3];
    o[4] = x3d;
    o[5] = y3d + 0.5f * bx[8];                             // KITTI wants the bottom centre
    o[6] = z;
    o[7] = bx[7];
    o[8] = bx[8];
    o[9] = bx[9];
    o[10] = bx[10];
    o[11] = theta;
}

}  // namespace

extern "C" int vd3d_kitti_postpath(const float* boxes, const int32_t* counts, const float* P2s, const float* xform,
                                   float* out, int B, int cap, void* stream) {
    if (B < 0 || cap < 0) { vd3d_set_error("kitti_postpath: negative sizes"); return VD3D_EINVAL; }
    if (B == 0 || cap == 0) return VD3D_OK;
    if (!boxes || !P2s || !xform || !out) { vd3d_set_error("kitti_postpath: null pointer"); return VD3D_EINVAL; }
    hipLaunchKernelGGL(kitti_postpath_kernel, dim3((cap + 63) / 64, B), dim3(64), 0, (hipStream_t)stream, boxes, counts, P2s,
                       xform, out, cap);
    return vd3d_check_launch("kitti_postpath");
}
