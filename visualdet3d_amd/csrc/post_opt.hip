// Hill-climbing yaw refinement of decoded boxes, batched on the device: one lane per detection, fp64 like the
// reference's numpy/numba code.  Replaces the per-box host loop of heads/detection_3d_head.py:294-308 (_post_process)
// and lib/fast_utils/hill_climbing.py:7-122 (post_opt / hill_climb / test_projection), bbox3d.py:19-82 (project_3d),
// bbox2d.py:39-66 (iou_2d), utils/utils.py:30-45 (alpha <-> ry).  Each box costs <= ~60 projections of 8 corners, so
// the kernel is latency-trivial; what it removes is N x (D2H sync + numpy round trip) per frame.
#pragma clang fp contract(off)
#include "common.h"

namespace {

constexpr double kPi = 3.14159265358979323846;

struct Cam {
    double p[3][4];     // P2
    double inv[3][4];   // first three rows of inverse([P2; 0 0 0 1])
};

__device__ inline void make_cam(const float* P2, Cam& c) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) c.p[i][j] = (double)P2[i * 4 + j];
    const double (*m)[4] = c.p;
    double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
    double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
    double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
    double id = 1.0 / det;
    double a[3][3];
    a[0][0] = c00 * id;
    a[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
    a[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
    a[1][0] = c01 * id;
    a[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
    a[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
    a[2][0] = c02 * id;
    a[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
    a[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c.inv[i][j] = a[i][j];
        c.inv[i][3] = -(a[i][0] * m[0][3] + a[i][1] * m[1][3] + a[i][2] * m[2][3]);
    }
}

struct Box {
    double b[4];            // 2D box
    double X, Y, Z;         // back-projected centre
    double w, h, l;
    double clamp_w, clamp_h;
};

// IoU of the 2D box with the hull of the projected 3D box at yaw ry (hill_climbing.py:84-122).
__device__ inline double projected_iou(const Cam& c, const Box& q, double ry) {
    double sn, cs;
    sincos(ry, &sn, &cs);
    double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
    // corner order of bbox3d.py:47-49: x {0,l,l,l,l,0,0,0}, y {0,0,h,h,0,0,h,h}, z {0,0,0,w,w,w,w,0} minus half sizes
    const unsigned xm = 0x1Eu, ym = 0xCCu, zm = 0x78u;
    for (int i = 0; i < 8; ++i) {
        double x = ((xm >> i) & 1u) ? q.l : 0.0; x -= q.l / 2;
        double y = ((ym >> i) & 1u) ? q.h : 0.0; y -= q.h / 2;
        double z = ((zm >> i) & 1u) ? q.w : 0.0; z -= q.w / 2;
        double X = cs * x + sn * z + q.X;
        double Y = y + q.Y;
        double Z = -sn * x + cs * z + q.Z;
        double u = c.p[0][0] * X + c.p[0][1] * Y + c.p[0][2] * Z + c.p[0][3];
        double v = c.p[1][0] * X + c.p[1][1] * Y + c.p[1][2] * Z + c.p[1][3];
        double s = c.p[2][0] * X + c.p[2][1] * Y + c.p[2][2] * Z + c.p[2][3];
        u /= s; v /= s;
        umin = fmin(umin, u); umax = fmax(umax, u);
        vmin = fmin(vmin, v); vmax = fmax(vmax, v);
    }
    double xn = fmax(0.0, umin), yn = fmax(0.0, vmin);
    double x2n = fmin(umax, q.clamp_w), y2n = fmin(vmax, q.clamp_h);
    double x1 = fmax(q.b[0], xn), x2 = fmin(q.b[2], x2n);
    double y1 = fmax(q.b[1], yn), y2 = fmin(q.b[3], y2n);
    double dx = x2 - x1, dy = y2 - y1;
    if (dx <= 0 || dy <= 0) return 0.0;
    double a0 = (q.b[2] - q.b[0]) * (q.b[3] - q.b[1]);
    double a1 = (x2n - xn) * (y2n - yn);
    double ov = dx * dy;
    return ov / (a0 + a1 - ov);
}

__global__ void post_opt_kernel(float* __restrict__ boxes, const int32_t* __restrict__ labels,
                                const int32_t* __restrict__ counts, const float* __restrict__ P2s, int cap,
                                float clamp_w, float clamp_h, float min_depth, int target_label) {
    int b = blockIdx.y;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = counts ? counts[b] : cap;
    if (i >= n || i >= cap) return;
    float* bx = boxes + ((size_t)b * cap + i) * 11;
    float zf = bx[6];
    if (!(zf > min_depth) || labels[(size_t)b * cap + i] != target_label) return;

    Cam c;
    make_cam(P2s + b * 12, c);
    Box q;
    for (int k = 0; k < 4; ++k) q.b[k] = (double)bx[k];
    double cx = bx[4], cy = bx[5], z = zf;
    q.w = bx[7]; q.h = bx[8]; q.l = bx[9];
    q.clamp_w = clamp_w; q.clamp_h = clamp_h;
    double hx = cx * z, hy = cy * z;
    q.X = c.inv[0][0] * hx + c.inv[0][1] * hy + c.inv[0][2] * z + c.inv[0][3];
    q.Y = c.inv[1][0] * hx + c.inv[1][1] * hy + c.inv[1][2] * z + c.inv[1][3];
    q.Z = c.inv[2][0] * hx + c.inv[2][1] * hy + c.inv[2][2] * z + c.inv[2][3];

    double ray = atan2(cx - c.p[0][2], c.p[0][0]);
    double ry = (double)bx[10] + ray;                  // utils.py:30-37
    if (ry > kPi) ry -= 2 * kPi;
    if (ry <= -kPi) ry += 2 * kPi;

    double step = 0.4, best = projected_iou(c, q, ry);
    for (int it = 0; it < 4096 && step > 0.01; ++it) {  // hill_climbing.py:53-81
        double neg = projected_iou(c, q, ry - step);
        double pos = projected_iou(c, q, ry + step);
        bool pi_ = (pos - best) > 0.0, ni_ = (neg - best) > 0.0;
        if (!pi_ && !ni_) step *= 0.5;
        else if (pi_ && pos > neg) { ry += step; best = pos; }
        else if (ni_) { ry -= step; best = neg; }
        else step *= 0.5;
    }
    while (ry > 3.14) ry -= 3.14 * 2;                   // sic: the reference mixes 3.14 and pi
    while (ry < -3.14) ry += kPi * 2;

    double alpha = ry - ray;                            // utils.py:39-45
    if (alpha > kPi) alpha -= 2 * kPi;
    if (alpha <= -kPi) alpha += 2 * kPi;
    bx[10] = (float)alpha;
}

}  // namespace

extern "C" int vd3d_post_opt(float* boxes, const int32_t* labels, const int32_t* counts, const float* P2s, int B, int cap,
                             float clamp_w, float clamp_h, float min_depth, int target_label, void* stream) {
    if (B < 0 || cap < 0) { vd3d_set_error("post_opt: negative sizes"); return VD3D_EINVAL; }
    if (B == 0 || cap == 0) return VD3D_OK;
    if (!boxes || !labels || !P2s) { vd3d_set_error("post_opt: null pointer"); return VD3D_EINVAL; }
    dim3 grid((cap + 63) / 64, B);
    hipLaunchKernelGGL(post_opt_kernel, grid, dim3(64), 0, (hipStream_t)stream, boxes, labels, counts, P2s, cap, clamp_w,
                       clamp_h, min_depth, target_label);
    return vd3d_check_launch("post_opt");
}
