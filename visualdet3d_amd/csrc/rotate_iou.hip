// rotate_iou.hip -- rotated-rectangle IoU of the KITTI AP evaluator on the device (gfx950).
//
// Replaces the numba.cuda kernel of evaluator/kitti/rotate_iou.py (:261-292 rotate_iou_kernel_eval, :245-258
// devRotateIoUEval, :230-243 inter, :205-227 rbbox_to_corners, :180-202 quadrilateral_intersection, :160-177
// point_in_quadrilateral, :71-116 line_segment_intersection, :32-68 sort_vertex_in_convex_polygon, :15-29 area) called from
// evaluator/kitti/eval.py:124,173.  One lane per (box, query) pair; fp32 with fp64 exactly where numba's typing promotes
// (the "/ 2.0" of the triangle area, the area sum and the final ratio); contraction off.  The reference keeps at most 8
// intersection points in a 16-float local array and writes past it when two nearly identical rectangles yield more
// (undefined behaviour in the numba kernel); here points beyond the eighth are dropped.
#pragma clang fp contract(off)
#include "common.h"

namespace {

__device__ inline void corners_of(const float* r, float* c) {
    const float a_cos = cosf(r[4]), a_sin = sinf(r[4]);
    const float xs[4] = {-r[2] / 2, -r[2] / 2, r[2] / 2, r[2] / 2};
    const float ys[4] = {-r[3] / 2, r[3] / 2, r[3] / 2, -r[3] / 2};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = a_cos * xs[i] + a_sin * ys[i] + r[0];
        c[2 * i + 1] = -a_sin * xs[i] + a_cos * ys[i] + r[1];
    }
}

__device__ inline bool in_quad(float px, float py, const float* c) {
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1];
    const float ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = px - c[0], ap1 = py - c[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

__device__ inline bool seg_x(const float* p1, const float* p2, int i, int j, float* t) {
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    const bool acd = DA1 * CA0 > CA1 * DA0;
    const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd != bcd) {
        const bool abc = CA1 * BA0 > BA1 * CA0;
        const bool abd = DA1 * BA0 > BA1 * DA0;
        if (abc != abd) {
            const float DC0 = D0 - C0, DC1 = D1 - C1;
            const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
            const float DH = BA1 * DC0 - BA0 * DC1;
            t[0] = (ABBA * DC0 - BA0 * CDDC) / DH;
            t[1] = (ABBA * DC1 - BA1 * CDDC) / DH;
            return true;
        }
    }
    return false;
}

__device__ double inter_area(const float* r1, const float* r2) {
    float c1[8], c2[8], pts[16];
    corners_of(r1, c1);
    corners_of(r2, c2);
    int n = 0;
    auto push = [&](float x, float y) {
        if (n < 8) { pts[2 * n] = x; pts[2 * n + 1] = y; }
        ++n;
    };
    for (int i = 0; i < 4; ++i) {
        if (in_quad(c1[2 * i], c1[2 * i + 1], c2)) push(c1[2 * i], c1[2 * i + 1]);
        if (in_quad(c2[2 * i], c2[2 * i + 1], c1)) push(c2[2 * i], c2[2 * i + 1]);
    }
    float t[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_x(c1, c2, i, j, t)) push(t[0], t[1]);
    if (n > 8) n = 8;
    if (n > 0) {   // order the vertices around their centroid (monotone key of the direction), insertion sort
        float cx = 0.f, cy = 0.f, vs[8];
        for (int i = 0; i < n; ++i) { cx += pts[2 * i]; cy += pts[2 * i + 1]; }
        cx /= (float)n;
        cy /= (float)n;
        for (int i = 0; i < n; ++i) {
            float v0 = pts[2 * i] - cx, v1 = pts[2 * i + 1] - cy;
            const float d = sqrtf(v0 * v0 + v1 * v1);
            v0 = v0 / d;
            v1 = v1 / d;
            if (v1 < 0) v0 = -2 - v0;
            vs[i] = v0;
        }
        for (int i = 1; i < n; ++i) {
            if (vs[i - 1] > vs[i]) {
                const float temp = vs[i], tx = pts[2 * i], ty = pts[2 * i + 1];
                int j = i;
                while (j > 0 && vs[j - 1] > temp) {
                    vs[j] = vs[j - 1];
                    pts[2 * j] = pts[2 * j - 2];
                    pts[2 * j + 1] = pts[2 * j - 1];
                    --j;
                }
                vs[j] = temp;
                pts[2 * j] = tx;
                pts[2 * j + 1] = ty;
            }
        }
    }
    double area = 0.0;
    for (int i = 0; i < n - 2; ++i) {
        const float* a = pts;
        const float* b = pts + 2 * i + 2;
        const float* c = pts + 2 * i + 4;
        area += fabs((double)((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0);
    }
    return area;
}

__global__ void rotate_iou_kernel(const float* __restrict__ boxes, const float* __restrict__ query, float* __restrict__ out,
                                  int N, int K, int criterion) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx - (int64_t)n * K);
    float r1[5], r2[5];       // rbox1 = the QUERY box, rbox2 = the box (argument order of rotate_iou.py:290-291)
#pragma unroll
    for (int e = 0; e < 5; ++e) { r1[e] = query[k * 5 + e]; r2[e] = boxes[n * 5 + e]; }
    const float area1 = r1[2] * r1[3], area2 = r2[2] * r2[3];
    const double ai = inter_area(r1, r2);
    double v;
    if (criterion == -1) v = ai / ((double)(area1 + area2) - ai);
    else if (criterion == 0) v = ai / (double)area1;
    else if (criterion == 1) v = ai / (double)area2;
    else v = ai;
    out[idx] = (float)v;
}

}  // namespace

extern "C" int vd3d_rotate_iou_eval(const float* boxes, const float* query_boxes, int N, int K, int criterion, float* iou,
                                    void* stream) {
    if (N < 0 || K < 0) { vd3d_set_error("rotate_iou_eval: negative sizes"); return VD3D_EINVAL; }
    if (N == 0 || K == 0) return VD3D_OK;
    if (!boxes || !query_boxes || !iou) { vd3d_set_error("rotate_iou_eval: null pointer"); return VD3D_EINVAL; }
    const int64_t total = (int64_t)N * K;
    hipLaunchKernelGGL(rotate_iou_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes,
                       query_boxes, iou, N, K, criterion);
    return vd3d_check_launch("rotate_iou_eval");
}
