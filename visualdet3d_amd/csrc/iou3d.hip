// iou3d.hip -- rotated BEV overlap / IoU / NMS for gfx950.
//
// Replaces the reference's iou3d_cuda extension (lib/ops/iou3d/src/iou3d.cpp:174-179; kernels
// src/iou3d_kernel.cu:223-248 pairwise, :250-348 NMS masks; host scan iou3d.cpp:73-170).
// Differences by design (MI355X-first):
//   * pairwise kernels: one 64-lane wave covers 64 consecutive b-boxes of one a-box (a-box is wave-uniform -> SGPRs);
//   * NMS: the 64x64 suppression bitmask is one wave per block (64 == wave size, one ballot word per row), and the
//     greedy scan that the reference runs on the HOST after a blocking cudaMemcpy runs on the DEVICE in one wave
//     (remv words live in lanes) -- no host round trip, stream-ordered, graph-capturable;
//   * errors are return codes, never exit().
// Arithmetic: fp32, contraction off, the reference's operation order (intersection points, contained corners with
// MARGIN 1e-5, angular ordering via atan2, fan shoelace), so areas agree to round-off.
#pragma clang fp contract(off)
#include "common.h"

namespace {

constexpr float kEps = 1e-8f;
constexpr float kMargin = 1e-5f;

struct P2 { float x, y; };

VD3D_DEV float cross3(const P2& p1, const P2& p2, const P2& p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

VD3D_DEV bool seg_intersect(const P2& p1, const P2& p0, const P2& q1, const P2& q0, P2& ans) {
    const bool bb = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                    fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
    if (!bb) return false;
    const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0.f && s3 * s4 > 0.f)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > kEps) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

VD3D_DEV P2 rot(const P2& c, float cs, float sn, const P2& p) {
    P2 r;
    r.x = (p.x - c.x) * cs + (p.y - c.y) * sn + c.x;
    r.y = -(p.x - c.x) * sn + (p.y - c.y) * cs + c.y;
    return r;
}

VD3D_DEV bool in_box(const float* box, const P2& p) {
    const float cx = (box[0] + box[2]) / 2.f, cy = (box[1] + box[3]) / 2.f;
    const float cs = cosf(-box[4]), sn = sinf(-box[4]);
    const float rx = (p.x - cx) * cs + (p.y - cy) * sn + cx;
    const float ry = -(p.x - cx) * sn + (p.y - cy) * cs + cy;
    return rx > box[0] - kMargin && rx < box[2] + kMargin && ry > box[1] - kMargin && ry < box[3] + kMargin;
}

// rotated rectangle intersection area (semantics of iou3d_kernel.cu:108-212)
__device__ float box_overlap(const float* a, const float* b) {
    const P2 ca = {(a[0] + a[2]) / 2.f, (a[1] + a[3]) / 2.f};
    const P2 cb = {(b[0] + b[2]) / 2.f, (b[1] + b[3]) / 2.f};
    P2 A[5] = {{a[0], a[1]}, {a[2], a[1]}, {a[2], a[3]}, {a[0], a[3]}, {0.f, 0.f}};
    P2 B[5] = {{b[0], b[1]}, {b[2], b[1]}, {b[2], b[3]}, {b[0], b[3]}, {0.f, 0.f}};
    const float acs = cosf(a[4]), asn = sinf(a[4]), bcs = cosf(b[4]), bsn = sinf(b[4]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        A[k] = rot(ca, acs, asn, A[k]);
        B[k] = rot(cb, bcs, bsn, B[k]);
    }
    A[4] = A[0];
    B[4] = B[0];
    P2 pts[16];
    float ang[16];
    int cnt = 0;
    P2 sum = {0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 x;
            if (seg_intersect(A[i + 1], A[i], B[j + 1], B[j], x)) {
                sum.x = sum.x + x.x; sum.y = sum.y + x.y;
                pts[cnt++] = x;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (in_box(a, B[k])) { sum.x = sum.x + B[k].x; sum.y = sum.y + B[k].y; pts[cnt++] = B[k]; }
        if (in_box(b, A[k])) { sum.x = sum.x + A[k].x; sum.y = sum.y + A[k].y; pts[cnt++] = A[k]; }
    }
    if (cnt == 0) return 0.f;
    const P2 c = {sum.x / (float)cnt, sum.y / (float)cnt};
    for (int i = 0; i < cnt; ++i) ang[i] = atan2f(pts[i].y - c.y, pts[i].x - c.x);
    // same adjacent-swap order as the reference's bubble sort (ties keep their relative order)
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                const P2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
                const float u = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = u;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const float ux = pts[k].x - pts[0].x, uy = pts[k].y - pts[0].y;
        const float vx = pts[k + 1].x - pts[0].x, vy = pts[k + 1].y - pts[0].y;
        area += ux * vy - uy * vx;
    }
    return fabsf(area) / 2.0f;
}

VD3D_DEV float iou_bev(const float* a, const float* b) {
    const float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
    const float ov = box_overlap(a, b);
    return ov / fmaxf(sa + sb - ov, kEps);
}

VD3D_DEV float iou_normal(const float* a, const float* b) {
    const float l = fmaxf(a[0], b[0]), r = fminf(a[2], b[2]), t = fmaxf(a[1], b[1]), bt = fminf(a[3], b[3]);
    const float w = fmaxf(r - l, 0.f), h = fmaxf(bt - t, 0.f);
    const float inter = w * h;
    const float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
    return inter / fmaxf(sa + sb - inter, kEps);
}

template <int MODE>  // 0: overlap area, 1: rotated IoU
__global__ void __launch_bounds__(256) pairwise_kernel(const float* __restrict__ A, int na, const float* __restrict__ B, int nb,
                                                       float* __restrict__ out) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);   // wave-uniform
    if (i >= na || j >= nb) return;
    float a[5], b[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) { a[k] = A[i * 5 + k]; b[k] = B[j * 5 + k]; }
    out[(int64_t)i * nb + j] = MODE == 0 ? box_overlap(a, b) : iou_bev(a, b);
}

// mask[i][cb] bit j: box (64*cb + j) is suppressed by box i (only j > i inside the diagonal block)
template <int NORMAL>
__global__ void __launch_bounds__(64) nms_mask_kernel(const float* __restrict__ boxes, int n, float thr, uint64_t* __restrict__ mask) {
    const int rb = blockIdx.y, cb = blockIdx.x, lane = threadIdx.x;
    const int col_blocks = (n + 63) / 64;
    __shared__ float cbox[64 * 5];
    const int csize = min(64, n - cb * 64), rsize = min(64, n - rb * 64);
    if (lane < csize) {
#pragma unroll
        for (int k = 0; k < 5; ++k) cbox[lane * 5 + k] = boxes[(int64_t)(cb * 64 + lane) * 5 + k];
    }
    __syncthreads();
    if (lane < rsize) {
        const int i = rb * 64 + lane;
        float a[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) a[k] = boxes[(int64_t)i * 5 + k];
        uint64_t t = 0;
        for (int j = (rb == cb ? lane + 1 : 0); j < csize; ++j) {
            const float v = NORMAL ? iou_normal(a, cbox + j * 5) : iou_bev(a, cbox + j * 5);
            if (v > thr) t |= 1ull << j;
        }
        mask[(int64_t)i * col_blocks + cb] = t;
    }
}

// greedy scan (iou3d.cpp:100-116) in one wave: lane l owns removed-words l, l+64, ...
__global__ void __launch_bounds__(64) nms_scan_kernel(const uint64_t* __restrict__ mask, int n, int32_t* __restrict__ keep, int32_t* __restrict__ count) {
    const int lane = threadIdx.x;
    const int col_blocks = (n + 63) / 64;
    constexpr int kMaxWords = 16;            // supports n <= 64*64*16 = 65536
    uint64_t remv[kMaxWords];
#pragma unroll
    for (int w = 0; w < kMaxWords; ++w) remv[w] = 0;
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const int nblock = i >> 6, inblock = i & 63;
        // word `nblock` lives in lane (nblock & 63), slot (nblock >> 6)
        uint64_t word = 0;
#pragma unroll
        for (int w = 0; w < kMaxWords; ++w)
            if (w == (nblock >> 6)) word = remv[w];
        const uint32_t lo = __shfl((int)(uint32_t)word, nblock & 63), hi = __shfl((int)(uint32_t)(word >> 32), nblock & 63);
        const uint64_t rw = ((uint64_t)hi << 32) | lo;
        if (!((rw >> inblock) & 1ull)) {
            if (lane == 0) keep[kept] = i;
            ++kept;
            const uint64_t* row = mask + (int64_t)i * col_blocks;
#pragma unroll
            for (int w = 0; w < kMaxWords; ++w) {
                const int cbk = w * 64 + lane;
                if (cbk >= nblock && cbk < col_blocks) remv[w] |= row[cbk];
            }
        }
    }
    if (lane == 0) *count = kept;
}

}  // namespace

extern "C" int vd3d_boxes_overlap_bev(const float* a, int na, const float* b, int nb, float* out, void* stream) {
    if (na < 0 || nb < 0 || (na && nb && (!a || !b || !out))) { vd3d_set_error("boxes_overlap_bev: bad args"); return VD3D_EINVAL; }
    if (na == 0 || nb == 0) return VD3D_OK;
    hipLaunchKernelGGL(pairwise_kernel<0>, dim3((nb + 63) / 64, (na + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, out);
    return vd3d_check_launch("boxes_overlap_bev");
}

extern "C" int vd3d_boxes_iou_bev(const float* a, int na, const float* b, int nb, float* out, void* stream) {
    if (na < 0 || nb < 0 || (na && nb && (!a || !b || !out))) { vd3d_set_error("boxes_iou_bev: bad args"); return VD3D_EINVAL; }
    if (na == 0 || nb == 0) return VD3D_OK;
    hipLaunchKernelGGL(pairwise_kernel<1>, dim3((nb + 63) / 64, (na + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, out);
    return vd3d_check_launch("boxes_iou_bev");
}

extern "C" int64_t vd3d_nms_bev_workspace_bytes(int n) {
    const int64_t cb = (n + 63) / 64;
    return (int64_t)(n > 0 ? n : 1) * cb * 8 + 64;
}

extern "C" int vd3d_nms_bev(const float* boxes, int n, float thr, int normal, int32_t* keep, int32_t* count, void* workspace, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || !keep || !count || (n > 0 && (!boxes || !workspace))) { vd3d_set_error("nms_bev: bad args"); return VD3D_EINVAL; }
    if (n > 65536) { vd3d_set_error("nms_bev: n > 65536 not supported"); return VD3D_EINVAL; }
    uint64_t* mask = (uint64_t*)workspace;
    const int cb = (n + 63) / 64;
    if (n > 0) {
        if (normal) hipLaunchKernelGGL(nms_mask_kernel<1>, dim3(cb, cb), dim3(64), 0, s, boxes, n, thr, mask);
        else hipLaunchKernelGGL(nms_mask_kernel<0>, dim3(cb, cb), dim3(64), 0, s, boxes, n, thr, mask);
        int rc = vd3d_check_launch("nms_mask");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, s, mask, n, keep, count);
    return vd3d_check_launch("nms_scan");
}
