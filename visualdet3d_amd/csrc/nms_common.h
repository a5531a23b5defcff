// nms_common.h -- device helpers shared by postprocess.hip and km3d_decode.hip: orderable float keys, in-LDS bitonic
// sort, order-preserving compaction, greedy NMS over score-sorted boxes.  Include AFTER `#pragma clang fp contract(off)`:
// the IoU test must evaluate in plain fp32 exactly like the oracle (oracle/nms_ref.py).
#pragma once
#include "common.h"

namespace {

constexpr int kNmsThreads = 1024;

VD3D_DEV uint32_t orderable_desc(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;  // ascending-orderable
    return ~u;                                   // ascending key == descending score
}
// ---- shared device pieces ----------------------------------------------------------------------------
// in-LDS bitonic sort of n (padded to pow2 P) 64-bit keys, ascending
__device__ void bitonic_sort(uint64_t* keys, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

VD3D_DEV bool iou_gt(const f32x4& a, float area_a, const f32x4& b, float area_b, float thr) {
    const float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]);
    const float xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float iou = inter / ((area_a + area_b) - inter);
    return iou > thr;
}

// Greedy NMS over K boxes already in decreasing-score order.  box(i) -> f32x4.  alive[] in LDS (bytes).
// 64-box chunks: wave 0 resolves a chunk wave-synchronously, then every thread tests later boxes against the
// chunk's survivors.
template <typename BoxFn>
__device__ void nms_sorted(BoxFn box, int K, float thr, unsigned char* alive, f32x4* chunk_box, float* chunk_area,
                           unsigned char* chunk_alive) {
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < K; i += blockDim.x) alive[i] = 1;
    __syncthreads();
    for (int c0 = 0; c0 < K; c0 += 64) {
        if (threadIdx.x < 64) {
            const int j = c0 + lane;
            const bool in = j < K;
            f32x4 bj = {0.f, 0.f, 0.f, 0.f};
            if (in) bj = box(j);
            const float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
            bool a = in && alive[j];
            const int cn = min(64, K - c0);
            for (int i = 0; i < cn; ++i) {
                const bool ai = __shfl((int)a, i) != 0;
                if (!ai) continue;  // wave-uniform
                f32x4 bi;
                bi[0] = __shfl(bj[0], i); bi[1] = __shfl(bj[1], i); bi[2] = __shfl(bj[2], i); bi[3] = __shfl(bj[3], i);
                const float areai = __shfl(aj, i);
                if (lane > i && a && iou_gt(bi, areai, bj, aj, thr)) a = false;
            }
            if (in) alive[j] = a;
            chunk_box[lane] = bj;
            chunk_area[lane] = aj;
            chunk_alive[lane] = a;
        }
        __syncthreads();
        const int cn = min(64, K - c0);
        for (int j = c0 + 64 + threadIdx.x; j < K; j += blockDim.x) {
            if (!alive[j]) continue;
            const f32x4 bj = box(j);
            const float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
            for (int i = 0; i < cn; ++i) {
                if (chunk_alive[i] && iou_gt(chunk_box[i], chunk_area[i], bj, aj, thr)) { alive[j] = 0; break; }
            }
        }
        __syncthreads();
    }
}

// block-wide order-preserving compaction positions: pos[i] = number of set flags before i; *total = count.
// Each thread owns a contiguous span; span sums are scanned with wave shuffles + one cross-wave step.
__device__ void compact_positions(const unsigned char* flag, int K, int* pos, int* scratch, int* total) {
    const int per = (K + blockDim.x - 1) / blockDim.x;
    const int beg = threadIdx.x * per, end = min(K, beg + per);
    int s = 0;
    for (int i = beg; i < end; ++i) s += flag[i];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) scratch[wv] = inc;
    __syncthreads();
    if (wv == 0) {
        const int ws = lane < nw ? scratch[lane] : 0;
        int inc2 = ws;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc2, o);
            if (lane >= o) inc2 += t;
        }
        if (lane < nw) scratch[lane] = inc2 - ws;
        if (lane == nw - 1) *total = inc2;
    }
    __syncthreads();
    int run = scratch[wv] + inc - s;
    for (int i = beg; i < end; ++i) { pos[i] = run; run += flag[i]; }
    __syncthreads();
}


}  // namespace
