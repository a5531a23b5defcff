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

// Greedy NMS over K boxes already in decreasing-score order.  box(i) -> f32x4.  alive[] in LDS (bytes).  64-box chunks.
// Round 5: a chunk is resolved in PARALLEL.  (Before: wave 0 walked the chunk's 64 boxes one after the other -- 64 dependent steps of five
// shuffles + an IoU test -- while the other 15 waves waited at a barrier; with ~1 500 candidates per frame (config 3's workload) the stage
// took 400 us, with 4 000 two milliseconds.)  Now: (a) wave 0 stages the chunk's boxes; (b) every wave takes rows of the chunk's 64 x 64
// suppression matrix: lane j tests box j against row box i, one ballot = the row's 64-bit mask; (c) the greedy pass over the chunk is then 64
// steps of `alive &= ~mask[i]` on one 64-bit word, run redundantly by every thread (no barrier to publish it); (d) every thread tests the
// later boxes against the chunk's survivors, as before.  A chunk without a live box costs two barriers.  Same comparisons in the same
// operand order as the sequential walk: the same survivors.
template <typename BoxFn>
__device__ void nms_sorted(BoxFn box, int K, float thr, unsigned char* alive, f32x4* chunk_box, float* chunk_area,
                           unsigned char* chunk_alive, uint64_t* chunk_mask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i = threadIdx.x; i < K; i += blockDim.x) alive[i] = 1;
    for (int c0 = 0; c0 < K; c0 += 64) {
        __syncthreads();                                  // alive[] of this chunk is final; the chunk buffers are free
        const int cn = min(64, K - c0);
        if (threadIdx.x < 64) {
            const int j = c0 + lane;
            const bool in = j < K;
            f32x4 bj = {0.f, 0.f, 0.f, 0.f};
            if (in) bj = box(j);
            chunk_box[lane] = bj;
            chunk_area[lane] = (bj[2] - bj[0]) * (bj[3] - bj[1]);
            chunk_alive[lane] = in && alive[j];
        }
        __syncthreads();
        uint64_t am = __builtin_amdgcn_ballot_w64(chunk_alive[lane] != 0);       // the same word in every wave
        if (!am) continue;                                // block-uniform: nothing alive in this chunk
        const f32x4 bl = chunk_box[lane];
        const float al = chunk_area[lane];
        for (int i = wave; i < cn; i += nw) {
            if (!((am >> i) & 1ull)) continue;            // wave-uniform: a dead box suppresses nothing
            const uint64_t row = __builtin_amdgcn_ballot_w64(lane > i && lane < cn && iou_gt(chunk_box[i], chunk_area[i], bl, al, thr));
            if (lane == 0) chunk_mask[i] = row;
        }
        __syncthreads();
        for (int i = 0; i < cn; ++i)
            if ((am >> i) & 1ull) am &= ~chunk_mask[i];   // (LDS broadcast reads; every thread computes the same word)
        if (threadIdx.x < cn) alive[c0 + threadIdx.x] = (am >> threadIdx.x) & 1ull;
        for (int j = c0 + 64 + threadIdx.x; j < K; j += blockDim.x) {
            if (!alive[j]) continue;
            const f32x4 bj = box(j);
            const float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
            for (int i = 0; i < cn; ++i) {
                if (((am >> i) & 1ull) && iou_gt(chunk_box[i], chunk_area[i], bj, aj, thr)) { alive[j] = 0; break; }
            }
        }
    }
    __syncthreads();
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
