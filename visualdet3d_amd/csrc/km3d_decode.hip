// km3d_decode.hip -- KM3D / RTM3D keypoint-head decoding on the device (gfx950).
//
// Replaces KM3DHead.get_bboxes / _decode (heads/km3d_head.py:255-314, :155-252) and the helpers it calls
// (networks/utils/rtm3d_utils.py: _nms :122-127, _topk :201-216, _topk_channel :219-228, _transpose_and_gather_feat
// :195-199, gen_position :314-455) plus torchvision nms.  The reference runs ~60 tensor ops with several host syncs on
// batch 1; here it is three launches for the whole batch and no host round trip:
//   1. km3d_peaks_kernel : sigmoid + 3x3 "is local maximum" test on the class heat-map and the keypoint heat-map (NHWC
//                          fp32 logits); peaks above the threshold that matters downstream (score_thr for hm, 0.1 for
//                          hm_hp) are appended to a per-(sample, channel) list.  Peaks at or below those thresholds can
//                          never influence the result (they are dropped / masked later), so the top-K of the list equals
//                          the reference's top-K restricted to what survives.
//   2. km3d_topk_kernel  : one workgroup per (sample, channel): LDS bitonic sort, top-K (K = 100).
//   3. km3d_decode_kernel: one workgroup per sample: merge the per-class top-K into the overall top-K, gather the
//                          regression maps (NHWC: one contiguous read per detection), keypoint <-> heat-map association,
//                          x4 rescale, rotation decode, 16x3 least squares in fp64 (closed-form 3x3 inverse; the
//                          reference's random 1e-8 jitter before torch.inverse is omitted), re-projection, clip, score
//                          mask, class-agnostic NMS.
// fp32 arithmetic with contraction off, in the reference's operation order.
#pragma clang fp contract(off)
#include "common.h"
#include "nms_common.h"

namespace {

constexpr int kMaxJ = 9;
constexpr int kMaxK = 128;

struct KArgs {
    const float *hm, *wh, *hps, *rot, *dim, *prob, *reg, *hm_hp, *hp_offset, *P2, *kconst;
    int B, H, W, n_cls, J, K, max_peaks, img_h, img_w;
    float score_thr, nms_thr;
    int32_t* peak_count;   // [B][n_ch]
    float* peak_score;     // [B][n_ch][max_peaks]
    int32_t* peak_idx;     // [B][n_ch][max_peaks]
    float* top_score;      // [B][n_ch][K]
    int32_t* top_idx;      // [B][n_ch][K]
    int32_t* overflow;     // [B]
    uint64_t* big_keys;    // max_peaks > 8192 only: [B][n_ch][max_peaks] sort keys in global memory (the caller's retry after a peak-list overflow)
    float* out_scores; float* out_boxes; int32_t* out_cls; int32_t* out_count;
};

VD3D_DEV float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void km3d_zero_kernel(int32_t* a, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = 0;
}

// ---- 1. peaks -----------------------------------------------------------------------------------------------------
// One (sample, channel) per blockIdx.y; a workgroup owns kPeakSpan consecutive pixels, collects its peaks in LDS (wave-aggregated
// LDS append) and reserves room in the channel's global list with ONE atomicAdd.  (One global atomic per wave and iteration --
// the first version -- put ~900 atomics on each of the B x 12 counters: with many candidates, e.g. untrained weights, those
// serialised into 0.7 ms per launch at 16 x 128 x 440.)  The order inside a list is irrelevant: the top-K kernel sorts it.
// One workgroup = one 8 x 64-pixel tile of ONE heat map (blockIdx.z: class map | keypoint map) with ALL its channels: the tile's logits
// (+ a one-pixel ring) are read once, as contiguous [pixel][channel] row segments, their sigmoid goes to LDS in the same layout, and
// the 3x3 test compares LDS values (ring positions outside the image hold 0, below every candidate).  Peaks are collected per channel
// in LDS (wave-aggregated append) and each channel's run reserves room in the global list with ONE atomicAdd.
// History: v1 one (sample, channel) per workgroup, sigmoid of the eight neighbours re-evaluated from global memory per candidate, one
// global atomic per wave (0.7 ms at 16 x 128 x 440 with untrained weights); v2 workgroup-aggregated atomics (127 us); v3 the span's
// sigmoid once into LDS (97 us: every workgroup still walked its channel with a 12 / 36-byte stride, 9 channels re-reading the same
// lines).  Same function, same comparisons: the lists hold the same peaks (their order inside a list is irrelevant: the top-K kernel
// sorts by (score, index)).
constexpr int kPeakTH = 8, kPeakTW = 64, kPeakPix = kPeakTH * kPeakTW, kPeakRing = (kPeakTH + 2) * (kPeakTW + 2);
constexpr int kPeakMaxC = 9;
__global__ void __launch_bounds__(256) km3d_peaks_kernel(const KArgs p, int tiles_x, FastDiv fd_tiles_x, FastDiv fd_rowlen0, FastDiv fd_c0, FastDiv fd_rowlen1,
                                                         FastDiv fd_c1) {
    extern __shared__ __attribute__((aligned(16))) char peaks_smem[];
    __shared__ int s_n[kPeakMaxC], s_base[kPeakMaxC];
    __shared__ uint64_t s_mask[kPeakMaxC * kPeakTH];
    const bool is_hm = blockIdx.z == 0;
    const int C = is_hm ? p.n_cls : p.J;
    if (C <= 0) return;
    const FastDiv fd_rowlen = is_hm ? fd_rowlen0 : fd_rowlen1;
    (void)fd_c0; (void)fd_c1;
    const float* m = is_hm ? p.hm : p.hm_hp;
    const float thr = is_hm ? p.score_thr : 0.1f;
    const int ch0 = is_hm ? 0 : p.n_cls, nch = p.n_cls + p.J;
    const int b = blockIdx.y, H = p.H, W = p.W;
    const int ty = fastdiv((int)blockIdx.x, fd_tiles_x), tx = (int)blockIdx.x - ty * tiles_x;
    const int rowlen = (kPeakTW + 2) * C;                 // floats of one ring row
    float* sg = (float*)peaks_smem;                       // [kPeakTH + 2][kPeakTW + 2][C] sigmoid
    const int lane = threadIdx.x & 63;
    const float* mb = m + (int64_t)b * H * W * C;
    const int x0c = (tx * kPeakTW - 1) * C, WC = W * C;
    // (eight independent loads in flight per thread: one load per loop trip made the fill a chain of ~23 memory round trips)
    for (int i0 = threadIdx.x; i0 < kPeakRing * C; i0 += 256 * 8) {
        float raw[8];
        bool ok[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            const int r = fastdiv(i, fd_rowlen), j = i - r * rowlen;
            const int y = ty * kPeakTH - 1 + r, xc = x0c + j;
            ok[u] = i < kPeakRing * C && (unsigned)y < (unsigned)H && (unsigned)xc < (unsigned)WC;
            raw[u] = ok[u] ? mb[(int64_t)y * WC + xc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            if (i < kPeakRing * C) sg[i] = ok[u] ? sigm(raw[u]) : 0.f;
        }
    }
    __syncthreads();
    // thread <-> pixel (two per thread), channels one after the other: every lane of a wave tests the same channel, so a step's result
    // is ONE ballot -- kept as a 64-bit word per (channel, 64-pixel row) in LDS.  (Peak LISTS in LDS, sized for the worst case -- a
    // plateau makes every pixel a peak -- cost 37 KB and left two workgroups per CU; the bit masks leave six.)
    const int wave = threadIdx.x >> 6;
    for (int pl = threadIdx.x; pl < kPeakPix; pl += 256) {
        const int yl = pl / kPeakTW, xl = pl - yl * kPeakTW;             // (a wave = one 64-pixel tile row)
        const int y = ty * kPeakTH + yl, x = tx * kPeakTW + xl;
        const bool inside = y < H && x < W;
        const float* ctr0 = sg + (yl + 1) * rowlen + (xl + 1) * C;
        for (int c = 0; c < C; ++c) {
            const float* ctr = ctr0 + c;
            const float v = ctr[0];
            bool peak = inside && v > thr;
            if (peak) {
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx)
                        if ((dx != 0 || dy != 0) && ctr[dy * rowlen + dx * C] > v) peak = false;
            }
            const uint64_t mask = __ballot(peak);
            if (lane == 0) s_mask[c * kPeakTH + yl] = mask;
        }
    }
    (void)wave;
    __syncthreads();
    // per channel: count, reserve the run in the global list with one atomicAdd, then every peak writes itself at base + its rank
    if (threadIdx.x < C) {
        int n = 0;
        for (int r = 0; r < kPeakTH; ++r) n += __popcll(s_mask[threadIdx.x * kPeakTH + r]);
        s_n[threadIdx.x] = n;
        s_base[threadIdx.x] = n > 0 ? atomicAdd(p.peak_count + b * nch + ch0 + threadIdx.x, n) : 0;
    }
    __syncthreads();
    for (int pl = threadIdx.x; pl < kPeakPix; pl += 256) {
        const int yl = pl / kPeakTW, xl = pl - yl * kPeakTW;
        const int pix = (ty * kPeakTH + yl) * W + tx * kPeakTW + xl;
        for (int c = 0; c < C; ++c) {
            if (s_n[c] == 0) continue;
            const uint64_t mask = s_mask[c * kPeakTH + yl];
            if (!((mask >> xl) & 1ull)) continue;
            int rank = __popcll(mask & ((1ull << xl) - 1ull));
            for (int r = 0; r < yl; ++r) rank += __popcll(s_mask[c * kPeakTH + r]);
            const int pos = s_base[c] + rank;
            if (pos < p.max_peaks) {
                const int64_t slot = (int64_t)(b * nch + ch0 + c) * p.max_peaks;
                p.peak_score[slot + pos] = sg[(yl + 1) * rowlen + (xl + 1) * C + c];
                p.peak_idx[slot + pos] = pix;
            }
        }
    }
}

// ---- 2. per-channel top-K -------------------------------------------------------------------------------------------
// Keys are unique 64-bit values (orderable score << 32 | pixel index), ascending key = descending score, ties by index -- the order
// of torch.topk on the flattened map.  With thousands of peaks per channel (untrained weights: ~half the pixels of the keypoint map
// pass 0.1) a full bitonic sort of the padded list is 91 passes over 8192 keys; only the K = 100 smallest are wanted.  So the list is
// first cut by a byte-wise radix SELECT on the score word: per pass a 256-bin histogram of the next byte among the keys that still
// share the prefix, the bin where the running count crosses K extends the prefix -- until at most kTopkSort keys are <= the prefix
// bound; those are compacted and sorted.  Every one of the K smallest keys has its score prefix <= the bound by construction, so
// the result equals the full sort's.  (Scores equal to the last bit in more than kTopkSort peaks: the full sort, as before.)
constexpr int kTopkSort = 1024;
__global__ void __launch_bounds__(kNmsThreads) km3d_topk_kernel(const KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nch = p.n_cls + p.J;
    const int slot = blockIdx.y * nch + blockIdx.x;   // (b, ch)
    // capacities beyond the LDS sort size (the caller's retry after an overflow; the reference's top-K has no cap, rtm3d_utils.py:201-228): the key
    // list lives in global memory, same steps (a workgroup's global accesses are ordered by __syncthreads like its LDS accesses)
    uint64_t* keys = p.big_keys ? p.big_keys + (int64_t)slot * p.max_peaks : (uint64_t*)smem;     // [max_peaks]
    uint64_t* cand = p.big_keys ? (uint64_t*)smem : keys + p.max_peaks;                            // [kTopkSort]
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix, s_below;
    __shared__ int s_ncand;
    int n = p.peak_count[slot];
    if (n > p.max_peaks) {
        if (threadIdx.x == 0) p.overflow[blockIdx.y] = 1;
        n = p.max_peaks;
    }
    const float* sc = p.peak_score + (int64_t)slot * p.max_peaks;
    const int32_t* ix = p.peak_idx + (int64_t)slot * p.max_peaks;
    uint64_t* sorted = keys;
    int P = 1;
    if (n <= kTopkSort) {
        while (P < n) P <<= 1;
        for (int i = threadIdx.x; i < P; i += blockDim.x)
            keys[i] = i < n ? (((uint64_t)orderable_desc(sc[i]) << 32) | (uint32_t)ix[i]) : ~0ull;   // score desc, index asc
        __syncthreads();
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) keys[i] = ((uint64_t)orderable_desc(sc[i]) << 32) | (uint32_t)ix[i];
        if (threadIdx.x == 0) { s_prefix = 0; s_below = 0; }
        __syncthreads();
        // after pass q the bound is: score word's top 8 (q + 1) bits <= prefix's;  below = keys strictly under the prefix bin
        int shift = 24, nle = n;                      // nle: keys with score word <= bound (all of them before the first pass)
        for (; shift >= 0 && nle > kTopkSort; shift -= 8) {
            for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix, below = s_below;
            const uint32_t himask = shift == 24 ? 0u : ~0u << (shift + 8);
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const uint32_t w = (uint32_t)(keys[i] >> 32);
                if ((w & himask) == prefix) atomicAdd(&hist[(w >> shift) & 255], 1);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int run = (int)below, bin = 0;
                for (; bin < 255; ++bin) {            // the bin in which the running count reaches K
                    if (run + hist[bin] >= p.K) break;
                    run += hist[bin];
                }
                s_prefix = prefix | ((uint32_t)bin << shift);
                s_below = (uint32_t)run;
                s_ncand = run + hist[bin];
            }
            __syncthreads();
            nle = s_ncand;
        }
        if (nle > kTopkSort) {
            // (more than kTopkSort peaks share the K-th score exactly) the full sort
            while (P < n) P <<= 1;
            for (int i = n + threadIdx.x; i < P; i += blockDim.x) keys[i] = ~0ull;
            __syncthreads();
        } else {
            const int sh = shift + 8;                 // the bound covers the score word's bits [31 : sh]
            const uint32_t bound = s_prefix >> sh;
            if (threadIdx.x == 0) s_ncand = 0;
            __syncthreads();
            for (int i0 = 0; i0 < n; i0 += blockDim.x) {
                const int i = i0 + threadIdx.x;
                const uint64_t k = i < n ? keys[i] : ~0ull;
                const bool take = i < n && ((uint32_t)(k >> 32) >> sh) <= bound;
                const uint64_t mask = __ballot(take);
                if (mask) {
                    const int lane = threadIdx.x & 63;
                    int pos0 = 0;
                    if (lane == 0) pos0 = atomicAdd(&s_ncand, __popcll(mask));
                    pos0 = __shfl(pos0, 0);
                    if (take) cand[pos0 + __popcll(mask & ((1ull << lane) - 1ull))] = k;
                }
            }
            __syncthreads();
            n = s_ncand;                              // >= min(K, n): every key that can be among the K smallest
            while (P < n) P <<= 1;
            for (int i = n + threadIdx.x; i < P; i += blockDim.x) cand[i] = ~0ull;
            __syncthreads();
            sorted = cand;
        }
    }
    bitonic_sort(sorted, P);
    for (int k = threadIdx.x; k < p.K; k += blockDim.x) {
        float s = 0.f;
        int id = 0;
        if (k < n) {
            const uint32_t hi = ~(uint32_t)(sorted[k] >> 32);
            const uint32_t u = (hi & 0x80000000u) ? (hi ^ 0x80000000u) : ~hi;   // inverse of the orderable transform
            s = __builtin_bit_cast(float, u);
            id = (int)(uint32_t)sorted[k];
        }
        p.top_score[(int64_t)slot * p.K + k] = s;
        p.top_idx[(int64_t)slot * p.K + k] = id;
    }
}

// ---- 3. per-sample decode ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) km3d_decode_kernel(const KArgs p) {
    __shared__ uint64_t mkeys[512];
    __shared__ float s_score[kMaxK];
    __shared__ int s_ind[kMaxK], s_cls[kMaxK];
    __shared__ float hx[kMaxJ][kMaxK], hy[kMaxJ][kMaxK], hs[kMaxJ][kMaxK];
    __shared__ float kps[kMaxK][2 * kMaxJ];
    __shared__ float bbox[kMaxK][4];
    __shared__ float det[kMaxK][11];
    __shared__ unsigned char keep[kMaxK], alive[kMaxK], chunk_alive[64];
    __shared__ int pos[kMaxK], cidx[kMaxK], scratch[32], total;
    __shared__ f32x4 chunk_box[64];
    __shared__ uint64_t chunk_mask[64];
    __shared__ float chunk_area[64];

    const int b = blockIdx.x, tid = threadIdx.x;
    const int K = p.K, J = p.J, W = p.W, nch = p.n_cls + p.J;
    const int64_t HW = (int64_t)p.H * p.W;
    if (p.overflow[b]) {
        if (tid == 0) p.out_count[b] = -1;
        return;
    }
    // (a) merge per-class top-K -> overall top-K (rtm3d_utils.py:210-214): key = (score desc, flattened cls*K + rank asc)
    const int nm = p.n_cls * K;
    int P = 1;
    while (P < nm) P <<= 1;
    for (int i = tid; i < P; i += blockDim.x) {
        uint64_t k = ~0ull;
        if (i < nm) k = ((uint64_t)orderable_desc(p.top_score[(int64_t)(b * nch) * K + i]) << 32) | (uint32_t)i;
        mkeys[i] = k;
    }
    __syncthreads();
    bitonic_sort(mkeys, P);
    for (int k = tid; k < K; k += blockDim.x) {
        const int flat = (int)(uint32_t)mkeys[k];
        s_score[k] = p.top_score[(int64_t)(b * nch) * K + flat];
        s_ind[k] = p.top_idx[(int64_t)(b * nch) * K + flat];
        s_cls[k] = flat / K;
    }
    // (b) keypoint heat-map candidates per joint (:208-223)
    for (int i = tid; i < J * K; i += blockDim.x) {
        const int j = i / K, m = i - j * K;
        const int64_t slot = (int64_t)(b * nch + p.n_cls + j) * K + m;
        const float s = p.top_score[slot];
        const int id = p.top_idx[slot];
        const float* off = p.hp_offset + ((int64_t)b * HW + id) * 2;
        float xs = (float)(id % W) + off[0], ys = (float)(id / W) + off[1];
        const float mask = s > 0.1f ? 1.0f : 0.0f;
        hs[j][m] = (1.0f - mask) * -1.0f + mask * s;
        hy[j][m] = (1.0f - mask) * -10000.0f + mask * ys;
        hx[j][m] = (1.0f - mask) * -10000.0f + mask * xs;
    }
    __syncthreads();
    // (c) regression gathers (:170-196)
    for (int k = tid; k < K; k += blockDim.x) {
        const int id = s_ind[k];
        const int64_t px = (int64_t)b * HW + id;
        const float xs = (float)(id % W), ys = (float)(id / W);
        for (int j = 0; j < J; ++j) {
            kps[k][2 * j] = p.hps[px * (2 * J) + 2 * j] + xs;
            kps[k][2 * j + 1] = p.hps[px * (2 * J) + 2 * j + 1] + ys;
        }
        const float xr = xs + p.reg[px * 2], yr = ys + p.reg[px * 2 + 1];
        const float w0 = p.wh[px * 2], w1 = p.wh[px * 2 + 1];
        bbox[k][0] = xr - w0 / 2.0f; bbox[k][1] = yr - w1 / 2.0f; bbox[k][2] = xr + w0 / 2.0f; bbox[k][3] = yr + w1 / 2.0f;
    }
    __syncthreads();
    // (d) association (:224-244): nearest heat-map peak of the same joint, accepted if inside the box, confident and close
    for (int i = tid; i < J * K; i += blockDim.x) {
        const int j = i / K, k = i - j * K;
        const float kx = kps[k][2 * j], ky = kps[k][2 * j + 1];
        float best = INFINITY;
        int bi = 0;
        for (int m = 0; m < K; ++m) {
            const float dx = kx - hx[j][m], dy = ky - hy[j][m];
            const float d = sqrtf(dx * dx + dy * dy);
            if (d < best) { best = d; bi = m; }
        }
        const float sx = hx[j][bi], sy = hy[j][bi], ssc = hs[j][bi];
        const float l = bbox[k][0], t = bbox[k][1], r = bbox[k][2], bt = bbox[k][3];
        const bool bad = (sx < l) || (sx > r) || (sy < t) || (sy > bt) || (ssc < 0.1f) || (best > fmaxf(bt - t, r - l) * 0.3f);
        if (!bad) { kps[k][2 * j] = sx; kps[k][2 * j + 1] = sy; }
    }
    __syncthreads();
    // (e) per detection: x4, rotation, least squares, re-projection, clip (:246-291; rtm3d_utils.py:314-455)
    for (int k = tid; k < K; k += blockDim.x) {
        const int id = s_ind[k];
        const int64_t px = (int64_t)b * HW + id;
        const float* P = p.P2 + b * 12;
        const float fx = P[0], cx = P[2], tx = P[3], fy = P[5], cy = P[6], ty = P[7];
        float kp[18];
        for (int i = 0; i < 18; ++i) kp[i] = kps[k][i] * 4.0f;
        float bb[4];
        for (int i = 0; i < 4; ++i) bb[i] = bbox[k][i] * 4.0f;
        const float* rt = p.rot + px * 8;
        const float* dm = p.dim + px * 3;
        const float pi = 3.14159265358979323846f;
        const float aidx = rt[1] > rt[5] ? 1.0f : 0.0f;
        const float alpha1 = atanf(rt[2] / rt[3]) + (-0.5f * pi);
        const float alpha2 = atanf(rt[6] / rt[7]) + (0.5f * pi);
        const float alpha = alpha1 * aidx + alpha2 * (1.0f - aidx);
        float rot_y = alpha + atan2f(kp[16] - cx, fx);
        if (rot_y > pi) rot_y = rot_y - 2.0f * pi;
        if (rot_y < -pi) rot_y = rot_y + 2.0f * pi;
        const float l = dm[2], h = dm[1], w = dm[0];
        const float co = cosf(rot_y), sn = sinf(rot_y);
        const float lc = l * 0.5f * co, ls = l * 0.5f * sn, wc = w * 0.5f * co, ws = w * 0.5f * sn, hh = h * 0.5f;
        const float Bx[8] = {-lc - ws, -lc + ws, -lc + ws, lc + ws, lc + ws, lc - ws, lc - ws, -lc - ws};
        const float By[8] = {-hh, -hh, hh, hh, -hh, -hh, hh, hh};
        const float Cz[8] = {ls - wc, ls + wc, ls + wc, -ls + wc, -ls + wc, -ls - wc, -ls - wc, ls - wc};
        // A (16x3) = [const | kp_norm], rhs = B - kp_norm * C
        double M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        float A[16][3], rhs[16];
        for (int i = 0; i < 16; ++i) {
            const float kn = (kp[i] - ((i & 1) ? cy : cx)) / fx;
            A[i][0] = p.kconst[2 * i]; A[i][1] = p.kconst[2 * i + 1]; A[i][2] = kn;
            const float Bv = (i & 1) ? By[i >> 1] : Bx[i >> 1];
            rhs[i] = Bv - kn * Cz[i >> 1];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) M[r][c] += (double)A[i][r] * (double)A[i][c];
        }
        const double c00 = M[1][1] * M[2][2] - M[1][2] * M[2][1], c01 = M[1][2] * M[2][0] - M[1][0] * M[2][2], c02 = M[1][0] * M[2][1] - M[1][1] * M[2][0];
        const double detM = M[0][0] * c00 + M[0][1] * c01 + M[0][2] * c02;
        double inv[3][3];
        inv[0][0] = c00 / detM; inv[1][0] = c01 / detM; inv[2][0] = c02 / detM;
        inv[0][1] = (M[0][2] * M[2][1] - M[0][1] * M[2][2]) / detM;
        inv[1][1] = (M[0][0] * M[2][2] - M[0][2] * M[2][0]) / detM;
        inv[2][1] = (M[0][1] * M[2][0] - M[0][0] * M[2][1]) / detM;
        inv[0][2] = (M[0][1] * M[1][2] - M[0][2] * M[1][1]) / detM;
        inv[1][2] = (M[0][2] * M[1][0] - M[0][0] * M[1][2]) / detM;
        inv[2][2] = (M[0][0] * M[1][1] - M[0][1] * M[1][0]) / detM;
        float posv[3];
        for (int r = 0; r < 3; ++r) {
            float acc = 0.f;
            for (int i = 0; i < 16; ++i) {
                const float pr = (float)(inv[r][0] * (double)A[i][0] + inv[r][1] * (double)A[i][1] + inv[r][2] * (double)A[i][2]);  // pinv = (inv . A^T).float()
                acc += pr * rhs[i];
            }
            posv[r] = acc;
        }
        posv[0] -= tx / fx;
        const float z3d = posv[2];
        const float cx3d = (posv[0] * fx + tx + cx * z3d) / z3d;
        const float cy3d = (posv[1] * fy + ty + cy * z3d) / z3d;
        det[k][0] = fmaxf(bb[0], 0.0f); det[k][1] = fmaxf(bb[1], 0.0f);
        det[k][2] = fminf(bb[2], (float)p.img_w); det[k][3] = fminf(bb[3], (float)p.img_h);
        det[k][4] = cx3d; det[k][5] = cy3d; det[k][6] = z3d; det[k][7] = dm[0]; det[k][8] = dm[1]; det[k][9] = dm[2]; det[k][10] = alpha;
        keep[k] = s_score[k] > p.score_thr;
    }
    __syncthreads();
    // (f) score mask (order preserving; scores are already in decreasing order) + class-agnostic NMS (:296-311)
    compact_positions(keep, K, pos, scratch, &total);
    const int Kc = total;
    __syncthreads();
    for (int k = tid; k < K; k += blockDim.x)
        if (keep[k]) cidx[pos[k]] = k;
    __syncthreads();
    auto box = [&](int j) -> f32x4 {
        const int k = cidx[j];
        f32x4 r = {det[k][0], det[k][1], det[k][2], det[k][3]};
        return r;
    };
    nms_sorted(box, Kc, p.nms_thr, alive, chunk_box, chunk_area, chunk_alive, chunk_mask);
    compact_positions(alive, Kc, pos, scratch, &total);
    const int kept = total;
    for (int j = tid; j < Kc; j += blockDim.x) {
        if (!alive[j]) continue;
        const int k = cidx[j], o = pos[j];
        const int64_t ob = (int64_t)b * K + o;
        for (int e = 0; e < 11; ++e) p.out_boxes[ob * 11 + e] = det[k][e];
        p.out_scores[ob] = s_score[k];
        p.out_cls[ob] = s_cls[k];
    }
    if (tid == 0) p.out_count[b] = kept;
}

constexpr int kTopkLds = 8192;        // key lists up to this many peaks are sorted in LDS
struct Ws {
    int32_t* peak_count; float* peak_score; int32_t* peak_idx; float* top_score; int32_t* top_idx; int32_t* overflow; uint64_t* big_keys;
};
inline int64_t ws_bytes(int B, int nch, int max_peaks, int K) {
    return 1024 + (int64_t)B * nch * 4 + (int64_t)B * 4 + (int64_t)B * nch * max_peaks * 8 + (int64_t)B * nch * K * 8 +
           (max_peaks > kTopkLds ? 256 + (int64_t)B * nch * max_peaks * 8 : 0);
}
inline Ws carve(void* base, int B, int nch, int max_peaks, int K) {
    Ws w;
    char* p = (char*)base;
    w.peak_count = (int32_t*)p; p += ((int64_t)B * nch * 4 + 255) / 256 * 256;
    w.overflow = (int32_t*)p; p += ((int64_t)B * 4 + 255) / 256 * 256;
    w.peak_score = (float*)p; p += (int64_t)B * nch * max_peaks * 4;
    w.peak_idx = (int32_t*)p; p += (int64_t)B * nch * max_peaks * 4;
    w.top_score = (float*)p; p += (int64_t)B * nch * K * 4;
    w.top_idx = (int32_t*)p; p += (int64_t)B * nch * K * 4;
    w.big_keys = max_peaks > kTopkLds ? (uint64_t*)(((uintptr_t)p + 255) & ~(uintptr_t)255) : nullptr;
    return w;
}

}  // namespace

extern "C" int64_t vd3d_km3d_workspace_bytes(int B, int n_cls, int n_joints, int max_peaks, int K) {
    return ws_bytes(B, n_cls + n_joints, max_peaks, K);
}

extern "C" int vd3d_km3d_decode(const vd3d_km3d_params* q, void* stream) {
    if (!q || !q->hm || !q->wh || !q->hps || !q->rot || !q->dim || !q->prob || !q->reg || !q->hm_hp || !q->hp_offset || !q->P2 ||
        !q->kconst || !q->workspace || !q->out_scores || !q->out_boxes || !q->out_cls || !q->out_count) {
        vd3d_set_error("km3d_decode: null pointer");
        return VD3D_EINVAL;
    }
    if (q->n_joints != kMaxJ || q->K < 1 || q->K > kMaxK || q->n_cls < 1 || q->n_cls * q->K > 512 || q->max_peaks < q->K ||
        q->max_peaks > (1 << 24) || (q->max_peaks & (q->max_peaks - 1))) {
        vd3d_set_error("km3d_decode: need 9 joints, K <= 128, n_cls*K <= 512, max_peaks a power of two in [K, 2^24]");
        return VD3D_EINVAL;
    }
    // validated with the other parameters, BEFORE anything is enqueued (include/vd3d.h states the limit)
    if (q->n_cls > kPeakMaxC || q->n_joints > kPeakMaxC) { vd3d_set_error("km3d_decode: more than 9 heat-map channels per map"); return VD3D_ERANGE; }
    hipStream_t s = (hipStream_t)stream;
    const int nch = q->n_cls + q->n_joints;
    Ws w = carve(q->workspace, q->B, nch, q->max_peaks, q->K);
    KArgs a;
    a.hm = q->hm; a.wh = q->wh; a.hps = q->hps; a.rot = q->rot; a.dim = q->dim; a.prob = q->prob; a.reg = q->reg; a.hm_hp = q->hm_hp;
    a.hp_offset = q->hp_offset; a.P2 = q->P2; a.kconst = q->kconst;
    a.B = q->B; a.H = q->H; a.W = q->W; a.n_cls = q->n_cls; a.J = q->n_joints; a.K = q->K; a.max_peaks = q->max_peaks;
    a.img_h = q->img_h; a.img_w = q->img_w; a.score_thr = q->score_thr; a.nms_thr = q->nms_iou_thr;
    a.peak_count = w.peak_count; a.peak_score = w.peak_score; a.peak_idx = w.peak_idx; a.top_score = w.top_score; a.top_idx = w.top_idx;
    a.overflow = w.overflow;
    a.big_keys = w.big_keys;
    a.out_scores = q->out_scores; a.out_boxes = q->out_boxes; a.out_cls = q->out_cls; a.out_count = q->out_count;
    const int nz = (int)(((int64_t)q->B * nch * 4 + 255) / 256 * 256 + (int64_t)q->B * 4) / 4;
    hipLaunchKernelGGL(km3d_zero_kernel, dim3((nz + 255) / 256), dim3(256), 0, s, w.peak_count, nz);
    {
        const int tiles_x = (q->W + kPeakTW - 1) / kPeakTW, tiles_y = (q->H + kPeakTH - 1) / kPeakTH;
        const int cmax = q->n_cls > q->n_joints ? q->n_cls : q->n_joints;
        const int peaks_lds = kPeakRing * cmax * 4;                       // the tile's sigmoid ring
        const int c0 = q->n_cls > 0 ? q->n_cls : 1, c1 = q->n_joints > 0 ? q->n_joints : 1;
        hipLaunchKernelGGL(km3d_peaks_kernel, dim3((unsigned)(tiles_x * tiles_y), (unsigned)q->B, 2), dim3(256), peaks_lds, s, a, tiles_x,
                           make_fastdiv((uint32_t)tiles_x), make_fastdiv((uint32_t)((kPeakTW + 2) * c0)), make_fastdiv((uint32_t)c0),
                           make_fastdiv((uint32_t)((kPeakTW + 2) * c1)), make_fastdiv((uint32_t)c1));
    }
    int rc = vd3d_check_launch("km3d_peaks");
    if (rc) return rc;
    const int lds = ((q->max_peaks > kTopkLds ? 0 : q->max_peaks) + kTopkSort) * 8;      // the key list (when it fits) + the compacted candidates of the radix select
    static Vd3dLdsLimit lim;
    rc = vd3d_raise_lds_limit((const void*)km3d_topk_kernel, lds, lim, "hipFuncSetAttribute(km3d_topk)");
    if (rc) return rc;
    hipLaunchKernelGGL(km3d_topk_kernel, dim3(nch, q->B), dim3(kNmsThreads), lds, s, a);
    rc = vd3d_check_launch("km3d_topk");
    if (rc) return rc;
    hipLaunchKernelGGL(km3d_decode_kernel, dim3(q->B), dim3(256), 0, s, a);
    return vd3d_check_launch("km3d_decode");
}
