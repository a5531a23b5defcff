// postprocess.hip -- anchor-head post-processing and NMS for gfx950.
//
// Replaces, per sample and without any host round trip:
//   heads/detection_3d_head.py:341-400 get_bboxes (sigmoid, ground-filter mask, score threshold, argmax),
//   :218-263 _decode, networks/utils/utils.py:186-196 ClipBoxes, heads/anchors.py:99-111 (ground filter),
//   and torchvision.ops.nms (third party; semantics restated in oracle/nms_ref.py).
// The reference does this with ~8 boolean-index ops (each a device->host sync) at batch 1 only.
//
// Numerics: everything here is plain fp32 with FMA contraction OFF and the same operation order as the
// reference's torch ops, so that the discrete decisions (mask, threshold, NMS suppression) agree.
// NMS keep order = decreasing score, ties -> lower index (stable), exactly the oracle's.
#pragma clang fp contract(off)
#include "common.h"
#include "nms_common.h"

namespace {

constexpr int kSelThreads = 256;
constexpr int kMaxSort = 8192;  // LDS bitonic sort capacity (64 KiB of keys); candidate capacities beyond it sort in global memory (the overflow path)

VD3D_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct Workspace {  // per-sample slices of one flat buffer
    int32_t* count;       // [B]
    int32_t* cand_idx;    // [B][max_cand]
    float* cand_score;    // [B][max_cand]
    float* boxes;         // [B][max_cand][11]  (sorted, decoded, compacted)
    float* scores;        // [B][max_cand]
    int32_t* labels;      // [B][max_cand]
    int32_t* anchor;      // [B][max_cand]
    char* big;            // max_cand > kMaxSort only: [B][big_bytes(max_cand)] -- the sort keys / positions / flags that otherwise live in LDS
};
// The reference's get_bboxes has NO cap on the candidate list (heads/detection_3d_head.py:341-400: boolean indexing, then nms).  Capacities up to
// kMaxSort keep the per-frame lists in LDS (the production path); a larger capacity -- the caller's retry after an overflow, up to every anchor of the
// frame -- runs the very same steps with those lists in global memory, one workgroup per frame.
__host__ __device__ inline int64_t big_bytes(int max_cand) {
    return max_cand > kMaxSort ? ((int64_t)max_cand * (8 + 4 + 4 + 1 + 1) + 255) / 256 * 256 : 0;
}
__host__ __device__ inline int64_t ws_bytes(int B, int max_cand) {
    return 256 + (int64_t)B * 4 + (int64_t)B * max_cand * (4 + 4 + 44 + 4 + 4 + 4) + 256 + 256 + (int64_t)B * big_bytes(max_cand);
}
__host__ __device__ inline Workspace carve(void* base, int B, int max_cand) {
    Workspace w;
    char* p = (char*)base;
    w.count = (int32_t*)p; p += ((int64_t)B * 4 + 255) / 256 * 256;
    const int64_t n = (int64_t)B * max_cand;
    w.cand_idx = (int32_t*)p; p += n * 4;
    w.cand_score = (float*)p; p += n * 4;
    w.boxes = (float*)p; p += n * 44;
    w.scores = (float*)p; p += n * 4;
    w.labels = (int32_t*)p; p += n * 4;
    w.anchor = (int32_t*)p; p += n * 4;
    w.big = max_cand > kMaxSort ? (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255) : nullptr;
    return w;
}

struct HeadArgs {
    const float* cls; const float* reg; const float* anchors; const float* prior; const float* P2;
    int B, N, A, n_cls, n_types, img_h, img_w;
    float score_thr, nms_thr, y_min, y_max, x_max;
    int use_filter, max_cand, max_det;
    Workspace ws;
    float* out_scores; float* out_boxes; int32_t* out_labels; int32_t* out_anchor; int32_t* out_count;
};

// counters are cleared by a kernel (not hipMemsetAsync): a memset issued on a capturing stream was observed not to be
// replayed by the hipGraph on ROCm 7.2, which let the candidate counters accumulate across replays.
__global__ void zero_counts_kernel(int32_t* c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = 0;
}

// ---- stage 1: mask + sigmoid + threshold -> candidate list (unordered) ---------------------------------
// A workgroup owns ONE chunk of ONE sample's anchors (grid = chunks x samples, <= 64 iterations of 256 anchors) and appends all of its
// candidates with ONE atomic: pass 1 evaluates every anchor, counts the hits per wave (ballot + popcount) and remembers each lane's hits as
// one bit per iteration; the wave counts meet in LDS, thread 0 reserves the workgroup's range of the sample's list; pass 2 walks the hit
// bits again (a wave without hits in an iteration skips it) and writes (anchor, score) at base + rank.  History: one atomic per candidate
// serialised on the sample's counter (277 us at 32 x 69 120 anchors with thousands of candidates per frame); one per (wave, sample) still did:
// config 3's launch took 496 us for 26 MB of logits -- ~900 dependent atomics per counter; now 16 per counter.
// (The list is unordered either way: stage 2 sorts it into anchor order.)
VD3D_DEV bool head_anchor_hit(const HeadArgs& p, int b, int n, float& best) {
    bool useful = true;
    if (p.use_filter) {
        // heads/anchors.py:99-111 -- both back-projections divide by fy
        const int a = n % p.A;
        const float* P = p.P2 + b * 12;
        const float fy = P[5], cy = P[6], cx = P[2];
        const f32x4 an = *(const f32x4*)(p.anchors + (int64_t)n * 4);
        const float xc = (an[0] + an[2]) / 2.0f, yc = (an[1] + an[3]) / 2.0f;
        useful = false;
        for (int t = 0; t < p.n_types; ++t) {
            const float z = p.prior[((a * p.n_types + t) * 6 + 0) * 2 + 0];
            const float x3d = (xc * z - cx * z) / fy;
            const float y3d = (yc * z - cy * z) / fy;
            useful |= (y3d > p.y_min) && (y3d < p.y_max) && (fabsf(x3d) < p.x_max);
        }
    }
    best = 0.f;
    if (!useful) return false;
    const float* c = p.cls + ((int64_t)b * p.N + n) * (p.n_cls + 1);
    best = sigmoidf_(c[0]);
    for (int k = 1; k < p.n_cls; ++k) best = fmaxf(best, sigmoidf_(c[k]));
    return best > p.score_thr;
}

__global__ void __launch_bounds__(kSelThreads) head_select_kernel(const HeadArgs p, int chunk) {
    __shared__ int wsum[kSelThreads / 64];
    __shared__ int wg_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * chunk;
    const int n1 = n0 + chunk < p.N ? n0 + chunk : p.N;
    const int iters = (n1 - n0 + kSelThreads - 1) / kSelThreads;           // <= 64 (host)
    uint64_t mine = 0;                                                      // bit `it`: this lane's anchor of iteration `it` is a candidate
    int wcount = 0;
    for (int it = 0; it < iters; ++it) {
        const int n = n0 + it * kSelThreads + tid;
        float best;
        const bool hit = n < n1 && head_anchor_hit(p, b, n, best);
        wcount += __builtin_popcountll(__builtin_amdgcn_ballot_w64(hit));
        if (hit) mine |= 1ull << it;
    }
    if (lane == 0) wsum[wave] = wcount;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < kSelThreads / 64; ++w) tot += wsum[w];
        wg_base = tot ? atomicAdd(p.ws.count + b, tot) : 0;                 // ONE atomic per workgroup, none without candidates
    }
    __syncthreads();
    int run = wg_base;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    if (wcount == 0) return;                                                // (wave-uniform)
    for (int it = 0; it < iters; ++it) {
        const bool hit = (mine >> it) & 1ull;
        const uint64_t grp = __builtin_amdgcn_ballot_w64(hit);
        if (!grp) continue;
        if (hit) {
            const int n = n0 + it * kSelThreads + tid;
            float best;
            head_anchor_hit(p, b, n, best);                                 // the same arithmetic again: the score of a hit (its bits are the list's)
            const int pos = run + __builtin_popcountll(grp & ((1ull << lane) - 1ull));
            if (pos < p.max_cand) {
                p.ws.cand_idx[(int64_t)b * p.max_cand + pos] = n;
                p.ws.cand_score[(int64_t)b * p.max_cand + pos] = best;
            }
        }
        run += __builtin_popcountll(grp);
    }
}

// decode of ONE candidate (heads/detection_3d_head.py:218-263) + ClipBoxes (utils.py:186-196): 11 box fields -> o[0..10], its score,
// label and "prior z-mean > 0".  Shared by the block-wide path and the single-wave path below: identical arithmetic.
VD3D_DEV void decode_candidate(const HeadArgs& p, int b, int n, float* o, float& best, int& label, bool& zok) {
    const int nc1 = p.n_cls + 1;
    const int a = n % p.A;
    const float* c = p.cls + ((int64_t)b * p.N + n) * nc1;
    best = sigmoidf_(c[0]);
    label = 0;
    for (int k = 1; k < p.n_cls; ++k) {
        const float s = sigmoidf_(c[k]);
        if (s > best) { best = s; label = k; }
    }
    const float alpha_score = sigmoidf_(c[p.n_cls]);
    const float* ms = p.prior + ((a * p.n_types + label) * 6) * 2;  // [6][2] = (mean, std)
    const f32x4 an = *(const f32x4*)(p.anchors + (int64_t)n * 4);
    const float* d = p.reg + ((int64_t)b * p.N + n) * 12;
    const float w = an[2] - an[0], h = an[3] - an[1];
    const float cx = an[0] + 0.5f * w, cy = an[1] + 0.5f * h;
    const float pcx = cx + (d[0] * 0.1f) * w, pcy = cy + (d[1] * 0.1f) * h;
    const float pw = expf(d[2] * 0.2f) * w, ph = expf(d[3] * 0.2f) * h;
    float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
    const float c3x = cx + (d[4] * 0.1f) * w, c3y = cy + (d[5] * 0.1f) * h;
    const float z = d[6] * ms[1] + ms[0];
    const float s2 = d[7] * ms[3] + ms[2];
    const float c2 = d[8] * ms[5] + ms[4];
    const float w3 = d[9] * ms[7] + ms[6];
    const float h3 = d[10] * ms[9] + ms[8];
    const float l3 = d[11] * ms[11] + ms[10];
    float alpha = atan2f(s2, c2) / 2.0f;
    if (alpha_score < 0.5f) alpha += 3.14159265358979323846f;
    if (p.img_w > 0) {    // ClipBoxes (networks/utils/utils.py:186-196); img_w <= 0: the reference's `img_batch is None` (no clipping)
        x1 = fmaxf(x1, 0.0f); y1 = fmaxf(y1, 0.0f);
        x2 = fminf(x2, (float)p.img_w); y2 = fminf(y2, (float)p.img_h);
    }
    o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = c3x; o[5] = c3y; o[6] = z; o[7] = w3; o[8] = h3; o[9] = l3; o[10] = alpha;
    zok = ms[0] > 0.0f;
}

// ---- stage 2, single-wave path: at most 256 candidates (the common case: a frame has a few dozen to ~200) -----------------------
// The block-wide path below costs ~70 workgroup barriers of 16 waves (two bitonic sorts, two compactions, chunked NMS): 52 us per
// step for a dozen boxes.  With <= 256 candidates ONE wave does the same steps with no barrier at all: lane l owns list slots
// l, l + 64, l + 128, l + 192; both orderings are RANK computations (keys are unique: anchor index / (score, list position)): every
// lane counts the smaller keys while the key list streams through LDS broadcast reads; compactions are ballots + popcounts; NMS walks
// the sorted list once, paying the IoU tests only for boxes that are still alive (= the kept ones).  Same decode function, same
// comparisons, same tie rule -> the same detections in the same order as the block-wide path and the oracle.
constexpr int kWaveCand = 256, kWaveR = kWaveCand / 64;
__device__ void head_nms_wave(const HeadArgs& p, int b, int K, char* smem) {
    const int lane = threadIdx.x;                          // wave 0 only
    uint64_t* keys2 = (uint64_t*)smem;                     // [256]
    f32x4* sbox = (f32x4*)(keys2 + kWaveCand);             // [256]   2D boxes in NMS order
    float* lbox = (float*)(sbox + kWaveCand);              // [256][11] decoded boxes, anchor order
    float* lscore = lbox + kWaveCand * 11;                 // [256]
    int* llabel = (int*)(lscore + kWaveCand);              // [256]
    int* lanchor = llabel + kWaveCand;                     // [256]
    uint32_t* list = (uint32_t*)(lanchor + kWaveCand);     // [256]   candidate anchors, list order
    int* sorted = (int*)(list + kWaveCand);                // [256]   anchor order
    int* perm = sorted + kWaveCand;                        // [256]   NMS position -> anchor-order slot
    int* qof = perm + kWaveCand;                           // [256]   anchor-order slot -> filtered position
    unsigned char* alive = (unsigned char*)(qof + kWaveCand);   // [256]
    const int64_t cb = (int64_t)b * p.max_cand;
    const uint64_t lt = (1ull << lane) - 1ull;
    // 1. anchor order
    uint32_t n0[kWaveR];
#pragma unroll
    for (int r = 0; r < kWaveR; ++r) {
        const int i = lane + 64 * r;
        n0[r] = i < K ? (uint32_t)p.ws.cand_idx[cb + i] : 0xffffffffu;
        list[i] = n0[r];
    }
    int rk[kWaveR] = {0, 0, 0, 0};
    for (int j = 0; j < K; ++j) {
        const uint32_t v = list[j];                        // same address in every lane: an LDS broadcast
#pragma unroll
        for (int r = 0; r < kWaveR; ++r) rk[r] += v < n0[r];
    }
#pragma unroll
    for (int r = 0; r < kWaveR; ++r)
        if (lane + 64 * r < K) sorted[rk[r]] = (int)n0[r];
    // 2. decode slot i = lane + 64 r of the anchor-ordered list; 3. z-prior filter (order preserving): filtered position q
    float best[kWaveR];
    bool zok[kWaveR];
    int qbase = 0;
#pragma unroll
    for (int r = 0; r < kWaveR; ++r) {
        const int i = lane + 64 * r;
        best[r] = 0.f;
        zok[r] = false;
        if (i < K) {
            const int n = sorted[i];
            int label;
            decode_candidate(p, b, n, lbox + i * 11, best[r], label, zok[r]);
            lscore[i] = best[r];
            llabel[i] = label;
            lanchor[i] = n;
        }
        const uint64_t fm = __builtin_amdgcn_ballot_w64(i < K && zok[r]);
        const int q = qbase + __builtin_popcountll(fm & lt);
        qbase += __builtin_popcountll(fm);
        if (i < K && zok[r]) {
            qof[i] = q;
            keys2[q] = ((uint64_t)orderable_desc(best[r]) << 32) | (uint32_t)q;      // score descending, then list position
            perm[q] = i;                                                             // (for now: filtered position -> slot)
        }
    }
    const int Kc = qbase;
    // 4. NMS order: rank of every filtered candidate's key
    uint64_t myk[kWaveR];
    int rk2[kWaveR] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < kWaveR; ++r) myk[r] = lane + 64 * r < Kc ? keys2[lane + 64 * r] : ~0ull;      // lane owns filtered positions q = lane + 64 r
    for (int j = 0; j < Kc; ++j) {
        const uint64_t v = keys2[j];
#pragma unroll
        for (int r = 0; r < kWaveR; ++r) rk2[r] += v < myk[r];
    }
    int slot_of[kWaveR];
#pragma unroll
    for (int r = 0; r < kWaveR; ++r) slot_of[r] = lane + 64 * r < Kc ? perm[lane + 64 * r] : 0;
#pragma unroll
    for (int r = 0; r < kWaveR; ++r) {
        if (lane + 64 * r < Kc) {
            const float* o = lbox + slot_of[r] * 11;
            sbox[rk2[r]] = f32x4{o[0], o[1], o[2], o[3]};
        }
    }
    // perm is rewritten as NMS position -> anchor-order slot (every lane has read its old entries above: same wave, program order)
#pragma unroll
    for (int r = 0; r < kWaveR; ++r)
        if (lane + 64 * r < Kc) perm[rk2[r]] = slot_of[r];
    // greedy NMS over the sorted list: lane owns sorted positions j = lane + 64 r
    f32x4 bj[kWaveR];
    float aj[kWaveR];
    bool a[kWaveR];
#pragma unroll
    for (int r = 0; r < kWaveR; ++r) {
        const int j = lane + 64 * r;
        a[r] = j < Kc;
        bj[r] = a[r] ? sbox[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        aj[r] = (bj[r][2] - bj[r][0]) * (bj[r][3] - bj[r][1]);
        alive[j] = a[r];
    }
    for (int i = 0; i < Kc; ++i) {
        if (!alive[i]) continue;                           // broadcast read, wave-uniform
        const f32x4 bi = sbox[i];
        const float areai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
#pragma unroll
        for (int r = 0; r < kWaveR; ++r) {
            const int j = lane + 64 * r;
            if (j > i && a[r] && iou_gt(bi, areai, bj[r], aj[r], p.nms_thr)) { a[r] = false; alive[j] = 0; }
        }
    }
    // output compaction in sorted order
    int obase = 0, kept = 0;
    uint64_t km[kWaveR];
#pragma unroll
    for (int r = 0; r < kWaveR; ++r) { km[r] = __builtin_amdgcn_ballot_w64(a[r]); kept += __builtin_popcountll(km[r]); }
    const int nout = min(kept, p.max_det);
#pragma unroll
    for (int r = 0; r < kWaveR; ++r) {
        const int o_ = obase + __builtin_popcountll(km[r] & lt);
        obase += __builtin_popcountll(km[r]);
        if (a[r] && o_ < nout) {
            const int src = perm[lane + 64 * r];
            const int64_t ob = (int64_t)b * p.max_det + o_;
#pragma unroll
            for (int e = 0; e < 11; ++e) p.out_boxes[ob * 11 + e] = lbox[src * 11 + e];
            p.out_scores[ob] = lscore[src];
            p.out_labels[ob] = llabel[qof[src]];           // the reference's unfiltered-label quirk: label of the q-th UNFILTERED candidate
            p.out_anchor[ob] = lanchor[src];
        }
    }
    if (lane == 0) p.out_count[b] = kept > p.max_det ? -2 : kept;
}

// ---- stage 2: per-sample decode + clip + z-mask + score sort + NMS -----------------------------------
// Order of operations mirrors get_bboxes: candidates in ANCHOR order -> decode -> z-prior mask -> nms (which
// sorts by score, stable) -> outputs in decreasing-score order.
// QUIRK reproduced from the reference (detection_3d_head.py:375-379,392-394): bboxes / max_score are filtered by
// the z-prior mask but `label` is not, and is then indexed with keep indices of the FILTERED list.  The label
// reported for the detection at filtered position q is therefore the label of the q-th UNFILTERED candidate.
__global__ void __launch_bounds__(kNmsThreads) head_nms_kernel(const HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const int cnt = p.ws.count[b];
    if (cnt > p.max_cand) {
        if (threadIdx.x == 0) p.out_count[b] = -1;
        return;
    }
    const int K = cnt;
    if (K <= kWaveCand) {                                  // block-uniform
        if (threadIdx.x < 64) head_nms_wave(p, b, K, smem);
        return;
    }
    int P = 1;
    while (P < K) P <<= 1;
    // LDS carve (capacities beyond kMaxSort: the per-candidate lists live in the frame's slice of the global workspace instead; __syncthreads
    // orders a workgroup's global accesses like its LDS accesses, so every step below is unchanged)
    const bool big = p.max_cand > kMaxSort;
    const int lcap = big ? 0 : p.max_cand;                             // list entries carved from LDS
    char* gbase = big ? p.ws.big + (int64_t)b * big_bytes(p.max_cand) : nullptr;
    uint64_t* keys = big ? (uint64_t*)gbase : (uint64_t*)smem;         // [max_cand]
    int* pos = big ? (int*)(gbase + (int64_t)p.max_cand * 8) : (int*)(keys + p.max_cand);               // [max_cand]
    int* cidx = pos + p.max_cand;                                      // [max_cand] filtered position q -> anchor-order p
    int* scratch = big ? (int*)smem : cidx + p.max_cand;               // [kNmsThreads]
    f32x4* chunk_box = (f32x4*)(scratch + kNmsThreads);                // [64]
    float* chunk_area = (float*)(chunk_box + 64);                      // [64]
    unsigned char* flag = big ? (unsigned char*)(cidx + p.max_cand) : (unsigned char*)(chunk_area + 64);   // [max_cand]
    unsigned char* alive = flag + p.max_cand;                          // [max_cand]
    unsigned char* chunk_alive = big ? (unsigned char*)(chunk_area + 64) : alive + lcap;                // [64]
    int* total = (int*)(chunk_alive + 64);
    uint64_t* chunk_mask = (uint64_t*)(((uintptr_t)(total + 4) + 15) & ~(uintptr_t)15);      // [64] rows of the chunk's suppression matrix
    f32x4* sbox = p.max_cand <= 4096 ? (f32x4*)(chunk_mask + 64) : nullptr;   // [max_cand] 2-D boxes in NMS order (when they fit; else from global)

    const int64_t cb = (int64_t)b * p.max_cand;
    // 1. anchor order (boolean-mask indexing in the reference preserves anchor order)
    for (int i = threadIdx.x; i < P; i += blockDim.x) keys[i] = i < K ? (uint64_t)(uint32_t)p.ws.cand_idx[cb + i] : ~0ull;
    __syncthreads();
    bitonic_sort(keys, P);

    // 2. decode (heads/detection_3d_head.py:218-263) + clip (utils.py:186-196); flag = prior z-mean > 0
    float* tbox = p.ws.boxes + cb * 11;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const int n = (int)(uint32_t)keys[i];
        float best;
        int label;
        bool zok;
        decode_candidate(p, b, n, tbox + (int64_t)i * 11, best, label, zok);
        p.ws.scores[cb + i] = best;
        p.ws.labels[cb + i] = label;
        p.ws.anchor[cb + i] = n;
        flag[i] = zok;
    }
    __syncthreads();
    // 3. z-prior filter (order preserving)
    compact_positions(flag, K, pos, scratch, total);
    const int Kc = *total;
    int P2 = 1;
    while (P2 < Kc) P2 <<= 1;
    __syncthreads();
    for (int i = threadIdx.x; i < P2; i += blockDim.x) keys[i] = ~0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        if (!flag[i]) continue;
        const int q = pos[i];
        cidx[q] = i;
        keys[q] = ((uint64_t)orderable_desc(p.ws.scores[cb + i]) << 32) | (uint32_t)q;  // score desc, then list order
    }
    __syncthreads();
    // 4. NMS order
    bitonic_sort(keys, P2);
    auto box = [&](int j) -> f32x4 {
        const float* o = tbox + (int64_t)cidx[(uint32_t)keys[j]] * 11;
        f32x4 r = {o[0], o[1], o[2], o[3]};
        return r;
    };
    if (sbox) {
        // the boxes once, in NMS order, into LDS: the greedy pass reads every live box once per chunk -- through two indirections into
        // global memory that was a dependent L2 round trip per chunk and thread
        for (int j = threadIdx.x; j < Kc; j += blockDim.x) sbox[j] = box(j);
        __syncthreads();
        nms_sorted([&](int j) -> f32x4 { return sbox[j]; }, Kc, p.nms_thr, alive, chunk_box, chunk_area, chunk_alive, chunk_mask);
    } else {
        nms_sorted(box, Kc, p.nms_thr, alive, chunk_box, chunk_area, chunk_alive, chunk_mask);
    }
    compact_positions(alive, Kc, pos, scratch, total);
    const int kept = *total;
    const int nout = min(kept, p.max_det);
    for (int j = threadIdx.x; j < Kc; j += blockDim.x) {
        if (!alive[j] || pos[j] >= nout) continue;
        const int q = (int)(uint32_t)keys[j];
        const int i = cidx[q], o_ = pos[j];
        const int64_t ob = (int64_t)b * p.max_det + o_;
        for (int e = 0; e < 11; ++e) p.out_boxes[ob * 11 + e] = tbox[(int64_t)i * 11 + e];
        p.out_scores[ob] = p.ws.scores[cb + i];
        p.out_labels[ob] = p.ws.labels[cb + q];  // the reference's unfiltered-label quirk (see above)
        p.out_anchor[ob] = p.ws.anchor[cb + i];
    }
    if (threadIdx.x == 0) p.out_count[b] = kept > p.max_det ? -2 : kept;
}

// ---- standalone torchvision.ops.nms replacement --------------------------------------------------------
__global__ void __launch_bounds__(kNmsThreads) nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                          int n, float thr, int32_t* keep, int32_t* count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int P = 1;
    while (P < n) P <<= 1;
    uint64_t* keys = (uint64_t*)smem;                      // [P]
    unsigned char* alive = (unsigned char*)(keys + P);     // [n]
    int* pos = (int*)(alive + ((n + 15) / 16) * 16);       // [n]
    int* scratch = pos + n;                                // [kNmsThreads]
    f32x4* chunk_box = (f32x4*)(scratch + kNmsThreads);
    float* chunk_area = (float*)(chunk_box + 64);
    unsigned char* chunk_alive = (unsigned char*)(chunk_area + 64);
    int* total = (int*)(chunk_alive + 64);
    uint64_t* chunk_mask = (uint64_t*)(((uintptr_t)(total + 4) + 15) & ~(uintptr_t)15);
    for (int i = threadIdx.x; i < P; i += blockDim.x)
        keys[i] = i < n ? (((uint64_t)orderable_desc(scores[i]) << 32) | (uint32_t)i) : ~0ull;
    __syncthreads();
    bitonic_sort(keys, P);
    auto box = [&](int j) -> f32x4 { return *(const f32x4*)(boxes + (int64_t)(uint32_t)keys[j] * 4); };
    nms_sorted(box, n, thr, alive, chunk_box, chunk_area, chunk_alive, chunk_mask);
    compact_positions(alive, n, pos, scratch, total);
    for (int j = threadIdx.x; j < n; j += blockDim.x)
        if (alive[j]) keep[pos[j]] = (int32_t)(uint32_t)keys[j];
    if (threadIdx.x == 0) *count = *total;
}

inline int64_t head_nms_lds(int max_cand) {
    if (max_cand > kMaxSort) max_cand = 0;        // the overflow path keeps its per-candidate lists in global memory: scratch + chunk buffers only
    const int64_t block_path = (int64_t)max_cand * 8 + (int64_t)max_cand * 8 + kNmsThreads * 4 + 64 * 16 + 64 * 4 + max_cand * 2 + 64 + 16 + 32 + 64 * 8 +
                               (max_cand <= 4096 ? (int64_t)max_cand * 16 : 0);      // + chunk masks + (when they fit) the boxes in NMS order
    const int64_t wave_path = kWaveCand * (8 + 16 + 44 + 4 * 7 + 1) + 64;      // head_nms_wave's carve (24.3 KB)
    return block_path > wave_path ? block_path : wave_path;
}
inline int64_t nms_lds(int n) {
    int P = 1;
    while (P < n) P <<= 1;
    return (int64_t)P * 8 + ((n + 15) / 16) * 16 + (int64_t)n * 4 + kNmsThreads * 4 + 64 * 16 + 64 * 4 + 64 + 16 + 32 + 64 * 8;   // (+ chunk masks)
}

}  // namespace

extern "C" int64_t vd3d_head_workspace_bytes(int B, int max_cand) { return ws_bytes(B, max_cand); }

static int head_args(const vd3d_head_params* q, HeadArgs& a, bool need_select, bool need_nms) {
    if (!q || !q->cls || !q->anchors || !q->prior_mean_std || !q->P2 || !q->workspace ||
        (need_nms && (!q->reg || !q->out_scores || !q->out_boxes || !q->out_labels || !q->out_anchor || !q->out_count))) {
        vd3d_set_error("head_postprocess: null pointer");
        return VD3D_EINVAL;
    }
    (void)need_select;
    if (q->B <= 0 || q->N <= 0 || q->A <= 0 || q->N % q->A || q->n_cls < 1 || q->n_types < q->n_cls ||
        q->max_cand < 1 || q->max_cand > (1 << 24) || (q->max_cand & (q->max_cand - 1)) || q->max_det < 1 ||
        ((uintptr_t)q->anchors & 15)) {
        vd3d_set_error("head_postprocess: bad sizes (max_cand must be a power of two <= 2^24)");
        return VD3D_EINVAL;
    }
    a.cls = q->cls; a.reg = q->reg; a.anchors = q->anchors; a.prior = q->prior_mean_std; a.P2 = q->P2;
    a.B = q->B; a.N = q->N; a.A = q->A; a.n_cls = q->n_cls; a.n_types = q->n_types; a.img_h = q->img_h; a.img_w = q->img_w;
    a.score_thr = q->score_thr; a.nms_thr = q->nms_iou_thr; a.y_min = q->filter_y_min; a.y_max = q->filter_y_max; a.x_max = q->filter_x_max;
    a.use_filter = q->use_filter; a.max_cand = q->max_cand; a.max_det = q->max_det;
    a.ws = carve(q->workspace, q->B, q->max_cand);
    a.out_scores = q->out_scores; a.out_boxes = q->out_boxes; a.out_labels = q->out_labels; a.out_anchor = q->out_anchor; a.out_count = q->out_count;
    return VD3D_OK;
}

// stage 1 alone: ground filter + sigmoid + threshold -> the per-sample candidate lists in `workspace`.  Needs only the CLASS logits
// (reg / out_* may be NULL), so a caller whose cls tower finishes before its reg tower can run it early, on the cls tower's stream.
extern "C" int vd3d_head_select(const vd3d_head_params* q, void* stream) {
    HeadArgs a;
    if (const int rc = head_args(q, a, true, false)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (q->B > 65535) { vd3d_set_error("head_select: more than 65535 samples"); return VD3D_ERANGE; }      // (before anything is launched)
    hipLaunchKernelGGL(zero_counts_kernel, dim3((q->B + 63) / 64), dim3(64), 0, s, a.ws.count, q->B);
    // chunks x samples workgroups: ~512 in all (two per CU), a chunk a multiple of 256 anchors and at most 64 iterations of them
    int cps = (512 + q->B - 1) / q->B;
    const int max_cps = (q->N + kSelThreads - 1) / kSelThreads, min_cps = (q->N + 64 * kSelThreads - 1) / (64 * kSelThreads);
    if (cps > max_cps) cps = max_cps;
    if (cps < min_cps) cps = min_cps;
    const int chunk = ((q->N + cps - 1) / cps + kSelThreads - 1) / kSelThreads * kSelThreads;
    hipLaunchKernelGGL(head_select_kernel, dim3((unsigned)((q->N + chunk - 1) / chunk), (unsigned)q->B), dim3(kSelThreads), 0, s, a, chunk);
    return vd3d_check_launch("head_select");
}

// stage 2 alone: decode + clip + z-prior filter + NMS over the candidate lists a vd3d_head_select call with the SAME parameters left
// in `workspace` (stream-ordered after it).
extern "C" int vd3d_head_nms(const vd3d_head_params* q, void* stream) {
    HeadArgs a;
    if (const int rc = head_args(q, a, false, true)) return rc;
    const int lds = (int)head_nms_lds(q->max_cand);
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)head_nms_kernel, lds, lim, "hipFuncSetAttribute(head_nms)")) return rc;
    hipLaunchKernelGGL(head_nms_kernel, dim3(q->B), dim3(kNmsThreads), lds, (hipStream_t)stream, a);
    return vd3d_check_launch("head_nms");
}

extern "C" int vd3d_head_postprocess(const vd3d_head_params* q, void* stream) {
    if (const int rc = vd3d_head_select(q, stream)) return rc;
    return vd3d_head_nms(q, stream);
}

__global__ void pack_detections_kernel(const float* __restrict__ scores, const float* __restrict__ boxes, const int32_t* __restrict__ labels,
                                       const int32_t* __restrict__ count, int B, int K, int k, float* __restrict__ pack) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // element of this frame's [k + 1][13] block
    if (i >= (k + 1) * 13) return;
    const int r = i / 13, f = i - r * 13;
    const int c = count[b];
    float v = 0.f;
    if (r == k) v = f == 0 ? (float)c : 0.f;
    else if (r < c && r < K) {                                  // K < k is allowed (KM3D decodes K = 100 rows): rows K .. k-1 stay zero
        const int64_t src = (int64_t)b * K + r;
        v = f == 0 ? scores[src] : (f == 12 ? (float)labels[src] : boxes[src * 11 + (f - 1)]);
    }
    pack[((int64_t)b * (k + 1)) * 13 + i] = v;
}

extern "C" int vd3d_pack_detections(const float* scores, const float* boxes, const int32_t* labels, const int32_t* count, int B, int K,
                                    int k, float* pack, void* stream) {
    if (B == 0) return VD3D_OK;
    if (!scores || !boxes || !labels || !count || !pack || B < 0 || k < 0 || K < 0) { vd3d_set_error("pack_detections: bad args"); return VD3D_EINVAL; }
    const int n = (k + 1) * 13;
    hipLaunchKernelGGL(pack_detections_kernel, dim3((n + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, scores, boxes, labels, count, B, K, k, pack);
    return vd3d_check_launch("pack_detections");
}

extern "C" int64_t vd3d_nms_workspace_bytes(int n) { (void)n; return 256; }

extern "C" int vd3d_nms(const float* boxes, const float* scores, int n, float iou_thr, int32_t* keep, int32_t* count,
                        void* workspace, void* stream) {
    (void)workspace;
    hipStream_t s = (hipStream_t)stream;
    if (!keep || !count || n < 0) { vd3d_set_error("nms: bad args"); return VD3D_EINVAL; }
    if (n == 0) {
        hipLaunchKernelGGL(zero_counts_kernel, dim3(1), dim3(64), 0, s, count, 1);
        return vd3d_check_launch("nms zero");
    }
    if (!boxes || !scores || ((uintptr_t)boxes & 15) || n > kMaxSort) {
        vd3d_set_error("nms: boxes must be 16-byte aligned and n <= 8192");
        return VD3D_EINVAL;
    }
    const int lds = (int)nms_lds(n);
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)nms_kernel, lds, lim, "hipFuncSetAttribute(nms)")) return rc;
    hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(kNmsThreads), lds, s, boxes, scores, n, iou_thr, keep, count);
    return vd3d_check_launch("nms");
}
