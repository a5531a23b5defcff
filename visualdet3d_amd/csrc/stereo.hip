// stereo.hip -- stereo cost-volume kernels for gfx950 (NHWC activations).
//
//  * psm_cosine   : PSMCosineModule.forward (lib/PSM_cost_volume.py:81-96).  The reference walks the D
//                   disparities with a Python loop (slice, mul, mean, scatter per disparity: ~100 tiny
//                   kernels, L and R re-read D times).  Here one workgroup stages a 64-pixel LEFT tile and
//                   the (64 + D - 1)-pixel RIGHT window of one image row in LDS (XOR-swizzled 16-byte slots,
//                   conflict-free ds_read_b128) and produces all D disparities from that single read of
//                   L and R -- HBM traffic = read L + read R + write cost.
//  * costvol_build: concat-volume of CostVolume.forward (lib/PSM_cost_volume.py:49-64), channels-last.
//  * conv3d_3x3x3 : the two Conv3d+BN3d+ReLU of CostVolume (lib/PSM_cost_volume.py:34-41), direct fp32 FMA
//                   (C <= 16: 0.25 GFLOP per pair, not MFMA-shaped); weights are wave-uniform -> scalar loads.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int XT = 64;  // pixels per workgroup

template <typename T>
__global__ void __launch_bounds__(256) psm_cosine_kernel(const T* __restrict__ left, const T* __restrict__ right,
                                                         T* __restrict__ cost, int H, int W, int C, int D,
                                                         int ips, int ops, int xtiles, int vec_store) {
    constexpr int VE = ElemTraits<T>::kVec;
    constexpr int CHUNK = 128 / (int)sizeof(T);  // channels staged per pass (128 B per pixel)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ls = smem;                  // [XT][128 B]
    char* Rs = smem + XT * 128;       // [XT + D - 1][128 B], window pixel wp <-> image x = x0 - (D-1) + wp

    const int xt = blockIdx.x % xtiles;
    const int row = blockIdx.x / xtiles;  // b*H + y
    const int x0 = xt * XT;
    const int tid = threadIdx.x;
    const int px = tid & 63, dg = tid >> 6;  // pixel within tile, disparity group (8 disparities each)
    const int64_t rowbase = (int64_t)row * W;
    const int wpix = XT + D - 1;

    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;

    for (int c0 = 0; c0 < C; c0 += CHUNK) {
        const int cw = min(CHUNK, C - c0);  // channels in this pass (multiple of VE)
        const int nv = cw / VE;             // 16-byte vectors per pixel (<= 8)
        // ---- stage L tile and R window (zero outside the row) ----
        for (int v = tid; v < XT * 8; v += 256) {
            const int p = v >> 3, s = v & 7;
            i32x4 val = {0, 0, 0, 0};
            if (s < nv && x0 + p < W) val = *(const i32x4*)(left + (rowbase + x0 + p) * ips + c0 + s * VE);
            *(i32x4*)(Ls + p * 128 + ((s ^ ((p >> 1) & 7)) << 4)) = val;
        }
        for (int v = tid; v < wpix * 8; v += 256) {
            const int p = v >> 3, s = v & 7;
            const int x = x0 - (D - 1) + p;
            i32x4 val = {0, 0, 0, 0};
            if (s < nv && x >= 0 && x < W) val = *(const i32x4*)(right + (rowbase + x) * ips + c0 + s * VE);
            *(i32x4*)(Rs + p * 128 + ((s ^ ((p >> 1) & 7)) << 4)) = val;
        }
        __syncthreads();
        if (dg * 8 < D) {
            for (int s = 0; s < nv; ++s) {
                Vec16<T> l;
                l.raw = *(const i32x4*)(Ls + px * 128 + ((s ^ ((px >> 1) & 7)) << 4));
                float lf[VE];
#pragma unroll
                for (int e = 0; e < VE; ++e) lf[e] = l.get(e);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int d = dg * 8 + i;
                    if (d < D) {
                        const int wp = px + (D - 1) - d;
                        Vec16<T> r;
                        r.raw = *(const i32x4*)(Rs + wp * 128 + ((s ^ ((wp >> 1) & 7)) << 4));
#pragma unroll
                        for (int e = 0; e < VE; ++e) acc[i] = fmaf(lf[e], r.get(e), acc[i]);
                    }
                }
            }
        }
        __syncthreads();
    }
    // ---- mean over C, store 8 disparities per thread ----
    const int x = x0 + px;
    if (x < W && dg * 8 < D) {
        const float inv = 1.0f / (float)C;
        T* o = cost + (rowbase + x) * ops + dg * 8;
        const int nd = min(8, D - dg * 8);
        if (nd == 8 && vec_store) {
            if constexpr (sizeof(T) == 2) {
                Vec16<T> v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v.set2(e, acc[2 * e] * inv, acc[2 * e + 1] * inv);
                *(i32x4*)o = v.raw;
            } else {
                f32x4 a = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
                f32x4 b = {acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv};
                *(f32x4*)o = a;
                *(f32x4*)(o + 4) = b;
            }
        } else {
            for (int i = 0; i < nd; ++i) o[i] = ElemTraits<T>::from_f(acc[i] * inv);
        }
    }
}

// PSM cosine volume on the matrix cores (bf16, C % 32 == 0).  cost[x, d] = (1/C) L[x] . R[x - d] is a BANDED product: for 16
// consecutive pixels the needed right-image window is [x_g - 32, x_g + 15] = three 16-pixel blocks, so per 32-channel step a wave issues
// v_mfma_f32_16x16x32 (A = 16 left pixels x 32 channels, B = 32 channels x 16 window pixels) three times and keeps the 16 x 24 band of the
// 16 x 48 products (50 % of the MFMA work is discarded -- the pipe is idle anyway: the kernel streams).  Fragments come straight from
// global memory (a lane's 16 bytes are 8 channels of one pixel; the window's 3x re-read stays in L1 / L2), no LDS staging, no workgroup
// barrier, any number of waves in flight; the band leaves through a 1 KiB per-wave LDS tile as whole 16-byte vectors.
// The VALU kernel above stages L / R single-buffered behind two barriers per 64-channel chunk, leaves one wave of four idle for D = 24
// and is FMA-issue bound (2.3 TB/s at 8 x 96 x 320 x 64, 0.6-1.9 TB/s on the R50 maps); it remains the fp32 path.
__global__ void __launch_bounds__(256) psm_cosine_mfma_kernel(const short* __restrict__ left, const short* __restrict__ right,
                                                              short* __restrict__ cost, int W, int C, int D, int ips, int ops,
                                                              int groups_x, int64_t ngroups, int vec_store) {
    __shared__ __attribute__((aligned(16))) short tile[4][16][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, kg = lane >> 4;
    const float inv = 1.0f / (float)C;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < ngroups; g += (int64_t)gridDim.x * 4) {
        const int64_t row = g / groups_x;                          // b * H + y
        const int xg = (int)(g - row * groups_x) * 16;
        const int64_t rowbase = row * W;
        f32x4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool av = xg + n < W;
        const short* ap = left + (rowbase + xg + n) * ips + kg * 8;
        const short* bp[3];
        bool bv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int wpx = xg - 32 + 16 * j + n;
            bv[j] = wpx >= 0 && wpx < W;
            bp[j] = right + (rowbase + wpx) * ips + kg * 8;
        }
        for (int k0 = 0; k0 < C; k0 += 32) {
            i32x4 a = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}, b2 = {0, 0, 0, 0};
            if (av) a = *(const i32x4*)(ap + k0);
            if (bv[0]) b0 = *(const i32x4*)(bp[0] + k0);
            if (bv[1]) b1 = *(const i32x4*)(bp[1] + k0);
            if (bv[2]) b2 = *(const i32x4*)(bp[2] + k0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b0), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b1), acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b2), acc[2], 0, 0, 0);
        }
        // accumulator (col n = window pixel, rows 4 kg + e = left pixel m): disparity d = m - n + 32 - 16 j; the 16 x D band -> LDS
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * kg + e, d = m - n + 32 - 16 * j;
                if (d >= 0 && d < D) tile[wave][m][d] = f2bf(acc[j][e] * inv);     // x < d: the window pixel was out of the row -> exact 0
            }
        if (vec_store) {
            const int nvec = D >> 3;
            for (int i = lane; i < 16 * nvec; i += 64) {
                const int pxl = i / nvec, part = i - pxl * nvec;
                if (xg + pxl < W) *(i32x4*)(cost + (rowbase + xg + pxl) * ops + part * 8) = *(const i32x4*)&tile[wave][pxl][part * 8];
            }
        } else {
            for (int i = lane; i < 16 * D; i += 64) {
                const int pxl = i / D, d = i - pxl * D;
                if (xg + pxl < W) cost[(rowbase + xg + pxl) * ops + d] = tile[wave][pxl][d];
            }
        }
    }
}

template <typename T>
__global__ void costvol_build_kernel(const T* __restrict__ left, const T* __restrict__ right, T* __restrict__ vol,
                                     int B, int H, int W, int F, int D, int ips) {
    constexpr int VE = ElemTraits<T>::kVec;
    const int fv = F / VE;
    const int64_t total = (int64_t)B * D * H * W * fv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % fv) * VE;
        int64_t r = i / fv;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H); r /= H;
        const int d = (int)(r % D);
        const int b = (int)(r / D);
        i32x4 l = {0, 0, 0, 0}, rr = {0, 0, 0, 0};
        if (x >= d) {
            const int64_t pl = ((int64_t)b * H + y) * W + x;
            l = *(const i32x4*)(left + pl * ips + f);
            rr = *(const i32x4*)(right + (pl - d) * ips + f);
        }
        T* o = vol + ((((int64_t)b * D + d) * H + y) * W + x) * (2 * F);
        *(i32x4*)(o + f) = l;
        *(i32x4*)(o + F + f) = rr;
    }
}

// Direct 3x3x3 conv, channels-last; one thread per output voxel, all COUT outputs in registers.
template <typename T, int CIN, int COUT>
__global__ void __launch_bounds__(256) conv3d_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     T* __restrict__ out, int B, int D, int H, int W, int relu,
                                                     int out_fd_major, int ops) {
    constexpr int VE = ElemTraits<T>::kVec;
    const int64_t total = (int64_t)B * D * H * W;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    int64_t r = i;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H); r /= H;
    const int d = (int)(r % D);
    const int b = (int)(r / D);
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    for (int kd = 0; kd < 3; ++kd) {
        const int id = d - 1 + kd;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y - 1 + ky;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = x - 1 + kx;
                const bool ok = (unsigned)id < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const float* wt = w + ((kd * 3 + ky) * 3 + kx) * (CIN * COUT);  // wave-uniform -> s_load
                const T* src = in + ((((int64_t)b * D + id) * H + iy) * W + ix) * CIN;
#pragma unroll
                for (int cv = 0; cv < CIN / VE; ++cv) {
                    Vec16<T> v;
                    v.raw = i32x4{0, 0, 0, 0};
                    if (ok) v.raw = *(const i32x4*)(src + cv * VE);
#pragma unroll
                    for (int e = 0; e < VE; ++e) {
                        const float xv = v.get(e);
#pragma unroll
                        for (int o = 0; o < COUT; ++o) acc[o] = fmaf(xv, wt[(cv * VE + e) * COUT + o], acc[o]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
        float v = acc[o] * scale[o] + shift[o];
        acc[o] = relu ? fmaxf(v, 0.f) : v;
    }
    if (out_fd_major) {
        T* dst = out + (((int64_t)b * H + y) * W + x) * ops;
#pragma unroll
        for (int o = 0; o < COUT; ++o) dst[o * D + d] = ElemTraits<T>::from_f(acc[o]);
    } else {
        T* dst = out + i * COUT;
#pragma unroll
        for (int o = 0; o < COUT; ++o) dst[o] = ElemTraits<T>::from_f(acc[o]);
    }
}

// 3x3x3 conv on the matrix cores (bf16, COUT = 8): the volume conv of the cost-volume block is a GEMM with M = voxels,
// N = 8, K = 27 * CIN.  v_mfma_f32_16x16x32_bf16 with the weights as the A operand (rows 8..15 zero) and 16 voxels as the
// B operand: one K slice of 32 = 2 taps x 16 channels (CIN 16) or 4 taps x 8 channels (CIN 8), so lane (voxel, q) fetches
// its 16 bytes = 8 channels of ONE neighbour voxel straight from global memory -- no LDS, zero padding by predication.
// The weight fragments (14 / 7 per lane) are built once per wave in registers; waves walk 16-voxel groups grid-stride.
// The VALU kernel above (one voxel per lane, 3456 scalar FMAs each) ran at 12 % of the fp32 vector peak: 86 + 49 us per
// stereo step for 2 GFLOP; it remains the fp32 (validation mode) path.
template <int CIN>
__global__ void __launch_bounds__(256) conv3d_mfma_kernel(const short* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          short* __restrict__ out, int B, int D, int H, int W, int relu,
                                                          int out_fd_major, int ops) {
    constexpr int COUT = 8;
    constexpr int TPS = 32 / CIN;                    // taps per 32-deep K slice: 2 or 4
    constexpr int NS = (27 + TPS - 1) / TPS;         // 14 or 7 slices
    const int lane = threadIdx.x & 63, l16 = lane & 15, q = lane >> 4;
    const int my_tap_off = CIN == 16 ? (q >> 1) : q; // tap of this lane inside a slice
    const int my_ch = CIN == 16 ? 8 * (q & 1) : 0;   // first of its 8 channels
    // weight fragments: row = output channel l16 (zero for rows >= 8), k = this lane's 8 (tap, channel) values
    i32x4 wf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int tap = s * TPS + my_tap_off;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (l16 < COUT && tap < 27) ? w[(tap * CIN + my_ch + e) * COUT + l16] : 0.f;
        Vec16<short> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.set2(e, v[2 * e], v[2 * e + 1]);
        wf[s] = o.raw;
    }
    const int64_t total = (int64_t)B * D * H * W;
    const int64_t ngroups = (total + 15) / 16;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t g = wave_id; g < ngroups; g += nwaves) {
        const int64_t vox = g * 16 + l16;
        const bool vvalid = vox < total;
        int64_t r = vvalid ? vox : 0;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H); r /= H;
        const int d = (int)(r % D);
        const int b = (int)(r / D);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int tap = s * TPS + my_tap_off;
            const int kd = tap / 9, ky = (tap - kd * 9) / 3, kx = tap - kd * 9 - ky * 3;
            const int id = d - 1 + kd, iy = y - 1 + ky, ix = x - 1 + kx;
            const bool ok = vvalid && tap < 27 && (unsigned)id < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            i32x4 frag = {0, 0, 0, 0};
            if (ok) frag = *(const i32x4*)(in + ((((int64_t)b * D + id) * H + iy) * W + ix) * CIN + my_ch);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, frag), acc, 0, 0, 0);
        }
        // accumulator: lane (voxel l16, q) holds output channels 4q .. 4q+3; q >= 2 are the zero padding rows
        if (vvalid && q < 2) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = 4 * q + e;
                const float t = acc[e] * scale[o] + shift[o];
                v[e] = relu ? fmaxf(t, 0.f) : t;
            }
            if (out_fd_major) {
                short* dst = out + (((int64_t)b * H + y) * W + x) * ops;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(4 * q + e) * D + d] = f2bf(v[e]);
            } else {
                i32x2 o2;
                o2[0] = (int)((uint32_t)(uint16_t)f2bf(v[0]) | ((uint32_t)(uint16_t)f2bf(v[1]) << 16));
                o2[1] = (int)((uint32_t)(uint16_t)f2bf(v[2]) | ((uint32_t)(uint16_t)f2bf(v[3]) << 16));
                *(i32x2*)(out + vox * COUT + 4 * q) = o2;
            }
        }
    }
}

// ---- CostVolume.forward in ONE launch (bf16, PSM_features = 8) ---------------------------------------------------------
// lib/PSM_cost_volume.py:45-68 after the 1x1 down-sample: concat volume -> Conv3d + BN3d + ReLU -> Conv3d + BN3d + ReLU -> reshape.
// As three launches (costvol_build + 2 x conv3d_mfma) the stage cost ~100 us of a 3.8 ms stereo step for 2 GFLOP on a 5.9 MB volume:
// pure launch latency and dependent global round trips.  Here a workgroup owns TY x TX pixels x all D disparities:
//   phase 1: the first conv on the (TY + 2) x (TX + 2) x D haloed tile; its B operand is gathered straight from the two 8-channel
//            feature maps (vol[.., d, y, x] = L[y, x] | R[y, x - d] for x >= d, else 0 -- the concat volume is never built), results
//            (BN + ReLU, rounded to bf16: the unfused path's rounding point) go to LDS, zero outside the image (conv padding);
//   phase 2: the second conv from LDS; outputs are parked in LDS as the pixel's 8 * D-channel run (channel = f * D + d) and
//   phase 3: leave as whole 16-byte vectors (the unfused kernel wrote 2-byte pieces).
// Same MFMA formulation as conv3d_mfma_kernel: v_mfma_f32_16x16x32_bf16, weights = A operand (8 of 16 rows used), 16 voxels = B.
template <int TY, int TX>
__global__ void __launch_bounds__(256) cost_volume_fused_kernel(const short* __restrict__ left, const short* __restrict__ right,
                                                                const float* __restrict__ w1, const float* __restrict__ s1, const float* __restrict__ t1,
                                                                const float* __restrict__ w2, const float* __restrict__ s2, const float* __restrict__ t2,
                                                                short* __restrict__ out, int H, int W, int D, int ips, int ops, int xtiles, int ytiles,
                                                                int vec_out) {
    constexpr int MY = TY + 2, MX = TX + 2, F = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* mid = smem;                                  // [D][MY][MX] x 16 B (8 bf16 channels)
    short* outt = (short*)(smem + (size_t)D * MY * MX * 16);   // [TY * TX][F * D]
    float* wst = (float*)smem;                         // prologue only: the fp32 weights of both convs, staged for the fragment build
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, q = lane >> 4;
    int t = blockIdx.x;
    const int xt = t % xtiles; t /= xtiles;
    const int yt = t % ytiles;
    const int b = t / ytiles;
    const int y0 = yt * TY, x0 = xt * TX;
    const short* lb = left + (int64_t)b * H * W * ips;
    const short* rb = right + (int64_t)b * H * W * ips;
    // ---- weights: coalesced into LDS, then every lane gathers its A-fragment values (row = output channel l16, rows >= 8 zero;
    // k = this lane's 8 (tap, channel) values).  (Built straight from global memory the 168 scattered 4-byte loads per lane were
    // ~10 us of a 90 us kernel.)
    constexpr int NW1 = 27 * 16 * F, NW2 = 27 * 8 * F;
    for (int i = tid; i < NW1; i += 256) wst[i] = w1[i];
    for (int i = tid; i < NW2; i += 256) wst[NW1 + i] = w2[i];
    __syncthreads();
    i32x4 wf1[14], wf2[7];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int tap = s * 2 + (q >> 1), ch = 8 * (q & 1);
        Vec16<short> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = (l16 < F && tap < 27) ? wst[(tap * 16 + ch + 2 * e) * F + l16] : 0.f;
            const float c = (l16 < F && tap < 27) ? wst[(tap * 16 + ch + 2 * e + 1) * F + l16] : 0.f;
            o.set2(e, a, c);
        }
        wf1[s] = o.raw;
    }
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int tap = s * 4 + q;
        Vec16<short> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = (l16 < F && tap < 27) ? wst[NW1 + (tap * 8 + 2 * e) * F + l16] : 0.f;
            const float c = (l16 < F && tap < 27) ? wst[NW1 + (tap * 8 + 2 * e + 1) * F + l16] : 0.f;
            o.set2(e, a, c);
        }
        wf2[s] = o.raw;
    }
    __syncthreads();                                   // the staging area is `mid` from here on
    float sc1[4], sh1[4], sc2[4], sh2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int o = (4 * q + e) & 7;
        sc1[e] = s1[o]; sh1[e] = t1[o]; sc2[e] = s2[o]; sh2[e] = t2[o];
    }
    // this lane's tap of every K slice is FIXED (slice s, lane quarter q): its (kd, ky, kx) displacement and the element offset of
    // its source pixel relative to the voxel's own pixel are computed once, not per voxel (the kernel was integer-ALU bound)
    //   conv 1: left operand (q even): pixel (y + dy, x + dx);  right operand (q odd): pixel (y + dy, x + dx - (d + dd))
    int g1[14], o1[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int tap = s * 2 + (q >> 1);
        const int kd = tap / 9, ky = (tap - kd * 9) / 3, kx = tap - kd * 9 - ky * 3;
        const int dd = kd - 1, dy = ky - 1, dx = kx - 1;
        g1[s] = tap < 27 ? ((dd + 1) | ((dy + 1) << 2) | ((dx + 1) << 4)) : -1;
        o1[s] = (dy * W + dx - ((q & 1) ? dd : 0)) * ips;
    }
    int g2[7], o2[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int tap = s * 4 + q;
        const int kd = tap / 9, ky = (tap - kd * 9) / 3, kx = tap - kd * 9 - ky * 3;
        g2[s] = tap < 27 ? (kd - 1) : -100;
        o2[s] = (((kd - 1) * MY + ky) * MX + kx) * 16;                  // byte offset inside `mid` relative to voxel (d, ty, tx)'s halo origin
    }
    const short* srcb = (q & 1) ? rb : lb;
    // ---- phase 1: first conv on the haloed tile -> mid (LDS)
    const int nmid = D * MY * MX;
    for (int g = wave; g * 16 < nmid; g += 4) {
        const int v = g * 16 + l16;
        const bool vin = v < nmid;
        const int vv = vin ? v : 0;
        const int mx = vv % MX, my = (vv / MX) % MY, d = vv / (MX * MY);
        const int y = y0 - 1 + my, x = x0 - 1 + mx;
        const bool img = vin && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const int base = (y * W + x - ((q & 1) ? d : 0)) * ips;         // this lane's source pixel for the centre tap
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int id = d + (g1[s] & 3) - 1, iy = y + ((g1[s] >> 2) & 3) - 1, ix = x + ((g1[s] >> 4) & 3) - 1;
            const bool ok = img && g1[s] >= 0 && (unsigned)id < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && ix >= id;
            i32x4 frag = {0, 0, 0, 0};
            if (ok) frag = *(const i32x4*)(srcb + base + o1[s]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf1[s]), __builtin_bit_cast(bf16x8, frag), acc, 0, 0, 0);
        }
        if (vin && q < 2) {
            i32x2 o2v = {0, 0};
            if (img) {
                float r[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = fmaxf(acc[e] * sc1[e] + sh1[e], 0.f);
                o2v[0] = Fmt16<short>::pack2(r[0], r[1]);
                o2v[1] = Fmt16<short>::pack2(r[2], r[3]);
            }
            *(i32x2*)(mid + (size_t)v * 16 + q * 8) = o2v;
        }
    }
    __syncthreads();
    // ---- phase 2: second conv from LDS -> the pixel's channel run (f * D + d) in LDS
    const int nout = D * TY * TX;
    for (int g = wave; g * 16 < nout; g += 4) {
        const int v = g * 16 + l16;
        const bool vin = v < nout;
        const int vv = vin ? v : 0;
        const int tx = vv % TX, ty = (vv / TX) % TY, d = vv / (TX * TY);
        const char* mbase = mid + (size_t)((d * MY + ty) * MX + tx) * 16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            i32x4 frag = {0, 0, 0, 0};
            if (vin && (unsigned)(d + g2[s]) < (unsigned)D) frag = *(const i32x4*)(mbase + o2[s]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf2[s]), __builtin_bit_cast(bf16x8, frag), acc, 0, 0, 0);
        }
        if (vin && q < 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) outt[(ty * TX + tx) * (F * D) + (4 * q + e) * D + d] = f2bf(fmaxf(acc[e] * sc2[e] + sh2[e], 0.f));
        }
    }
    __syncthreads();
    // ---- phase 3: whole vectors out
    const int run = F * D;                              // channels per pixel
    if (vec_out) {
        const int vpp = run / 8;                        // 16-byte vectors per pixel
        for (int i = tid; i < TY * TX * vpp; i += 256) {
            const int pix = i / vpp, part = i - pix * vpp;
            const int y = y0 + pix / TX, x = x0 + pix % TX;
            if (y < H && x < W) *(i32x4*)(out + (((int64_t)b * H + y) * W + x) * ops + part * 8) = *(const i32x4*)(outt + pix * run + part * 8);
        }
    } else {
        for (int i = tid; i < TY * TX * run; i += 256) {
            const int pix = i / run, c = i - pix * run;
            const int y = y0 + pix / TX, x = x0 + pix % TX;
            if (y < H && x < W) out[(((int64_t)b * H + y) * W + x) * ops + c] = outt[i];
        }
    }
}

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

}  // namespace

extern "C" int vd3d_psm_cosine(const void* left, const void* right, void* cost, int B, int H, int W, int C, int D,
                               int ips, int ops, int dtype, void* stream) {
    const int ve = dtype == VD3D_BF16 ? 8 : 4;
    if (!left || !right || !cost || C % ve || ips % ve || D < 1 || D > 32 || ((uintptr_t)left & 15) || ((uintptr_t)right & 15)) {
        vd3d_set_error("psm_cosine: need C, stride multiples of 16 bytes, 1 <= D <= 32");
        return VD3D_EINVAL;
    }
    const int es = dtype == VD3D_BF16 ? 2 : 4;
    const int vec_store = ((ops * es) % 16 == 0) && (((uintptr_t)cost & 15) == 0);
    const int xtiles = (W + XT - 1) / XT;
    const int64_t grid = (int64_t)B * H * xtiles;
    const int lds = (XT + XT + D - 1) * 128;
    if (dtype == VD3D_BF16 && C % 32 == 0 && !vd3d_switch(VD3D_SW_PSM_VALU)) {
        const int groups_x = (W + 15) / 16;
        const int64_t ngroups = (int64_t)B * H * groups_x;
        const int cus = vd3d_device_cu_count();
        if (cus <= 0) return VD3D_ELAUNCH;
        const int64_t want = (ngroups + 3) / 4;
        const int g2 = (int)(want < (int64_t)cus * 8 ? want : (int64_t)cus * 8);
        hipLaunchKernelGGL(psm_cosine_mfma_kernel, dim3((unsigned)g2), dim3(256), 0, (hipStream_t)stream, (const short*)left, (const short*)right,
                           (short*)cost, W, C, D, ips, ops, groups_x, ngroups, vec_store && D % 8 == 0);
    } else if (dtype == VD3D_BF16)
        hipLaunchKernelGGL(psm_cosine_kernel<short>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream,
                           (const short*)left, (const short*)right, (short*)cost, H, W, C, D, ips, ops, xtiles, vec_store);
    else if (dtype == VD3D_F32)
        hipLaunchKernelGGL(psm_cosine_kernel<float>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream,
                           (const float*)left, (const float*)right, (float*)cost, H, W, C, D, ips, ops, xtiles, vec_store);
    else { vd3d_set_error("bad dtype"); return VD3D_EINVAL; }
    return vd3d_check_launch("psm_cosine");
}

extern "C" int vd3d_costvol_build(const void* left, const void* right, void* vol, int B, int H, int W, int F, int D,
                                  int ips, int dtype, void* stream) {
    if (dtype != VD3D_BF16 && dtype != VD3D_F32) { vd3d_set_error("costvol_build: dtype must be VD3D_BF16 or VD3D_F32 (no fp16 instantiation)"); return VD3D_EINVAL; }
    const int ve = dtype == VD3D_BF16 ? 8 : 4;
    if (!left || !right || !vol || F % ve || ips % ve) { vd3d_set_error("costvol_build: F and stride must be 16-byte multiples"); return VD3D_EINVAL; }
    const int64_t total = (int64_t)B * D * H * W * (F / ve);
    if (dtype == VD3D_BF16)
        hipLaunchKernelGGL(costvol_build_kernel<short>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const short*)left, (const short*)right, (short*)vol, B, H, W, F, D, ips);
    else
        hipLaunchKernelGGL(costvol_build_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)left, (const float*)right, (float*)vol, B, H, W, F, D, ips);
    return vd3d_check_launch("costvol_build");
}

extern "C" int vd3d_conv3d_3x3x3(const void* in, const float* weight, const float* scale, const float* shift, void* out,
                                 int B, int D, int H, int W, int Cin, int Cout, int relu, int out_fd_major,
                                 int ops, int dtype, void* stream) {
    if (!in || !weight || !scale || !shift || !out) { vd3d_set_error("conv3d: null pointer"); return VD3D_EINVAL; }
    if (dtype != VD3D_BF16 && dtype != VD3D_F32) { vd3d_set_error("conv3d: dtype must be VD3D_BF16 or VD3D_F32 (no fp16 instantiation)"); return VD3D_EINVAL; }
    const int64_t total = (int64_t)B * D * H * W;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
#define VD3D_C3D(T, CI, CO)                                                                                         \
    hipLaunchKernelGGL((conv3d_kernel<T, CI, CO>), dim3(grid), dim3(256), 0, s, (const T*)in, weight, scale, shift, \
                       (T*)out, B, D, H, W, relu, out_fd_major, ops)
    const unsigned mgrid = (unsigned)(((total + 15) / 16 + 3) / 4 < 2048 ? ((total + 15) / 16 + 3) / 4 : 2048);   // 4 waves / block
    if (dtype == VD3D_BF16 && Cout == 8 && (Cin == 16 || Cin == 8) && ((uintptr_t)in & 15) == 0 && !vd3d_switch(VD3D_SW_CONV3D_VALU)) {
        if (Cin == 16) hipLaunchKernelGGL(conv3d_mfma_kernel<16>, dim3(mgrid), dim3(256), 0, s, (const short*)in, weight, scale, shift, (short*)out, B, D, H, W, relu, out_fd_major, ops);
        else hipLaunchKernelGGL(conv3d_mfma_kernel<8>, dim3(mgrid), dim3(256), 0, s, (const short*)in, weight, scale, shift, (short*)out, B, D, H, W, relu, out_fd_major, ops);
    }
    else if (Cin == 16 && Cout == 8) { if (dtype == VD3D_BF16) VD3D_C3D(short, 16, 8); else VD3D_C3D(float, 16, 8); }
    else if (Cin == 8 && Cout == 8) { if (dtype == VD3D_BF16) VD3D_C3D(short, 8, 8); else VD3D_C3D(float, 8, 8); }
    else { vd3d_set_error("conv3d: only (Cin,Cout) in {(16,8),(8,8)} are instantiated (CostVolume PSM_features=8)"); return VD3D_EINVAL; }
#undef VD3D_C3D
    return vd3d_check_launch("conv3d_3x3x3");
}

extern "C" int vd3d_cost_volume_fused(const void* left, const void* right, const float* w1, const float* scale1, const float* shift1,
                                      const float* w2, const float* scale2, const float* shift2, void* out, int B, int H, int W, int F,
                                      int D, int ips, int ops, int dtype, void* stream) {
    if (!left || !right || !w1 || !scale1 || !shift1 || !w2 || !scale2 || !shift2 || !out) { vd3d_set_error("cost_volume_fused: null pointer"); return VD3D_EINVAL; }
    if (dtype != VD3D_BF16 || F != 8 || D < 1 || D > 24 || ips % 8 || ((uintptr_t)left & 15) || ((uintptr_t)right & 15)) {
        vd3d_set_error("cost_volume_fused: bf16, PSM_features = 8, D <= 24, 16-byte aligned feature pixels only (other cases: the three-launch path)");
        return VD3D_EINVAL;
    }
    if (B == 0) return VD3D_OK;
    constexpr int TY = 2, TX = 20;                 // 384 workgroups at 8 x 24 x 80 (2 x 40 tiles: 192 workgroups of long serial loops, 92 us)
    const int xtiles = (W + TX - 1) / TX, ytiles = (H + TY - 1) / TY;
    int lds = D * (TY + 2) * (TX + 2) * 16 + TY * TX * F * D * 2;
    if (lds < (27 * 16 * 8 + 27 * 8 * 8) * 4) lds = (27 * 16 * 8 + 27 * 8 * 8) * 4;      // the prologue's weight staging
    const int vec_out = ((F * D) % 8 == 0) && (ops % 8 == 0) && (((uintptr_t)out & 15) == 0);
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)cost_volume_fused_kernel<TY, TX>, lds, lim, "hipFuncSetAttribute(cost_volume_fused)")) return rc;
    hipLaunchKernelGGL((cost_volume_fused_kernel<TY, TX>), dim3((unsigned)(B * ytiles * xtiles)), dim3(256), lds, (hipStream_t)stream,
                       (const short*)left, (const short*)right, w1, scale1, shift1, w2, scale2, shift2, (short*)out, H, W, D, ips, ops,
                       xtiles, ytiles, vec_out);
    return vd3d_check_launch("cost_volume_fused");
}
