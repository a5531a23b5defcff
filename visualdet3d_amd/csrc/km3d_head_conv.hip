// km3d_head_conv.hip -- the fused KM3D / RTM3D head as one persistent kernel (gfx950).
//
// Replaces KM3DHead.forward's nine branches (heads/km3d_head.py:132-153, :353-357: per branch Conv2d(C, 256, 3, padding 1) + ReLU +
// Conv2d(256, n_h, 1)) -- 18 convolutions and a 9 x 256-channel intermediate (4 GB at 16 x 128 x 440) in the reference.  Here a tile
// is 256 pixels x one branch: the 3x3 conv is the implicit GEMM of conv_igemm.hip (operands by LDS-DMA, 32x32x16 MFMAs, 4 x 2 waves of
// 64 pixels x 128 hidden channels), and the 1x1 conv is a second GEMM taken STRAIGHT FROM THE ACCUMULATORS: after ReLU + rounding to
// the 16-bit format (the rounding point of the unfused path's `mid` tensor) a lane's accumulator quads g, g + 1 of channel block i are
// the B fragment of a 32x32x16 MFMA whose k run is the 16-channel group 2 i + g / 2 in the order [0-3, 8-11 | 4-7, 12-15] -- the order
// the W2 image is stored in.  The two channel halves of a pixel (wn = 0 | 1) meet in LDS, 4 KB per wave.
//
// What a tile costs beside its 9 K slices (cycle stamps, first persistent version inside conv_igemm_dma_kernel: tile 35.8k cycles =
// 24.8k main loop + 3.9k prologue + 7.1k epilogue):
//   * prologue = issuing the first 1.5 slices (96 KB through the CU's 64 B / clk texture path) and waiting for them.  Here the NEXT
//     tile's first slice and head constants are requested right after the epilogue's first barrier, under the reduction and the stores;
//   * the first-conv bias is the accumulators' initial value (the epilogue was VALU bound: 128 add + 128 max + 64 cvt per lane);
//   * both waves of a pixel group finish 32 pixels each (the first version left half the waves idle during reduction and stores).
#include "conv_common.h"

namespace vd3d_conv {
namespace {

constexpr int kBM = 256, kBN = 256, kNW = 8;
constexpr int kAStage = kBM * 128, kStage = (kBM + kBN) * 128;          // 32 KiB of pixels + 32 KiB of weights per 64-deep K slice
constexpr int kW2 = 2 * kStage, kB1 = kW2 + 16384, kB2 = kB1 + 1024;    // W2 image [32][512 B] | bias1 [256] | bias2 [2 parities][32]
constexpr int kMeet = kStage, kOut = kStage + 32768;                    // (inside stage 1, free between two tiles)
constexpr int kHeadLds = kB2 + 256;

// STG > 0 (round 4): the two waves of a SIMD (w and w + 4) issue their DMA pieces at different points of a slice -- the lower four as before
// (half right behind the slice's barrier, half at the top of the next slice), the upper four all eight pieces of slice t + 1 at fragment STG of
// slice t: behind the barrier SIMD mates run in lock step, so both sat in their DMA bursts (60 - 185 issue cycles per piece) at the same time
// with the MFMA pipe idle (conv_igemm.hip, same idea on the 352 strips: +3.6 %).  Results do not depend on STG (same k order).
template <typename T, int STG>
__global__ void __launch_bounds__(kNW * 64) km3d_head_kernel(const ConvArgs p) {
    constexpr int TM = 2, TN = 4, NSUB = 4, WTM = 64, WTN = 128, A_PIECES = 4, NPIECE = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, half = lane >> 5;
    // XCD-aware walk: round r covers tiles [r * nwg, (r + 1) * nwg); inside a round an XCD's workgroups own a contiguous run
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int pos = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int total = p.tiles_m * p.tiles_n;
    if (pos >= total) return;

    // ---- per-tile state ------------------------------------------------------------------------------------------------------
    int tile_n = 0, m0 = 0;
    int a_off[A_PIECES], a_iy[A_PIECES], a_ix[A_PIECES];
    int kc = 0, tap = 0, dy = 0, dx = 0;
    uint32_t w_row = 0, w_off = 0;
    i32x4 hd_w2[2];
    f32x4 hd_b = {0.f, 0.f, 0.f, 0.f};
    const int prow = lane >> 3;
    const int slot = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);  // = (L % 8) ^ ((row / 2) % 8), constant over pieces
    const int HoWo = p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.weight, 0, p.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    auto setup = [&](int tile) {
        tile_n = fastdiv(tile, p.fd_tiles_m);
        m0 = (tile - tile_n * p.tiles_m) * kBM;
        // head constants of branch tile_n: requested first (the LDS write below waits for them, not for the DMAs issued after)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int v = tid + u * kNW * 64, r = v >> 5, sl = v & 31;
            hd_w2[u] = *(const i32x4*)(p.h_w2 + ((size_t)(tile_n * 32 + r) * 256 + sl * 8) * 2);
        }
        if (tid < 64) hd_b = *(const f32x4*)(p.shift + tile_n * kBN + tid * 4);
        else if (tid < 72) hd_b = *(const f32x4*)(p.h_b2 + tile_n * 32 + (tid - 64) * 4);
#pragma unroll
        for (int it = 0; it < A_PIECES; ++it) {
            const int m = m0 + 8 * (wave + it * kNW) + prow;
            if (m < p.M) {
                const int b = fastdiv(m, p.fd_howo), rem = m - b * HoWo;
                const int oy = fastdiv(rem, p.fd_wo), ox = rem - oy * p.Wo;
                a_iy[it] = oy - 1;
                a_ix[it] = ox - 1;
                a_off[it] = (int)(b * p.in_batch_stride) + a_iy[it] * p.in_row_stride + a_ix[it] * p.in_pix_stride;
            } else {
                a_iy[it] = -(1 << 28);
                a_ix[it] = 0;
                a_off[it] = 0;
            }
        }
        kc = slot * 8, tap = 0, dy = 0, dx = 0;                   // chunk-major / tap-minor K walk (Cin % 64 == 0)
        w_row = (uint32_t)(((tile_n * kBN + 8 * wave + prow) * p.Kpad + slot * 8) * 2);
        w_off = w_row;
    };
    auto write_hd = [&](int parity) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int v = tid + u * kNW * 64, r = v >> 5, sl = v & 31;
            // source slot (group, hh) = channels 8 hh + 0..7 of a 16-channel group -> low half to slot (group, 0), high half to slot
            // (group, 1), each at byte 8 hh: a slot then holds [0-3, 8-11] | [4-7, 12-15]
            const int ksg = sl >> 1, hh = sl & 1;
            *(i32x2*)(smem + kW2 + r * 512 + (((2 * ksg) ^ (r & 15)) << 4) + hh * 8) = i32x2{hd_w2[u][0], hd_w2[u][1]};
            *(i32x2*)(smem + kW2 + r * 512 + (((2 * ksg + 1) ^ (r & 15)) << 4) + hh * 8) = i32x2{hd_w2[u][2], hd_w2[u][3]};
        }
        if (tid < 64) *(f32x4*)(smem + kB1 + tid * 16) = hd_b;
        else if (tid < 72) *(f32x4*)(smem + kB2 + parity * 128 + (tid - 64) * 16) = hd_b;
    };
    // one K slice = 4 pixel pieces + 4 weight pieces of 1 KiB per wave, issued in 4 groups of 2
    auto issue_group = [&](int st, int g, bool enable) {
        char* base = smem + st * kStage + wave * 1024;
        const bool kvalid = enable && tap < 9;
        const int tap_off = dy * p.in_row_stride + dx * p.in_pix_stride + kc;
#pragma unroll
        for (int pi = 0; pi < NPIECE; ++pi) {
            if ((pi & 3) != g) continue;
            if (pi < A_PIECES) {
                const bool v = kvalid && (unsigned)(a_iy[pi] + dy) < (unsigned)p.H && (unsigned)(a_ix[pi] + dx) < (unsigned)p.W;
                const uint32_t off = v ? (uint32_t)(a_off[pi] + tap_off) * 2u : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(base + pi * kNW * 1024), 16, off, 0, 0, 0);
            } else {
                const int it = pi - A_PIECES;
                const uint32_t off = enable ? w_off + (uint32_t)(it * kNW * 8 * p.Kpad * 2) : kOOB;   // (branch-free: a disabled piece writes zeros)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(base + kAStage + it * kNW * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    auto advance_k = [&]() {
        ++tap;
        if (++dx == 3) { dx = 0; ++dy; }
        if (tap == 9) { tap = 0; dx = 0; dy = 0; kc += 64; }
        w_off = w_row + (uint32_t)((tap * p.Cin + kc - slot * 8) * 2);
    };
    auto ld_w = [&](int st, int ks, int i) {
        const int row = wn * WTN + i * 32 + lr;
        return *(const i32x4*)(smem + st * kStage + kAStage + row * 128 + (((2 * ks + half) ^ ((row >> 1) & 7)) << 4));
    };
    auto ld_a = [&](int st, int ks, int j) {
        const int row = wm * WTM + j * 32 + lr;
        return *(const i32x4*)(smem + st * kStage + row * 128 + (((2 * ks + half) ^ ((row >> 1) & 7)) << 4));
    };

    // ---- first tile of this workgroup ----------------------------------------------------------------------------------------
    int tile = pos, parity = 0;
    setup(tile);
#pragma unroll
    for (int g = 0; g < 4; ++g) issue_group(0, g, true);
    advance_k();
    write_hd(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool hiw = STG > 0 && wave >= kNW / 2;
    for (;;) {
        // slice 0 has landed for everybody and the head constants are in LDS; half of slice 1 now, the rest inside slice 0
        if (!hiw) {
            issue_group(1, 0, p.nk > 1);
            issue_group(1, 1, p.nk > 1);
        }
        f32x16 acc[TN][TM];
        {
            // one LDS read per accumulator quad, as volatile asm (the two pixel blocks read the same four biases separately, which the
            // compiler would merge into one read + 64 copies): the reads land in the accumulator registers
            const uint32_t bb = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + kB1) + (uint32_t)((wn * WTN + 4 * half) * 4);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < TM; ++j) {
                        f32x4 b1;
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b1) : "v"(bb), "n"((i * 32 + 8 * g) * 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b1[e];
                    }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // ---- main loop: the software-pipelined slice walk of conv_igemm_dma_kernel<.., PIPE, 32> (ring of TN weight fragments,
        // double-buffered pixel fragments, barrier before the last sub-step of a slice) --------------------------------------------
        {
            constexpr int R = TN, F0 = NSUB * TN - R;
            i32x4 fa[R], fb[2][TM];
#pragma unroll
            for (int i = 0; i < R; ++i) fa[i] = ld_w(0, 0, i);
#pragma unroll
            for (int j = 0; j < TM; ++j) fb[0][j] = ld_a(0, 0, j);
            for (int kt = 0; kt < p.nk; ++kt) {
                const int st = kt & 1;
                const bool more1 = kt + 1 < p.nk, more2 = kt + 2 < p.nk;
#pragma unroll
                for (int f = 0; f < NSUB * TN; ++f) {
                    const int ks = f / TN, i = f - ks * TN;
                    if (f == F0) {
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        if (!hiw) {
                            issue_group(st, 0, more2);
                            issue_group(st, 1, more2);
                        }
                    }
                    if ((i == 0 && ks < NSUB - 1) || f == F0) {
                        const int k2 = f == F0 ? NSUB - 1 : ks;
#pragma unroll
                        for (int j = 0; j < TM; ++j) fb[(k2 & 1) ^ 1][j] = ld_a(k2 < NSUB - 1 ? st : st ^ 1, (k2 + 1) % NSUB, j);
                        __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
                    }
                    if (i == 0 && ks == 0) {
                        if (!hiw) {
                            issue_group(st ^ 1, 2, more1);
                            issue_group(st ^ 1, 3, more1);
                            advance_k();
                        }
                    }
                    if constexpr (STG > 0) {
                        // (no pieces at all past the last slice: stage 1 is where the partial sums of the epilogue meet)
                        if (f == STG && hiw && more1) {
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) issue_group(st ^ 1, g4, true);
                            advance_k();
                        }
                    }
#pragma unroll
                    for (int j = 0; j < TM; ++j) Fmt16<T>::mfma32(fa[f % R], fb[ks & 1][j], acc[i][j]);
                    const int nf = f + R, nks = nf / TN, ni = nf - nks * TN;
                    fa[f % R] = ld_w(nks < NSUB ? st : st ^ 1, nks % NSUB, ni);
                    __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    // the sub-step's DMA pieces (4 in the first and in the last sub-step of a slice), two per fragment
                    if ((ks == 0 && i < 2) || (ks == NSUB - 1 && i < 2)) {
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            }
        }
        // ---- second GEMM from the registers: out_h[32][64 px] += W2_h[32][this wave's 128 channels] x relu(hidden) ----------------
        // (lane-derived epilogue addresses are re-derived per tile: hoisted out of the tile loop they would sit in registers through
        // the main loop, which has none to spare)
        int eln = lane;
        asm volatile("" : "+v"(eln));
        const int elr = eln & 31, ehalf = eln >> 5;
        f32x16 acc2[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const int ks = wn * 8 + i * 2 + gp;                       // 16-channel group inside the branch
                const i32x4 fa2 = *(const i32x4*)(smem + kW2 + elr * 512 + (((2 * ks + ehalf) ^ (elr & 15)) << 4));
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const f32x16& a = acc[i][j];             // round, then ReLU on the packed pairs (one v_pk_max_i16 per two values)
                    const i32x4 fb2 = {relu_pk16(Fmt16<T>::pack2_1(a[8 * gp], a[8 * gp + 1])), relu_pk16(Fmt16<T>::pack2_1(a[8 * gp + 2], a[8 * gp + 3])),
                                       relu_pk16(Fmt16<T>::pack2_1(a[8 * gp + 4], a[8 * gp + 5])), relu_pk16(Fmt16<T>::pack2_1(a[8 * gp + 6], a[8 * gp + 7]))};
                    Fmt16<T>::mfma32(fa2, fb2, acc2[j]);
                }
            }
        // No barrier here when nk is odd: past the last slice's barrier nobody reads live data from the operand stages any more (the
        // ring refills of the last sub-step read a dead stage and are never consumed) and the disabled (zero-writing) DMA pieces issued
        // at that barrier target stage 0, each wave its own pieces -- the same ones its next-tile DMA rewrites, in order.  With an even
        // slice count they target stage 1, where the partner's partial sums are about to be written: wait for them, for everybody.
        if (!(p.nk & 1)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // ---- the next tile's first slice and head constants go out now, under the reduction and the stores ------------------------
        const int cur_h = tile_n, cur_m0 = m0;
        const int next = tile + nwg;
        const bool has_next = next < total;
        if (has_next) {
            setup(next);
            issue_group(0, 0, true);
            issue_group(0, 1, true);
        }
        // pixel block j = wn is finished by this wave, block 1 - wn by its partner (wm, 1 - wn): hand the partial sums over
        f32x16 keep, give;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            keep[e] = wn ? acc2[1][e] : acc2[0][e];
            give[e] = wn ? acc2[0][e] : acc2[1][e];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(f32x4*)(smem + kMeet + ((wm * 2 + (1 - wn)) * 4 + g) * 1024 + eln * 16) = f32x4{give[4 * g], give[4 * g + 1], give[4 * g + 2], give[4 * g + 3]};
        __syncthreads();                // partial sums are there; every wave is done with the W2 image and bias1
        if (has_next) {
            issue_group(0, 2, true);
            issue_group(0, 3, true);
            advance_k();
            write_hd(parity ^ 1);
        }
        {
            // The wave's 32 pixels x nh outputs are one contiguous run of the [M][nh] map: laid out as [pixel][nh] in LDS first, so that
            // consecutive lanes store consecutive floats (whole lines) instead of nh scattered floats per lane.
            const int nh = p.h_n[cur_h];
            float* stage = (float*)(smem + kOut + wave * 4096);
            const float* b2 = (const float*)(smem + kB2 + parity * 128);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 other = *(const f32x4*)(smem + kMeet + ((wm * 2 + wn) * 4 + g) * 1024 + eln * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = 8 * g + 4 * ehalf + e;
                    if (n < nh) stage[elr * nh + n] = keep[4 * g + e] + other[e] + b2[n];
                }
            }
            const int mw = cur_m0 + wm * WTM + wn * 32;          // first pixel of this wave's run
            const int nrun = (p.M - mw < 32 ? (p.M - mw > 0 ? p.M - mw : 0) : 32) * nh;
            float* dst = p.h_out[cur_h] + (int64_t)mw * nh;
            for (int i = eln; i < nrun; i += 64) dst[i] = stage[i];
        }
        if (!has_next) break;
        tile = next;
        parity ^= 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next tile's slice 0 (and, in-order, this tile's stores)
        __syncthreads();                                        // ... for everybody; meeting area and run buffers are free again
    }
}

}  // namespace

bool km3d_head_shape_ok(const ConvArgs& a) {
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.Cin % 64 == 0 && a.Cout % kBN == 0 && a.chunk_major &&
           a.Ho == a.H && a.Wo == a.W;
}

int launch_km3d_head(ConvArgs& a, hipStream_t stream, int fmt) {
    a.tiles_m = (a.M + kBM - 1) / kBM;
    a.tiles_n = a.Cout / kBN;
    a.fd_tiles_m = make_fastdiv((uint32_t)a.tiles_m);
    const int64_t total = (int64_t)a.tiles_m * a.tiles_n;
    if (total <= 0 || total > 0x7fffffff) return VD3D_EINVAL;
    const int cus = vd3d_device_cu_count();
    if (cus <= 0) return VD3D_ELAUNCH;
    const int grid = total < cus ? (int)total : cus;
    // VD3D_HEAD_NO_STAGGER=1: every wave on the same DMA schedule (A/B; bit-identical results)
    const bool stg = !vd3d_switch(VD3D_SW_HEAD_NO_STAGGER);
    auto go = [&](auto kern, Vd3dLdsLimit& lim) -> int {
        if (const int rc = vd3d_raise_lds_limit((const void*)kern, kHeadLds, lim, "hipFuncSetAttribute(km3d_head)")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kNW * 64), kHeadLds, stream, a);
        return 0;
    };
    static Vd3dLdsLimit lim[4];
    constexpr int kStg = 5;
    int rc;
    if (fmt == VD3D_F16) rc = stg ? go(km3d_head_kernel<hf16, kStg>, lim[0]) : go(km3d_head_kernel<hf16, 0>, lim[1]);
    else rc = stg ? go(km3d_head_kernel<short, kStg>, lim[2]) : go(km3d_head_kernel<short, 0>, lim[3]);
    if (rc) return rc;
    return vd3d_check_launch("km3d_head");
}

}  // namespace vd3d_conv
