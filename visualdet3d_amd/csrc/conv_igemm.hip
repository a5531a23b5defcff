// conv_igemm.hip -- fused implicit-GEMM convolution for gfx950 (MI355X), NHWC activations.
//
//   out[m][n] = act( (sum_k A[m][k] * Wt[n][k]) * scale[n] + shift[n] (+ residual[m][n]) )
//   m = output pixel (b, oy, ox), n = output channel, k = (ky*kw + kx)*Cin + c
//
// Replaces the reference's nn.Conv2d + BatchNorm2d + ReLU (+ residual) chains (cuDNN + separate elementwise
// kernels): backbones/resnet.py:23-52,55-91; lib/blocks.py:24-43; lib/ghost_module.py:27-32;
// heads/detection_3d_head.py:54-79,508-530.
//
// CDNA4 mapping
//   * one workgroup = WARPS_M x WARPS_N waves (64 lanes each) computing a BM(pixels) x BN(channels) tile;
//   * K is walked in 128-byte slices (64 bf16 / 32 fp32 per row); both operand tiles are staged in LDS,
//     double buffered, XOR-swizzled on the 16-byte slot so that ds_read_b128 fragment reads and
//     ds_write_b128 staging writes are bank-conflict free;
//   * global loads are raw buffer loads (SRD bounds check supplies the zero padding: an out-of-image tap
//     gets an out-of-range offset and the hardware returns 0 -- no divergent branches in the loader);
//   * MFMA 32x32: the WEIGHT tile is the A operand (rows = channels), the PIXEL tile is the B operand
//     (cols = pixels), so each lane ends up holding 4 consecutive output channels of ONE pixel per
//     accumulator quad -> the NHWC epilogue stores 8/16 contiguous bytes per lane;
//   * bf16: v_mfma_f32_32x32x16_bf16; fp32 validation mode: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain);
//   * the 1-D grid is remapped so each XCD (private 4 MiB L2) walks a contiguous run of pixel tiles that
//     share one weight panel.
#include "conv_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

using namespace vd3d_conv;

namespace {

template <typename T> struct Mma {          // 16-bit formats: bf16 (T = short) | fp16 (T = hf16)
    static VD3D_DEV void run(const i32x4& a, const i32x4& b, f32x16& acc) { Fmt16<T>::mfma32(a, b, acc); }
};
template <> struct Mma<float> {
    // lane half h holds 4 consecutive k of an 8-wide k group; MFMA j pairs k = j (h=0) with k = 4 + j (h=1):
    // any bijection k -> (mfma, half) is valid as long as A and B use the same one.
    static VD3D_DEV void run(const i32x4& a, const i32x4& b, f32x16& acc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int aj = a[j], bj = b[j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(i2f(aj), i2f(bj), acc, 0, 0, 0);
        }
    }
};

// ---- epilogue: scale/shift (+residual) (+ReLU), NHWC store ------------------------------------------------
// Accumulator layout (32x32 MFMA, weights = A operand): lane (lr, half) holds, for accumulator quad g, the 4 consecutive
// output channels 8g + 4*half .. +3 of pixel lr.  bf16 output: the two half-waves of a pixel hold adjacent 8-byte runs;
// one v_permlane32_swap per dword pairs quads (g, g+1) so that every lane stores 16 contiguous bytes (half the store
// instructions -- the store tail of short-K layers is issue-bound, cf. guide T21).
template <typename T, int TM, int TN, int WTM, int WTN>
VD3D_DEV void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[TN][TM], const int (&mrow)[TM], int n0, int wn, int half, const float* ltab = nullptr, int ltn = 0) {
    // ltab: optional LDS copy of the tile's constants, scale[ltn] | shift[ltn] for channels n0 .. n0 + ltn - 1 (1 / 0 where absent)
    const bool f32_out = sizeof(T) == 4 || p.out_f32;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = mrow[j];      // flat output pixel index of this lane for accumulator column j, or -1
        const bool mvalid = m >= 0;
        const int64_t obase = (int64_t)(mvalid ? m : 0) * p.out_pix_stride;
        const int64_t rbase = (int64_t)(mvalid ? m : 0) * p.res_pix_stride;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int nt = n0 + wn * WTN + i * 32;     // wave-uniform
            if (nt >= p.Cout) continue;
            if (p.vec_epilogue) {
                i32x2 packed[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = nt + 8 * g + 4 * half;
                    const bool ok = mvalid && nb < p.Cout;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                    if (ok) {
                        if (ltab) {
                            const f32x4 s = *(const f32x4*)(ltab + (nb - n0)), t = *(const f32x4*)(ltab + ltn + (nb - n0));
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] * s[e] + t[e];
                        } else {
                        if (p.scale) {
                            const f32x4 s = *(const f32x4*)(p.scale + nb);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= s[e];
                        }
                        if (p.shift) {
                            const f32x4 s = *(const f32x4*)(p.shift + nb);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += s[e];
                        }
                        }
                        if (p.residual) {
                            if constexpr (sizeof(T) == 2) {
                                const i32x2 rr = *(const i32x2*)(p.residual + (rbase + nb) * 2);
                                const uint32_t r0 = (uint32_t)(int)rr[0], r1 = (uint32_t)(int)rr[1];
                                v[0] += Fmt16<T>::lo(r0);
                                v[1] += Fmt16<T>::hi(r0);
                                v[2] += Fmt16<T>::lo(r1);
                                v[3] += Fmt16<T>::hi(r1);
                            } else {
                                const f32x4 rr = *(const f32x4*)(p.residual + (rbase + nb) * 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += rr[e];
                            }
                        }
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        if (f32_out) {
                            f32x4 o = {v[0], v[1], v[2], v[3]};
                            *(f32x4*)(p.out + (obase + nb) * 4) = o;
                        }
                    }
                    packed[g][0] = Fmt16<T>::pack2(v[0], v[1]);
                    packed[g][1] = Fmt16<T>::pack2(v[2], v[3]);
                }
                if (!f32_out) {
                    if (p.wide_store) {
#pragma unroll
                        for (int g = 0; g < 4; g += 2) {
                            // x = quad g, y = quad g+1: after the swap the lower half-wave holds [own g | upper's g] and the
                            // upper half-wave [lower's g+1 | own g+1]: 16 contiguous bytes each
                            int x0 = packed[g][0], x1 = packed[g][1], y0 = packed[g + 1][0], y1 = packed[g + 1][1];
                            auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                            auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                            const int nb16 = nt + 8 * (g + half);
                            if (mvalid && nb16 < p.Cout) {
                                i32x4 o = {(int)r0[0], (int)r1[0], (int)r0[1], (int)r1[1]};
                                *(i32x4*)(p.out + (obase + nb16) * 2) = o;
                            }
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int nb = nt + 8 * g + 4 * half;
                            if (mvalid && nb < p.Cout) *(i32x2*)(p.out + (obase + nb) * 2) = packed[g];
                        }
                    }
                }
            } else {
                if (!mvalid) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = nt + 8 * g + 4 * half + e;
                        if (n >= p.Cout) continue;
                        float x = acc[i][j][4 * g + e];
                        if (p.scale) x *= p.scale[n];
                        if (p.shift) x += p.shift[n];
                        if (p.residual) x += ElemTraits<T>::to_f(((const T*)p.residual)[rbase + n]);
                        if (p.relu) x = fmaxf(x, 0.f);
                        if (f32_out) ((float*)p.out)[obase + n] = x;
                        else ((T*)p.out)[obase + n] = Fmt16<T>::one(x);
                    }
            }
        }
    }
}

// MFMA shape selector for the pipelined loop: 32 -> the Mma<T> above, 16 -> v_mfma_f32_16x16x32_bf16
template <typename T, int MS> struct MmaShape {
    typedef f32x16 acc_t;
    static VD3D_DEV void run(const i32x4& a, const i32x4& b, f32x16& acc) { Mma<T>::run(a, b, acc); }
};
template <typename T> struct MmaShape<T, 16> {
    typedef f32x4 acc_t;
    static VD3D_DEV void run(const i32x4& a, const i32x4& b, f32x4& acc) { Fmt16<T>::mfma16(a, b, acc); }
};

// Epilogue for the 16x16x32 layout (weights = A operand): lane (l16, q) holds output channels 4q .. 4q+3 of pixel l16 of
// the 16x16 block; bf16 only, 8-byte NHWC stores (vector path) or scalar stores.
template <typename T, int TM, int TN, int WTN>
VD3D_DEV void conv_epilogue16(const ConvArgs& p, f32x4 (&acc)[TN][TM], const int (&mrow)[TM], int n0, int wn, int q, const float* ltab = nullptr, int ltn = 0) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = mrow[j];
        if (m < 0) continue;
        const int64_t obase = (int64_t)m * p.out_pix_stride;
        const int64_t rbase = (int64_t)m * p.res_pix_stride;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int nb = n0 + wn * WTN + i * 16 + 4 * q;
            if (nb >= p.Cout) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
            if (p.vec_epilogue) {
                if (ltab) {
                    const f32x4 s = *(const f32x4*)(ltab + (nb - n0)), t = *(const f32x4*)(ltab + ltn + (nb - n0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * s[e] + t[e];
                } else {
                if (p.scale) {
                    const f32x4 s = *(const f32x4*)(p.scale + nb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= s[e];
                }
                if (p.shift) {
                    const f32x4 s = *(const f32x4*)(p.shift + nb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += s[e];
                }
                }
                if (p.residual) {
                    const i32x2 rr = *(const i32x2*)(p.residual + (rbase + nb) * 2);
                    const uint32_t r0 = (uint32_t)(int)rr[0], r1 = (uint32_t)(int)rr[1];
                    v[0] += Fmt16<T>::lo(r0);
                    v[1] += Fmt16<T>::hi(r0);
                    v[2] += Fmt16<T>::lo(r1);
                    v[3] += Fmt16<T>::hi(r1);
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (p.out_f32) {
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    *(f32x4*)(p.out + (obase + nb) * 4) = o;
                } else {
                    i32x2 o;
                    o[0] = Fmt16<T>::pack2(v[0], v[1]);
                    o[1] = Fmt16<T>::pack2(v[2], v[3]);
                    *(i32x2*)(p.out + (obase + nb) * 2) = o;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = nb + e;
                    if (n >= p.Cout) continue;
                    float x = v[e];
                    if (p.scale) x *= p.scale[n];
                    if (p.shift) x += p.shift[n];
                    if (p.residual) x += Fmt16<T>::tof(((const T*)p.residual)[rbase + n]);
                    if (p.relu) x = fmaxf(x, 0.f);
                    if (p.out_f32) ((float*)p.out)[obase + n] = x;
                    else ((T*)p.out)[obase + n] = Fmt16<T>::one(x);
                }
            }
        }
    }
}

// Whole-line variant of the 16x16x32 epilogue (16-bit output, every channel of the wave's strip inside Cout): the strips finish
// their ONE round of tiles together, so the output leaves as a synchronised burst -- in the accumulator layout as 8-byte pieces
// (32 bytes per pixel and store instruction).  Here a wave parks each 16-pixel block of its tile in LDS ([pixel][WTN channels +
// pad]; the operand stages are free by now) and re-reads it row-major: a store instruction then covers whole pixel runs of
// WTN * 2 contiguous bytes.  Residual reads stay in the accumulator layout.
template <typename T, int TM, int TN, int WTN>
VD3D_DEV void conv_epilogue16_lines(const ConvArgs& p, f32x4 (&acc)[TN][TM], const int (&mrow)[TM], int mblock0, int n0, int nw, int q, int l16,
                                    int lane, char* tile, const float* ltab, int ltn) {
    constexpr int ROWB = WTN * 2 + 16;                 // bytes per parked pixel row (16-byte aligned; + 16 spreads the banks)
    constexpr int CPR = WTN / 8;                       // 16-byte chunks per pixel row
    const int relu_floor = p.relu ? 0 : (int)0x80008000u;      // packed-pair floor: ReLU | identity
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = mrow[j];
        const bool mvalid = m >= 0;
        const int64_t rbase = (int64_t)(mvalid ? m : 0) * p.res_pix_stride;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int nb = nw + i * 16 + 4 * q;
            float v[4];
            const f32x4 s = *(const f32x4*)(ltab + (nb - n0)), t = *(const f32x4*)(ltab + ltn + (nb - n0));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * s[e] + t[e];
            if (p.residual && mvalid) {
                const i32x2 rr = *(const i32x2*)(p.residual + (rbase + nb) * 2);
                const uint32_t r0 = (uint32_t)(int)rr[0], r1 = (uint32_t)(int)rr[1];
                v[0] += Fmt16<T>::lo(r0);
                v[1] += Fmt16<T>::hi(r0);
                v[2] += Fmt16<T>::lo(r1);
                v[3] += Fmt16<T>::hi(r1);
            }
            // round, then ReLU on the packed pairs (bit-identical to fmaxf before the rounding: 4 instead of ~12 instructions per four values)
            *(i32x2*)(tile + l16 * ROWB + (i * 16 + 4 * q) * 2) = i32x2{max_pk16(Fmt16<T>::pack2_1(v[0], v[1]), relu_floor), max_pk16(Fmt16<T>::pack2_1(v[2], v[3]), relu_floor)};
        }
#pragma unroll
        for (int it = 0; it < (16 * CPR + 63) / 64; ++it) {
            const int c = it * 64 + lane;
            if (c < 16 * CPR) {
                const int r = c / CPR, ch = c - r * CPR;
                const i32x4 o = *(const i32x4*)(tile + r * ROWB + ch * 16);
                const int mm = mblock0 + j * 16 + r;
                if (mm < p.M) *(i32x4*)(p.out + ((int64_t)mm * p.out_pix_stride + nw) * 2 + ch * 16) = o;
            }
        }
    }
}

template <typename T, int BM, int BN, int WARPS_M, int WARPS_N>
__global__ void __launch_bounds__(WARPS_M* WARPS_N * 64) conv_igemm_kernel(const ConvArgs p) {
    constexpr int NT = WARPS_M * WARPS_N * 64;
    constexpr int ES = (int)sizeof(T);
    constexpr int VE = 16 / ES;    // elements per 16-byte vector
    constexpr int BKE = 128 / ES;  // elements per K slice
    constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int ROWS_PER_IT = NT / 8;
    constexpr int A_IT = BM / ROWS_PER_IT, W_IT = BN / ROWS_PER_IT;
    constexpr int A_STAGE = BM * 128, W_STAGE = BN * 128, STAGE = A_STAGE + W_STAGE;
    static_assert(A_IT >= 1 && W_IT >= 1, "tile too small for the loader");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- tile id: XCD-aware bijective remap (block b runs on XCD b % 8) -----------------------------
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = tile / p.tiles_m, tile_m = tile - tile_n * p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;

    // ---- loader state -------------------------------------------------------------------------------
    const int slot = tid & 7, lrow = tid >> 3;
    int a_off[A_IT], a_iy[A_IT], a_ix[A_IT];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = m0 + lrow + it * ROWS_PER_IT;
        if (m < p.M) {
            const int b = fastdiv(m, p.fd_howo), rem = m - b * HoWo;
            const int oy = fastdiv(rem, p.fd_wo), ox = rem - oy * p.Wo;
            a_iy[it] = oy * p.stride - p.pad;
            a_ix[it] = ox * p.stride - p.pad;
            a_off[it] = (int)(b * p.in_batch_stride) + a_iy[it] * p.in_row_stride + a_ix[it] * p.in_pix_stride;
        } else {
            a_iy[it] = -(1 << 28);
            a_ix[it] = 0;
            a_off[it] = 0;
        }
    }
    int kc = slot * VE, tap = 0, dy = 0, dx = 0;
    while (kc >= p.Cin) {
        kc -= p.Cin;
        ++tap;
        if (++dx == p.kw) { dx = 0; ++dy; }
    }
    uint32_t w_off = (uint32_t)(((n0 + lrow) * p.Kpad + slot * VE) * ES);

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.weight, 0, p.w_bytes, 0x00020000);

    i32x4 ra[A_IT], rw[W_IT];
    auto load_tile = [&]() {
        const bool kvalid = tap < p.ntaps;
        const int ddy = dy * p.dil, ddx = dx * p.dil;
        const int tap_off = ddy * p.in_row_stride + ddx * p.in_pix_stride + kc;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const bool v = kvalid && (unsigned)(a_iy[it] + ddy) < (unsigned)p.H && (unsigned)(a_ix[it] + ddx) < (unsigned)p.W;
            const uint32_t off = v ? (uint32_t)(a_off[it] + tap_off) * ES : kOOB;
            ra[it] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it)
            rw[it] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off + (uint32_t)(it * ROWS_PER_IT * p.Kpad * ES), 0, 0);
        // advance to the next K slice
        w_off += BKE * ES;
        kc += BKE;
        while (kc >= p.Cin) {
            kc -= p.Cin;
            ++tap;
            if (++dx == p.kw) { dx = 0; ++dy; }
        }
    };
    auto store_tile = [&](int st) {
        char* As = smem + st * STAGE;
        char* Ws = As + A_STAGE;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int row = lrow + it * ROWS_PER_IT;
            *(i32x4*)(As + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4)) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int row = lrow + it * ROWS_PER_IT;
            *(i32x4*)(Ws + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4)) = rw[it];
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int lr = lane & 31, half = lane >> 5;
    auto compute = [&](int st) {
        const char* As = smem + st * STAGE;
        const char* Ws = As + A_STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int sk = 2 * ks + half;
            i32x4 fa[TN], fb[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int row = wn * WTN + i * 32 + lr;
                fa[i] = *(const i32x4*)(Ws + row * 128 + ((sk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int row = wm * WTM + j * 32 + lr;
                fb[j] = *(const i32x4*)(As + row * 128 + ((sk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
    };

    // ---- main loop: global->reg prefetch of slice k+1 overlaps the MFMAs of slice k ------------------
    load_tile();
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < p.nk; ++kt) {
        const bool more = kt + 1 < p.nk;
        if (more) load_tile();
        compute(kt & 1);
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    int mrow[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * WTM + j * 32 + lr;
        mrow[j] = m < p.M ? m : -1;
    }
    conv_epilogue<T, TM, TN, WTM, WTN>(p, acc, mrow, n0, wn, half);
}

// ---- whole-line epilogue (16-bit output, 64 channels per wave) --------------------------------------------------------
// In the accumulator layout a store instruction writes 32 bytes into each of 32 different 128-byte lines; a stream of such
// partial-line writes tops out near 2 TB/s (the point-wise expansion kernel: 2.65 -> 4.7 TB/s once its stores covered whole
// lines).  Here a wave parks each 32-pixel x 64-channel block of its tile in LDS ([pixel][128 B + pad], the operand stages are
// free by now) and re-reads it row-major: one store instruction = 8 pixels x 128 contiguous bytes.  Residual reads stay in
// the accumulator layout.  Selected by the host for the layers whose output stream matters (ConvArgs::line_store).
constexpr int kLineRow = 144;                  // bytes per pixel row of the parking tile (16-byte aligned, 2-way bank conflicts at most)
template <typename T, int TM>
VD3D_DEV void conv_epilogue_lines(const ConvArgs& p, f32x16 (&acc)[2][TM], const int (&mrow)[TM], int mblock0, int nw, int half, int lr, int lane,
                                  char* tile, const float* ltab, int ltn, int n0) {
    const int prow = lane >> 3, pslot = lane & 7;
    const int relu_floor = p.relu ? 0 : (int)0x80008000u;      // packed-pair floor: ReLU | identity
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = mrow[j];
        const bool mvalid = m >= 0;
        const int64_t rbase = (int64_t)(mvalid ? m : 0) * p.res_pix_stride;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = nw + i * 32 + 8 * g + 4 * half;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                {
                    const f32x4 s = *(const f32x4*)(ltab + (nb - n0)), t = *(const f32x4*)(ltab + ltn + (nb - n0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * s[e] + t[e];
                }
                if (p.residual && mvalid) {
                    const i32x2 rr = *(const i32x2*)(p.residual + (rbase + nb) * 2);
                    const uint32_t r0 = (uint32_t)(int)rr[0], r1 = (uint32_t)(int)rr[1];
                    v[0] += Fmt16<T>::lo(r0);
                    v[1] += Fmt16<T>::hi(r0);
                    v[2] += Fmt16<T>::lo(r1);
                    v[3] += Fmt16<T>::hi(r1);
                }
                *(i32x2*)(tile + lr * kLineRow + (i * 32 + 8 * g + 4 * half) * 2) =
                    i32x2{max_pk16(Fmt16<T>::pack2_1(v[0], v[1]), relu_floor), max_pk16(Fmt16<T>::pack2_1(v[2], v[3]), relu_floor)};      // round, then ReLU on the pairs (bit-identical)
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = it * 8 + prow;
            const i32x4 o = *(const i32x4*)(tile + r * kLineRow + pslot * 16);
            const int mm = mblock0 + j * 32 + r;
            if (mm < p.M) *(i32x4*)(p.out + ((int64_t)mm * p.out_pix_stride + nw) * 2 + pslot * 16) = o;
        }
    }
}

// =====================================================================================================
// v2: same tile math, but both operand tiles go global -> LDS by DMA (buffer_load_dwordx4 ... lds): no VGPR
// staging, no ds_write pass.  An LDS-DMA instruction writes wave-uniform-base + lane*16, i.e. the LDS image is
// lane-linear: piece j (1 KiB = 8 rows x 128 B) is written by one wave instruction, lane L landing on
// (row 8j + L/8, physical slot L%8).  The XOR swizzle therefore moves to the SOURCE address: lane L fetches the
// logical slot (L%8) ^ ((row/2)%8) of its row -- the same 128 contiguous bytes per row, permuted among 8 lanes, so
// global coalescing is unchanged and the fragment reads keep the conflict-free swizzled addressing.
// Zero padding still comes from the SRD bounds check (out-of-range lanes write zeros into LDS).
// KS = 1: split-K instantiation (ConvArgs::ks_*): the low-parallelism shapes of a batch-1 call -- M = 1920 pixels at stride 16 is 15 x 11
// tiles of 128 x 128 for 256 CUs -- run as tiles x splits workgroups over disjoint K ranges + one reduction pass (two launches, no
// atomics: the partials are added in split order, results do not depend on scheduling).  KS = 0 instantiations compile to the code
// they were before the parameter existed.
// KG = 2 (round 6): TWO K groups inside one workgroup.  Waves [0, NW) and [NW, 2 NW) are two copies of the NW-wave tile kernel on the SAME output tile:
// each walks its own half of the K slices through its own pair of operand stages, with its own DMA pieces -- nothing is shared but the workgroup
// barrier and, at the end, LDS: group 1 parks its accumulators, group 0 adds them (fixed order: deterministic) and runs the epilogue.  For layers whose
// tiles fill the chip only ONCE with a 4-wave workgroup (config 2's 288 -> 288 neck convs: 240 tiles of 128 x 144; every deep layer of a batch-1 call):
// one wave per SIMD has nobody to hide its DMA issue, fragment reads and barrier skew behind; split-K over workgroups gives the second wave too, but
// pays fp32 partials through HBM and a reduction launch.  SIMD mates (w, w + NW) belong to different groups; with STG > 0 group 1 issues its pieces
// at fragment STG (the staggered schedule of the strips).  MEASURED AND NOT SHIPPED (tuning build only, see plan_kgroups below): correct (every forced id
// against the oracle on the awkward shapes of tests/test_conv_tiles_gpu.py), faster per launch in the micro-benchmark, no gain in any whole configuration.
template <typename T, int BM, int BN, int WARPS_M, int WARPS_N, bool PIPE = false, int MS = 32, int RING = 0, int ABL = 0, bool HEADF = false, int KS = 0, int STG = 0, int KG = 1>
__global__ void __launch_bounds__(WARPS_M* WARPS_N * KG * 64) conv_igemm_dma_kernel(const ConvArgs p) {
    constexpr int NW = WARPS_M * WARPS_N;
    static_assert(KG == 1 || (KG == 2 && PIPE && !HEADF && STG < 100), "K groups: pipelined loop, plain epilogues");
    constexpr int ES = (int)sizeof(T);
    constexpr int VE = 16 / ES;
    constexpr int BKE = 128 / ES;
    constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
    // MS = MFMA tile edge: 32 (32x32x16, four 16-deep sub-steps per slice) or 16 (16x16x32 bf16, two 32-deep sub-steps:
    // lets a wave own e.g. 64 x 176 -- 0.68 KB of LDS fragment reads per 32x32x16-equivalent instead of 1.09 for 32 x 352)
    constexpr int TM = WTM / MS, TN = WTN / MS;
    constexpr int SPK = MS == 32 ? 2 : 4;        // 16-byte k slots consumed per sub-step by one MFMA row
    constexpr int NSUB = 8 / SPK;                // sub-steps per 128-byte slice
    static_assert(MS == 32 || (MS == 16 && PIPE && sizeof(T) == 2), "16x16x32 path: bf16, pipelined loop only");
    static_assert(WTM % MS == 0 && WTN % MS == 0, "wave tile vs MFMA shape");
    constexpr int A_PIECES = BM / 8 / NW, W_PIECES = (BN / 8 + NW - 1) / NW;  // per wave (last W round may be partial)
    constexpr int A_STAGE = BM * 128, STAGE = (BM + BN) * 128;
    static_assert((BM / 8) % NW == 0 && BN % 16 == 0, "pixel rows must split evenly into 8-row pieces per wave");
    static_assert((8 * NW) % 16 == 0, "piece stride must keep the swizzle phase constant");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // K group of this wave and the group's own operand stages (KG == 1: the whole workgroup, the whole buffer)
    const int kg = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / (NW * 64)) : 0;
    char* const smem = smem_raw + kg * (2 * STAGE);

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    int split = 0;
    if constexpr (KS) {
        const int nt = p.tiles_m * p.tiles_n;       // phase 1: grid = tiles x splits (split-major); phase 2: grid = tiles
        split = tile / nt;
        tile -= split * nt;
    }
    int tile_n = tile / p.tiles_m, tile_m = tile - tile_n * p.tiles_m;
    if (!KS && p.group_m > 0) {
        // pixel matrix far larger than L2 + Infinity Cache (the 1.8 GB DCN column matrix of BASELINE config 3): N-major order makes
        // every N tile re-stream it from HBM (8 x 1.8 GB).  Grouped order: group_m pixel tiles x ALL N tiles run back to back on
        // one XCD, so a pixel slice is fetched once per group and a weight slice once per group_m pixel tiles.
        const int per = p.group_m * p.tiles_n, g = tile / per, rr = tile - g * per;
        const int gm = p.tiles_m - g * p.group_m < p.group_m ? p.tiles_m - g * p.group_m : p.group_m;      // (last group may be short)
        tile_n = rr / gm;
        tile_m = g * p.group_m + rr - tile_n * gm;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = (int)threadIdx.x - kg * (NW * 64);       // thread / wave index inside the K group
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
    // fused KM3D head: this tile's head constants go to LDS behind the operand stages while the first slices are in flight --
    // read from global in the epilogue, the bias vectors and the 16 second-GEMM weight fragments were ~7k exposed cycles of the
    // 45k-cycle tile (cycle stamps).  [2 * STAGE, +16 KiB): W2 of head tile_n as [32 rows][512 B], 16-byte slot s of row r at
    // s ^ (r & 15);  then bias1[256] and bias2[32] fp32.
    constexpr int HD_W2 = 2 * (BM + BN) * 128, HD_B1 = HD_W2 + 16384, HD_B2 = HD_B1 + 1024;
    // (requested here, written to LDS after the prologue's DMAs are issued -- see write_ltab below)
    constexpr int HDV = HEADF ? (32 * 32) / (NW * 64) : 1;
    i32x4 hd_w2[HDV];
    f32x4 hd_b = {0.f, 0.f, 0.f, 0.f};
    if constexpr (HEADF) {
#pragma unroll
        for (int u = 0; u < HDV; ++u) {
            const int v = tid + u * NW * 64, r = v >> 5, sl = v & 31;
            hd_w2[u] = *(const i32x4*)(p.h_w2 + ((size_t)(tile_n * 32 + r) * 256 + sl * 8) * 2);
        }
        if (tid < 64) hd_b = *(const f32x4*)(p.shift + n0 + tid * 4);
        else if (tid < 72) hd_b = *(const f32x4*)(p.h_b2 + tile_n * 32 + (tid - 64) * 4);
    }
    // the tile's folded-BN constants (scale | shift of channels n0 .. n0 + BN - 1; 1 / 0 where absent) behind the operand stages:
    // the epilogues read them with ds_read instead of two global loads per accumulator quad at the end of the tile
    // (requested here, written to LDS after the prologue's DMAs are issued: the wait for these two loads must not delay them)
    constexpr int TE = (BN + NW * 64 - 1) / (NW * 64);       // table entries per thread (1, or 2 for the 4-wave 288 / 352 strips)
    float* ltab = (float*)(smem_raw + KG * 2 * (BM + BN) * 128);     // (both K groups write the same values)
    float tsc[TE], tsh[TE];
#pragma unroll
    for (int u = 0; u < TE; ++u) {
        tsc[u] = 1.f;
        tsh[u] = 0.f;
        const int i = tid + u * NW * 64;
        if (!HEADF && i < BN && n0 + i < p.Cout) {
            if (p.scale) tsc[u] = p.scale[n0 + i];
            if (p.shift) tsh[u] = p.shift[n0 + i];
        }
    }
    auto write_ltab = [&]() {
        if constexpr (HEADF) {
#pragma unroll
            for (int u = 0; u < HDV; ++u) {
                const int v = tid + u * NW * 64, r = v >> 5, sl = v & 31;
                *(i32x4*)(smem + HD_W2 + r * 512 + ((sl ^ (r & 15)) << 4)) = hd_w2[u];
            }
            if (tid < 64) *(f32x4*)(smem + HD_B1 + tid * 16) = hd_b;
            else if (tid < 72) *(f32x4*)(smem + HD_B2 + (tid - 64) * 16) = hd_b;
        }
        if constexpr (!HEADF) {
#pragma unroll
            for (int u = 0; u < TE; ++u) {
                const int i = tid + u * NW * 64;
                if (i < BN) { ltab[i] = tsc[u]; ltab[BN + i] = tsh[u]; }
            }
        }
    };

    // ---- loader state: lane -> (row within piece, logical 16-byte slot) ---------------------------------
    const int prow = lane >> 3;
    const int slot = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);  // = (L%8) ^ ((row/2)%8), constant over pieces
    int a_off[A_PIECES], a_iy[A_PIECES], a_ix[A_PIECES];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int it = 0; it < A_PIECES; ++it) {
        const int m = m0 + 8 * (wave + it * NW) + prow;
        if (m < p.M) {
            const int b = fastdiv(m, p.fd_howo), rem = m - b * HoWo;
            const int oy = fastdiv(rem, p.fd_wo), ox = rem - oy * p.Wo;
            a_iy[it] = oy * p.stride - p.pad;
            a_ix[it] = ox * p.stride - p.pad;
            a_off[it] = (int)(b * p.in_batch_stride) + a_iy[it] * p.in_row_stride + a_ix[it] * p.in_pix_stride;
        } else {
            a_iy[it] = -(1 << 28);
            a_ix[it] = 0;
            a_off[it] = 0;
        }
    }
    // K walk.  Generic: tap-major (k = tap*Cin + c), works for any Cin.  When Cin is a multiple of the 128-byte slice
    // (p.chunk_major) the slices are visited chunk-major / tap-minor instead: the kh*kw taps of one 64-channel chunk
    // run back to back, so the shifted re-reads of the same input pixels hit L2 instead of going back to the fabric /
    // Infinity Cache once per tap (measured on the 1408-channel layers: FETCH_SIZE 9x the algorithmic bytes before).
    // The packed weight needs no re-ordering: slice (chunk, tap) is the contiguous run at k = tap*Cin + chunk*64.
    int kc = slot * VE, tap = 0, dy = 0, dx = 0;
    while (kc >= p.Cin) {
        kc -= p.Cin;
        ++tap;
        if (++dx == p.kw) { dx = 0; ++dy; }
    }
    const uint32_t w_row = (uint32_t)(((n0 + 8 * wave + prow) * p.Kpad + slot * VE) * ES);
    uint32_t w_off = w_row;
    int nk_l = p.nk;                             // K slices this workgroup (K group) walks
    int nk_loop = p.nk;                          // ... and the trip count of the slice loop (K groups: the same for both; a missing slice is zeros)
    if constexpr (KS != 0 || KG > 1) {
        int kt0 = 0;                             // first slice of this split / K group: put the walk's state there
        if constexpr (KS) {
            kt0 = split * p.ks_per;
            nk_l = p.nk - kt0 < p.ks_per ? p.nk - kt0 : p.ks_per;
        }
        nk_loop = nk_l;
        if constexpr (KG > 1) {
            const int per = (nk_l + KG - 1) / KG, k0g = kg * per;
            nk_loop = per;
            kt0 += k0g;
            nk_l = nk_l - k0g < per ? (nk_l - k0g > 0 ? nk_l - k0g : 0) : per;
        }
        if (kt0 > 0) {
            if (p.chunk_major) {
                const int chunk = kt0 / p.ntaps;
                tap = kt0 - chunk * p.ntaps;
                dy = tap / p.kw;
                dx = tap - dy * p.kw;
                kc = chunk * BKE + slot * VE;
                w_off = w_row + (uint32_t)((tap * p.Cin + chunk * BKE) * ES);
            } else {
                const int klin = kt0 * BKE + slot * VE;
                tap = klin / p.Cin;
                kc = klin - tap * p.Cin;
                dy = tap / p.kw;
                dx = tap - dy * p.kw;
                w_off = w_row + (uint32_t)(kt0 * BKE * ES);
            }
        }
    }

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.weight, 0, p.w_bytes, 0x00020000);

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    // One K slice = A_PIECES + W_PIECES DMA instructions per wave, issued in 4 groups so that the main loop can spread
    // them over the four 16-deep MFMA sub-steps of the slice in flight (a DMA costs 60-180 issue cycles; issuing all of
    // them right after the barrier left the MFMA pipe idle while every wave did the same thing).
    constexpr int NPIECE = A_PIECES + W_PIECES;
    auto issue_group = [&](int st, int g, bool enable = true) {
        char* base = smem + st * STAGE + wave * 1024;
        const bool kvalid = enable && tap < p.ntaps;
        const int ddy = dy * p.dil, ddx = dx * p.dil;
        const int tap_off = ddy * p.in_row_stride + ddx * p.in_pix_stride + kc;
#pragma unroll
        for (int pi = 0; pi < NPIECE; ++pi) {
            if ((pi & 3) != g) continue;
            if (ABL == 6 && pi < A_PIECES) continue;      // timing ablations: 6 = no pixel DMA, 7 = no weight DMA
            if (ABL == 7 && pi >= A_PIECES) continue;      // (ABL >= 8: VALU injection, the DMA schedule stays complete)
            if (pi < A_PIECES) {
                const int it = pi;
                const bool v = kvalid && (unsigned)(a_iy[it] + ddy) < (unsigned)p.H && (unsigned)(a_ix[it] + ddx) < (unsigned)p.W;
                const uint32_t off = v ? (uint32_t)(a_off[it] + tap_off) * ES : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(base + it * NW * 1024), 16, off, 0, 0, 0);
            } else {
                int it = pi - A_PIECES;
                if constexpr (PIPE) {
                    // branch-free (the pipelined loop wants one basic block): a wave beyond the partial last round
                    // repeats its previous piece (same source, same destination); a disabled issue writes zeros
                    if ((BN / 8) % NW != 0 && wave + it * NW >= BN / 8) it -= 1;
                    const uint32_t off = enable ? w_off + (uint32_t)(it * NW * 8 * p.Kpad * ES) : kOOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(base + A_STAGE + it * NW * 1024), 16, off, 0, 0, 0);
                } else {
                    if ((BN / 8) % NW == 0 || wave + it * NW < BN / 8)   // wave-uniform guard for a partial last round
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(base + A_STAGE + it * NW * 1024), 16,
                                                                 w_off + (uint32_t)(it * NW * 8 * p.Kpad * ES), 0, 0, 0);
                }
            }
        }
    };
    auto advance_k = [&]() {
        if (p.chunk_major) {
            ++tap;
            if (++dx == p.kw) { dx = 0; ++dy; }
            if (tap == p.ntaps) { tap = 0; dx = 0; dy = 0; kc += BKE; }
            w_off = w_row + (uint32_t)((tap * p.Cin + kc - slot * VE) * ES);
        } else {
            w_off += BKE * ES;
            kc += BKE;
            while (kc >= p.Cin) {
                kc -= p.Cin;
                ++tap;
                if (++dx == p.kw) { dx = 0; ++dy; }
            }
        }
    };

    typename MmaShape<T, MS>::acc_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < MS * MS / 64; ++e) acc[i][j][e] = 0.f;

    const int lr = lane & (MS - 1), half = lane / MS;   // half: k-slot group of the lane (0..1 for MS 32, 0..3 for MS 16)
    auto compute = [&](int st, bool more) {
      if constexpr (MS == 32) {
        const char* As = smem + st * STAGE;
        const char* Ws = As + A_STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int sk = 2 * ks + half;
            i32x4 fa[TN], fb[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int row = wn * WTN + i * 32 + lr;
                fa[i] = *(const i32x4*)(Ws + row * 128 + ((sk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int row = wm * WTM + j * 32 + lr;
                fb[j] = *(const i32x4*)(As + row * 128 + ((sk ^ ((row >> 1) & 7)) << 4));
            }
            if (more) issue_group(st ^ 1, ks);    // fills the fragment-read latency; lands during the MFMAs below
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
        if (more) advance_k();
      }
    };

    if constexpr (!PIPE) {
#pragma unroll
        for (int g = 0; g < 4; ++g) issue_group(0, g);
        advance_k();
        write_ltab();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk_l; ++kt) {
            compute(kt & 1, kt + 1 < nk_l);     // DMA of slice k+1 is spread over the MFMAs of slice k
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        // Software-pipelined main loop: the only point where the MFMA pipe can drain is the barrier skew.
        //  * fragments are consumed and refilled in place: right after the MFMAs of weight row-block i have issued, fa[i]
        //    is reloaded for the NEXT 16-deep sub-step (pixel fragments are double buffered), so no sub-step starts with a
        //    burst of exposed ds_reads;
        //  * the per-slice barrier sits before the LAST sub-step: by then the wave holds that sub-step's fragments in
        //    registers, so after the barrier the stage is free (DMA of slice t+2 goes into it) and the first fragments of
        //    slice t+1 are fetched under the last sub-step's MFMAs.
        auto ld_w = [&](int st, int ks, int i) {
            const int row = wn * WTN + i * MS + lr;
            return *(const i32x4*)(smem + st * STAGE + A_STAGE + row * 128 + (((SPK * ks + half) ^ ((row >> 1) & 7)) << 4));
        };
        auto ld_a = [&](int st, int ks, int j) {
            const int row = wm * WTM + j * MS + lr;
            return *(const i32x4*)(smem + st * STAGE + row * 128 + (((SPK * ks + half) ^ ((row >> 1) & 7)) << 4));
        };
#pragma unroll
        for (int g = 0; g < 4; ++g) issue_group(0, g, KG == 1 || nk_l > 0);
        advance_k();
        // STG > 0 (experiment, round 4): the two waves of a SIMD (w and w + NW / 2) issue their DMA pieces at DIFFERENT points of the slice
        // -- behind the barrier both waves of a SIMD are in lock step, so both sat in their DMA bursts (60 - 185 issue cycles per piece, ten
        // pieces per slice and wave) at the same time and the MFMA pipe idled; the upper half issues all four groups of slice t + 1 at
        // fragment STG of slice t instead (its stage was freed by the barrier of slice t - 1), the lower half keeps the schedule below
        // STG < 100: two phases (upper half at fragment STG); STG = 100 + d: FOUR phases -- wave pairs (0,1) (2,3) (4,5) (6,7) at the base
        // schedule, d, 2d, 3d: SIMD mates (w, w + 4) are 2d fragments apart and the CU's texture path sees four bursts of 20 pieces
        const int ph = KG > 1 ? (STG > 0 ? kg : 0) : (STG >= 100 ? (wave >> 1) : (STG > 0 && wave >= NW / 2 ? 1 : 0));
        const bool hiw = ph != 0;
        if (!hiw) {
            issue_group(1, 0, nk_l > 1);
            issue_group(1, 1, nk_l > 1);
        }
        write_ltab();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // weight fragments live in a ring of R registers refilled R fragments ahead (R | 4*TN keeps the register <->
        // fragment assignment identical in every slice); pixel fragments are double buffered per sub-step.
        constexpr int R = RING ? RING : (TM == 1 ? 4 : (MS == 16 ? 2 : TN));
        static_assert((NSUB * TN) % R == 0 && R <= TN, "fragment ring");
        constexpr int F0 = NSUB * TN - R;      // first fragment whose refill comes from the NEXT slice: the barrier sits here
        i32x4 fa[R], fb[2][TM];
        [[maybe_unused]] float abl_v[4] = {1.f, 2.f, 3.f, 4.f}, abl_w = 0.5f;     // (ABL 8 - 10: operands of the injected VALU work)
#pragma unroll
        for (int i = 0; i < R; ++i) fa[i] = ld_w(0, 0, i);
#pragma unroll
        for (int j = 0; j < TM; ++j) fb[0][j] = ld_a(0, 0, j);
        constexpr int NMFMA = sizeof(T) == 2 ? TM : 4 * TM;               // MFMA instructions per weight fragment
        auto group_size = [](int g) { return (NPIECE - g + 3) / 4; };
        for (int kt = 0; kt < nk_loop; ++kt) {
            const int st = kt & 1;
            const bool more1 = kt + 1 < nk_l, more2 = kt + 2 < nk_l;
#pragma unroll
            for (int f = 0; f < NSUB * TN; ++f) {
                const int ks = f / TN, i = f - ks * TN;
                if (f == F0) {
                    // Every read of this stage has been issued (lgkmcnt(0): and has completed), and slice t+1 must have
                    // landed for everybody: after the barrier the stage is free for slice t+2 and the ring starts
                    // refilling from slice t+1.
                    // ABL != 0: timing ablations for tools/bench_conv.py (WRONG results): 1 = no barrier, 2 = no DMA wait, 3 = neither
                    if constexpr (ABL == 0 || ABL == 1 || ABL >= 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if constexpr (ABL == 0 || ABL == 2 || ABL >= 8) __builtin_amdgcn_s_barrier();   // 4: MFMA only, 5: MFMA + ds_read
                    asm volatile("" ::: "memory");
                    // DMA of slice t+2: half of it here, the rest at the top of the next slice -- everything is in flight
                    // within the first tenth of a slice, i.e. has ~0.9 slice times to land before its barrier (PMC: with
                    // the pieces spread evenly over the slice a third of the wave cycles were spent parked at that barrier)
                    if constexpr (ABL < 4 || ABL >= 8) {
                        if (!hiw) { issue_group(st, 0, more2); issue_group(st, 1, more2); }
                    }
                }
                if ((i == 0 && ks < NSUB - 1) || f == F0) {
                    // pixel fragments of the next sub-step (past the last slice they read a dead stage; never consumed)
                    const int k2 = f == F0 ? NSUB - 1 : ks;
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        if constexpr (ABL != 4) fb[(k2 & 1) ^ 1][j] = ld_a(k2 < NSUB - 1 ? st : st ^ 1, (k2 + 1) % NSUB, j);
                        else fb[(k2 & 1) ^ 1][j] = fb[k2 & 1][j];
                    __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
                }
                if (i == 0 && ks == 0) {
                    if (!hiw) {
                        if constexpr (ABL < 4 || ABL >= 8) { issue_group(st ^ 1, 2, more1); issue_group(st ^ 1, 3, more1); }
                        advance_k();
                    }
                }
                if constexpr (STG > 0) {
                    constexpr int NPH = STG >= 100 ? 4 : 2, DPH = STG >= 100 ? STG - 100 : STG;
#pragma unroll
                    for (int k = 1; k < NPH; ++k)
                        if (f == k * DPH && ph == k) {
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) issue_group(st ^ 1, g4, more1);
                            advance_k();
                        }
                }
#pragma unroll
                for (int j = 0; j < TM; ++j) MmaShape<T, MS>::run(fa[f % R], fb[ks & 1][j], acc[i][j]);
                if constexpr (ABL >= 8) {
                    // timing experiment (tuning build only; results stay correct): INJECT the VALU issue load that blending the pixel operand inside this
                    // loop would add (config 3's DCN sampling fused into the strip GEMM).  Four independent fma chains
                    // (the blend has that much ILP), no memory traffic: 8 = 17 per fragment (~304 per 64-deep slice and wave on the 288 strip: the
                    // bf16 blend of this wave's 32 x 64 share incl. unpack / pack), 9 = 9 (~160: an fp16 blend on v_fma_mix); 20 + n = n per fragment
                    // (20 = none: only the empty volatile asm, i.e. its effect on the compiler's schedule)
                    constexpr int NV = ABL == 8 ? 17 : (ABL == 9 ? 9 : ABL - 20);
                    if constexpr (NV == 0) asm volatile("" ::: "memory");
#pragma unroll
                    for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(abl_v[q & 3]) : "v"(abl_w));
                }
                const int nf = f + R, nks = nf / TN, ni = nf - nks * TN;
                if constexpr (ABL != 4) fa[f % R] = ld_w(nks < NSUB ? st : st ^ 1, nks % NSUB, ni);
                // pin the schedule (otherwise the scheduler sinks every ds_read to its use): MFMAs of this fragment, its
                // ring refill, then this fragment's share of the DMA pieces issued in the sub-step (spread evenly)
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                const int i0 = ks == NSUB - 1 ? TN - R : 0, nfr = TN - i0;
                const int D = ks == 0 ? group_size(2) + group_size(3) : (ks == NSUB - 1 ? group_size(0) + group_size(1) : 0);
                const int o = i - i0;
                // front-loaded: ceil(D / nfr) pieces per fragment but at least 2, until the sub-step's pieces are placed
                const int per = (D + nfr - 1) / nfr > 2 ? (D + nfr - 1) / nfr : 2;
                const int q = o >= 0 ? (D - per * o > per ? per : (D - per * o > 0 ? D - per * o : 0)) : 0;
                if (q >= 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (q >= 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (q >= 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (q >= 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (q >= 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (q >= 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
    }
    if constexpr (KG > 1) {
        // the two K groups' partial sums meet in LDS (register order: one conflict-free 1 KiB row per vector and wave), at the END of the operand
        // stages -- group 0's epilogue parks its output lines at the start
        constexpr int AVK = MS * MS / 64 / 4, NVK = TN * TM * AVK;
        constexpr int kRedOff = KG * 2 * STAGE - NW * NVK * 1024;
        static_assert(kRedOff >= NW * 32 * 160 && kRedOff % 16 == 0, "K groups: the partial sums must not overlap the epilogue's line buffers");
        f32x4* red = (f32x4*)(smem_raw + kRedOff) + wave * NVK * 64 + lane;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (disabled look-ahead pieces write zeros into the stages)
        __syncthreads();                                      // every wave of both groups is done with the operand stages
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int v = 0; v < AVK; ++v)
                        red[((i * TM + j) * AVK + v) * 64] = f32x4{acc[i][j][4 * v], acc[i][j][4 * v + 1], acc[i][j][4 * v + 2], acc[i][j][4 * v + 3]};
        }
        __syncthreads();
        if (kg == 1) return;                                  // (no workgroup barrier below this point in a KG build)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int v = 0; v < AVK; ++v) {
                    const f32x4 q = red[((i * TM + j) * AVK + v) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * v + e] += q[e];
                }
    }
    int mrow[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * WTM + j * MS + lr;
        mrow[j] = m < p.M ? m : -1;
    }
    if constexpr (KS) {
        // partials in REGISTER order: vector v of wave w of workgroup (split, tile) is one fully coalesced 1 KiB row of the workspace;
        // splitk_reduce_kernel (below) adds the splits in order and runs the epilogue
        constexpr int AV = MS * MS / 64 / 4, NV = TN * TM * AV;           // f32x4 vectors per accumulator tile / per lane
        const int nt = p.tiles_m * p.tiles_n, tl = tile_n * p.tiles_m + tile_m;
        f32x4* dst = (f32x4*)p.ks_ws + (((size_t)(split * nt + tl) * NW + wave) * NV) * 64 + lane;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int v = 0; v < AV; ++v)
                    dst[((i * TM + j) * AV + v) * 64] = f32x4{acc[i][j][4 * v], acc[i][j][4 * v + 1], acc[i][j][4 * v + 2], acc[i][j][4 * v + 3]};
        return;
    }
    if constexpr (HEADF) {
        // ---- fused KM3D head, first version (VD3D_HEAD_PARKED=1; the product path is km3d_head_conv.hip): this N tile is head h = tile_n.  bias + ReLU, round to bf16 (the rounding point of the unfused
        // path's `mid` tensor), park the 256 x 256 tile in LDS, then out_h = tile x W2_h + b2_h on the matrix cores.  The
        // 9 x 256-channel intermediate (4 GB at 16 x 128 x 440) never exists in HBM.
        static_assert(BM == 256 && BN == 256 && MS == 32 && sizeof(T) == 2, "fused head: 256 x 256 bf16 tiles");
        __syncthreads();                                   // every wave is done with the operand stages
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int r = wm * WTM + j * 32 + lr;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = wn * WTN + i * 32 + 8 * g + 4 * half;       // channel inside the head
                    const f32x4 sh = *(const f32x4*)(smem + HD_B1 + c * 4);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][j][4 * g + e] + sh[e], 0.f);
                    i32x2 o;
                    o[0] = Fmt16<T>::pack2(v[0], v[1]);
                    o[1] = Fmt16<T>::pack2(v[2], v[3]);
                    const int slot16 = c >> 3;
                    *(i32x2*)(smem + r * 512 + ((slot16 ^ (r & 15)) << 4) + half * 8) = o;
                }
        }
        __syncthreads();
        const int h = tile_n;
        const int r2 = wave * 32 + lr;                      // wave w owns pixel rows 32w .. 32w+31 for the second GEMM
        f32x16 acc2;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int slot16 = 2 * ks + half;
            const i32x4 fa2 = *(const i32x4*)(smem + HD_W2 + lr * 512 + ((slot16 ^ (lr & 15)) << 4));
            const i32x4 fb2 = *(const i32x4*)(smem + r2 * 512 + ((slot16 ^ (r2 & 15)) << 4));
            Fmt16<T>::mfma32(fa2, fb2, acc2);
        }
        // The wave's 32 pixels x nh outputs are one contiguous run of the [M][nh] map.  Stored from the accumulator layout a lane
        // writes nh scattered floats (16 store instructions of 32 lines each); parked as [pixel][nh] in the wave's own rows of the
        // tile (its second-GEMM reads are done) the run goes out with consecutive lanes on consecutive floats: whole lines.
        const int nh = p.h_n[h];
        float* stage = (float*)(smem + wave * 32 * 512);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = 8 * g + 4 * half + e;
                if (n < nh) stage[lr * nh + n] = acc2[4 * g + e] + *(const float*)(smem + HD_B2 + n * 4);
            }
        const int mw = m0 + wave * 32;                   // first pixel of this wave's run
        const int nrun = (p.M - mw < 32 ? (p.M - mw > 0 ? p.M - mw : 0) : 32) * nh;
        float* dst = p.h_out[h] + (int64_t)mw * nh;
        for (int i = lane; i < nrun; i += 64) dst[i] = stage[i];
        return;
    }
    if constexpr (MS == 32 && WTN == 64 && sizeof(T) == 2) {
        if (p.line_store) {          // (host: 16-bit output, Cout % 64 == 0, aligned rows -- every wave's 64-channel strip is whole)
            if constexpr (KG == 1) __syncthreads();         // every wave is done with the operand stages
            if (n0 + wn * WTN < p.Cout)
                conv_epilogue_lines<T, TM>(p, acc, mrow, m0 + wm * WTM, n0 + wn * WTN, half, lr, lane, smem + wave * (32 * kLineRow), ltab, BN, n0);
            return;
        }
    }
    if constexpr (MS == 16 && WTN % 8 == 0 && !HEADF) {
        // strips: whole-line stores when the wave's channel strip lies inside Cout and the output rows are 16-byte aligned
        if (p.strip_lines) {
            if constexpr (KG == 1) __syncthreads();         // every wave is done with the operand stages (workgroup-uniform branch)
            if (n0 + wn * WTN + WTN <= p.Cout) {
                conv_epilogue16_lines<T, TM, TN, WTN>(p, acc, mrow, m0 + wm * WTM, n0, n0 + wn * WTN, half, lr, lane,
                                                      smem + wave * (16 * (WTN * 2 + 16)), ltab, BN);
                return;
            }
        }
    }
    if constexpr (MS == 32) conv_epilogue<T, TM, TN, WTM, WTN>(p, acc, mrow, n0, wn, half, ltab, BN);
    else conv_epilogue16<T, TM, TN, WTN>(p, acc, mrow, n0, wn, half, ltab, BN);
}

// =====================================================================================================
// v3 "halo" kernel for 3x3 / stride 1 / pad 1 convolutions (95 % of the FLOPs of every detector here).
// In the implicit-GEMM view the pixel operand of tap (ky,kx) is the same activation shifted by one pixel, so the v2
// kernel re-fetches every input pixel 9 times into LDS (once per tap).  LDS fill bandwidth (L2 -> LDS DMA) is what bounds
// the small-channel layers (layer1: ~52 flop per filled byte).  Here the workgroup owns a TH x TW patch of output pixels
// of one image and stages, per 64-channel chunk, the (TH+2) x (TW+2) input halo ONCE; the nine taps read their pixel
// fragments from the halo at shifted rows.  K is walked chunk-outer / tap-inner (the packed weight layout
// K = tap*Cin + c already makes every (chunk, tap) slice a contiguous 128-byte run).  Fill traffic per 64-channel
// chunk drops from 9*(BM + BN) rows to ~1.3*BM + 9*BN rows.
template <int N> VD3D_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int TH, int TW, int BN, int WARPS_M, int WARPS_N, int STAGES>
__global__ void __launch_bounds__(WARPS_M* WARPS_N * 64) conv_halo_kernel(const ConvArgs p) {
    constexpr int NW = WARPS_M * WARPS_N;
    constexpr int ES = (int)sizeof(T);
    constexpr int VE = 16 / ES;
    constexpr int BKE = 128 / ES;
    constexpr int BM = TH * TW;
    constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int HW2 = TW + 2;
    constexpr int HR = (TH + 2) * HW2;                 // halo pixels
    constexpr int H_PIECES_TOT = (HR + 7) / 8;
    constexpr int H_PIECES = (H_PIECES_TOT + NW - 1) / NW;
    constexpr int W_PIECES = BN / 8 / NW;               // identical for every wave: the counted vmcnt below relies on it
    constexpr int H_STAGE = H_PIECES_TOT * 1024, W_STAGE = BN * 128;
    static_assert(BM % (32 * WARPS_M) == 0 && BN % (32 * WARPS_N) == 0 && (8 * NW) % 16 == 0, "tile shape");
    static_assert((BN / 8) % NW == 0 && STAGES >= 2 && (STAGES - 2) * W_PIECES <= 63, "weight pieces per wave / vmcnt range");
    static_assert((TW & (TW - 1)) == 0, "TW must be a power of two");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Hs = smem;                       // 2 halo stages
    char* Ws = smem + 2 * H_STAGE;         // STAGES weight stages (ring)

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = tile / p.tiles_m, tile_m = tile - tile_n * p.tiles_m;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int b = tile_m / (tiles_x * tiles_y), trem = tile_m - b * (tiles_x * tiles_y);
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem - (trem / tiles_x) * tiles_x) * TW;
    const int n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;

    // ---- DMA lane state ---------------------------------------------------------------------------------
    const int prow = lane >> 3;
    const int slot = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
    // Swizzle key of a HALO row.  A wave reads the rows of 32 consecutive pixels (+ a tap shift); with TW = 16 those are two
    // runs of 16 rows 18 apart, and the plain (row/2)%8 key lets the ds_read_b128 lane groups that straddle the two runs hit
    // the same bank twice (PMC: SQ_LDS_BANK_CONFLICT = 1/3 of the LDS cycles of the 8x16 kernels).  Keying on the pixel
    // index py*TW + hx instead makes the 32 keys consecutive again for every tap shift.
    auto hkey = [&](int row) {
        // (measured: +1.3 % on the 8x16x256 tile, +5 % on the resident kernel, but -6...-10 % on the 8x16x128 tiles: kept
        // only where it pays)
        if constexpr (TW == 16 && BN == 256) { const int hy = row / HW2; return ((hy * TW + row - hy * HW2) >> 1) & 7; }
        else return (row >> 1) & 7;
    };
    uint32_t h_off[H_PIECES];   // byte offset of this lane's halo pixel (channel 0 of the slice, logical slot), or OOB
#pragma unroll
    for (int it = 0; it < H_PIECES; ++it) {
        const int hr = 8 * (wave + it * NW) + prow;
        const int hy = hr / HW2, hx = hr - hy * HW2;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        const bool v = hr < HR && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int hslot = (lane & 7) ^ hkey(hr);
        h_off[it] = v ? (uint32_t)((int)(b * p.in_batch_stride) + iy * p.in_row_stride + ix * p.in_pix_stride + hslot * VE) * ES : kOOB;
    }
    const uint32_t w_base = (uint32_t)(((n0 + 8 * wave + prow) * p.Kpad + slot * VE) * ES);

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.weight, 0, p.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    // DMA issue is split into 4 groups (g = piece index mod 4) so the main loop can spread one step's pieces over its four
    // 16-deep MFMA sub-steps instead of issuing them all right after the barrier.  g < 0: every piece.
    auto issue_halo = [&](int chunk, int g) {
        char* base = Hs + (chunk & 1) * H_STAGE + wave * 1024;
        const uint32_t coff = (uint32_t)(chunk * BKE * ES);
#pragma unroll
        for (int it = 0; it < H_PIECES; ++it) {
            if (g >= 0 && (it & 3) != g) continue;
            if (H_PIECES_TOT % NW == 0 || wave + it * NW < H_PIECES_TOT) {
                const uint32_t off = h_off[it] == kOOB ? kOOB : h_off[it] + coff;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(base + it * NW * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    auto issue_w = [&](int step, int g) {     // step = chunk * 9 + tap
        const int chunk = step / 9, tap = step - chunk * 9;
        char* base = Ws + (step % STAGES) * W_STAGE + wave * 1024;
        const uint32_t koff = (uint32_t)((tap * p.Cin + chunk * BKE) * ES);
#pragma unroll
        for (int it = 0; it < W_PIECES; ++it) {
            if (g >= 0 && ((it + 1) & 3) != g) continue;      // offset by one: halo piece 0 and weight piece 0 in different groups
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(base + it * NW * 1024), 16,
                                                     w_base + koff + (uint32_t)(it * NW * 8 * p.Kpad * ES), 0, 0, 0);
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int lr = lane & 31, half = lane >> 5;
    int hrow0[TM];   // halo row of this lane's pixel for tap (0,0)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int pp = wm * WTM + j * 32 + lr;
        hrow0[j] = (pp / TW) * HW2 + (pp & (TW - 1));
    }
    const int nchunk = p.Cin / BKE;
    const int nsteps = nchunk * 9;
    auto compute = [&](int step, int chunk, int tap) {
        const char* Hb = Hs + (chunk & 1) * H_STAGE;
        const char* Wb = Ws + (step % STAGES) * W_STAGE;
        const int dy = tap / 3, dx = tap - dy * 3;
        const int shift = dy * HW2 + dx;
        const bool next_halo = tap == 0 && chunk + 1 < nchunk;
        const bool next_w = step + STAGES - 1 < nsteps;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int sk = 2 * ks + half;
            i32x4 fa[TN], fb[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int row = wn * WTN + i * 32 + lr;
                fa[i] = *(const i32x4*)(Wb + row * 128 + ((sk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int row = hrow0[j] + shift;
                fb[j] = *(const i32x4*)(Hb + row * 128 + ((sk ^ hkey(row)) << 4));
            }
            if (next_halo) issue_halo(chunk + 1, ks);
            if (next_w) issue_w(step + STAGES - 1, ks);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
    };

    // ---- STAGES-deep weight ring with a COUNTED vmcnt: at step t only W(t) (and everything older, incl. the halo issued
    // nine steps earlier) must have landed; the DMAs of steps t+1 .. t+STAGES-2 stay in flight across the barrier, so the
    // L2 -> LDS latency (several hundred cycles) is covered by STAGES-2 slices of MFMA work instead of one.
    // scale | shift of this N tile go to LDS for the epilogue, as in the tile kernels: requested here, ahead of the first DMAs (older
    // than everything the counted waits below track), written once those are in flight
    static_assert(BN <= NW * 64, "one table entry per thread");
    float* ltab = (float*)(smem + 2 * H_STAGE + STAGES * W_STAGE);
    float tsc = 1.f, tsh = 0.f;
    if (tid < BN && n0 + tid < p.Cout) {
        if (p.scale) tsc = p.scale[n0 + tid];
        if (p.shift) tsh = p.shift[n0 + tid];
    }
    issue_halo(0, -1);
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0)
        if (s0 < nsteps) issue_w(s0, -1);
    if (tid < BN) { ltab[tid] = tsc; ltab[BN + tid] = tsh; }       // (visible after the first step's barrier)
    for (int step = 0; step < nsteps; ++step) {
        if (step + STAGES - 2 < nsteps) wait_vmcnt<(STAGES - 2) * W_PIECES>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int chunk = step / 9, tap = step - chunk * 9;
        compute(step, chunk, tap);
    }
    int mrow[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int pp = wm * WTM + j * 32 + lr;
        const int y = ty0 + pp / TW, x = tx0 + (pp & (TW - 1));
        mrow[j] = (y < p.H && x < p.W) ? (b * p.H + y) * p.W + x : -1;
    }
    conv_epilogue<T, TM, TN, WTM, WTN>(p, acc, mrow, n0, wn, half, ltab, BN);
}


template <typename T, int TH, int TW, int BN, int WARPS_M, int WARPS_N, int STAGES = 4>
int launch_halo(ConvArgs& a, hipStream_t stream) {
    constexpr int NT = WARPS_M * WARPS_N * 64;
    constexpr int HR = (TH + 2) * (TW + 2);
    constexpr int LDS = 2 * ((HR + 7) / 8) * 1024 + STAGES * BN * 128 + 2 * BN * 4;     // halo stages, weight stages, scale | shift table
    static_assert(LDS <= 160 * 1024, "halo tile does not fit the 160 KiB LDS");
    a.tiles_m = a.B * ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
    a.tiles_n = (a.Cout + BN - 1) / BN;
    static Vd3dLdsLimit lim;
    auto kern = conv_halo_kernel<T, TH, TW, BN, WARPS_M, WARPS_N, STAGES>;
    if (const int rc = vd3d_raise_lds_limit((const void*)kern, LDS, lim, "hipFuncSetAttribute(conv_halo)")) return rc;
    const int64_t grid = (int64_t)a.tiles_m * a.tiles_n;
    if (grid <= 0 || grid > 0x7fffffff) return VD3D_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), LDS, stream, a);
    return vd3d_check_launch("conv_halo");
}

template <typename T, int BM, int BN, int WARPS_M, int WARPS_N, bool DMA = false, bool PIPE = false, int MS = 32, int RING = 0, int ABL = 0, bool HEADF = false, int STG = 0, int KG = 1>
int launch(ConvArgs& a, hipStream_t stream) {
    constexpr int NT = WARPS_M * WARPS_N * KG * 64;
    constexpr int LDS = KG * 2 * (BM + BN) * 128 + (HEADF ? 16384 + 1024 + 128 : (DMA ? 2 * BN * 4 : 0));    // + fused head: W2 fragments and biases; DMA tiles: scale | shift table
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    static Vd3dLdsLimit lim;
    void (*kern)(const ConvArgs);
    if constexpr (DMA) kern = conv_igemm_dma_kernel<T, BM, BN, WARPS_M, WARPS_N, PIPE, MS, RING, ABL, HEADF, 0, STG, KG>;
    else kern = conv_igemm_kernel<T, BM, BN, WARPS_M, WARPS_N>;
    if (const int rc = vd3d_raise_lds_limit((const void*)kern, LDS, lim, "hipFuncSetAttribute(conv_igemm)")) return rc;
    const int64_t grid = (int64_t)a.tiles_m * a.tiles_n;
    if (grid <= 0 || grid > 0x7fffffff) return VD3D_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), LDS, stream, a);
    return vd3d_check_launch("conv_igemm");
}

// EXPERIMENTS of the tuning build (environment variables read once; never in the product library): routing decisions that were taken for ONE step in flight,
// re-tested with two (bench.py --in-flight 2), where a badly-filled round is filled by the other step.  VD3D_X_FILL_BLIND: a one-round tile choice is scored
// without its fill, and the strips are not split to fill the chip; VD3D_X_NO_SMALL_SPLIT: no split over workgroups when the tiles already fill half the chip;
// VD3D_X_L3_TILE: ResNet layer3 on 256 x 128 tiles also when they make one round.
// MEASURED (headline, two in flight, same box, alternating, ms per step): none 3.462 / 3.469; FILL_BLIND 3.470 / 3.485; NO_SMALL_SPLIT 3.461 / 3.463; L3_TILE 3.470 / 3.436 --
// nothing moves: the routes chosen for one step in flight stand.
#ifdef VD3D_TUNING
static bool x_flag(const char* name) { const char* v = getenv(name); return v && *v && *v != '0'; }
static bool x_fill_blind() { static const bool f = x_flag("VD3D_X_FILL_BLIND"); return f; }
static bool x_no_small_split() { static const bool f = x_flag("VD3D_X_NO_SMALL_SPLIT"); return f; }
static bool x_l3_tile() { static const bool f = x_flag("VD3D_X_L3_TILE"); return f; }
#else
static constexpr bool x_fill_blind() { return false; }
static constexpr bool x_no_small_split() { return false; }
static constexpr bool x_l3_tile() { return false; }
#endif

// ---- split-K (KS = 1 instantiations): two launches of the same kernel, phase 1 over tiles x splits, phase 2 over tiles -------------
struct SplitPlan { int bn = 0; int splits = 1; int64_t ws_bytes = 0; int kg = 1; };     // bn: 128 (128 x 128 tiles) | 64 (128 x 64) | 144 | 288, 0 = do not split; kg = 2: workgroups of two K groups

// Low-parallelism shapes only: fewer 128-row tiles than ~1.2 x CUs (a batch-1 / batch-2 call at stride 16 or 32) AND a K deep enough
// (>= 24 slices) that the extra reduction launch (~6 us) pays.  Splits fill two workgroup slots per CU, keep >= 4 slices per split.
// Of the batched configurations only the two thin cls-tower convs of config 2 qualify (1408 -> 256 and 256 -> 144 at 8 x 24 x 80: 240 / 120 tiles;
// measured there: 114 -> 105 us against the halo tile, 38.6 -> 32.7 us against the 128 x 128 tile); every other batched layer has thousands of tiles.
static SplitPlan plan_splitk(const ConvArgs& a, bool forced) {
    SplitPlan pl;
    const int cus = vd3d_device_cu_count() > 0 ? vd3d_device_cu_count() : 256;
    const int bn = a.Cout <= 64 ? 64 : 128;
    const int64_t tiles = (int64_t)((a.M + 127) / 128) * ((a.Cout + bn - 1) / bn);
    if (!forced && (a.nk < 24 || tiles * 10 > (int64_t)cus * 12)) return pl;
    if (!forced && x_no_small_split() && tiles * 2 >= cus) return pl;
    if (a.nk < 2) return pl;
    int s = (int)((2 * (int64_t)cus) / tiles);
    if (s > a.nk / 4) s = a.nk / 4;
    if (s > 32) s = 32;
    if (s < 2) { if (!forced) return pl; s = 2; }
    const int per = (a.nk + s - 1) / s;
    pl.splits = (a.nk + per - 1) / per;          // every split is non-empty
    if (pl.splits < 2) return pl;
    pl.bn = bn;
    pl.ws_bytes = (int64_t)pl.splits * tiles * 128 * bn * 4;
    return pl;
}

// Second pass of a split-K convolution: one thread per f32x4 of the (register-ordered) partial tiles -- reads are whole 1 KiB rows, the
// splits are added in index order (deterministic), then the ordinary epilogue: folded BN, residual, ReLU, 16-bit or fp32 store.
// Fully parallel (M x N / 4 threads): with the reduction inside the tile kernel (one workgroup per tile walking the splits one after
// the other) a 30-tile x 17-split layer spent 24 us here, more than in its MFMA pass.
template <typename T, int BN, int WARPS_M, int WARPS_N, int BM = 128, int MS = 32>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ConvArgs p, int64_t total) {
    constexpr int NW = WARPS_M * WARPS_N, WTM = BM / WARPS_M, WTN = BN / WARPS_N, TM = WTM / MS, TN = WTN / MS, AV = MS * MS / 256, NV = TN * TM * AV;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;           // vector index inside one split's partial image
    if (g >= total) return;
    const f32x4* src = (const f32x4*)p.ks_ws + g;
    f32x4 a = src[0];
    int sp = 1;
    for (; sp + 3 < p.ks_n; sp += 4) {                                   // four loads in flight; added in split order
        const f32x4 x0 = src[(int64_t)sp * total], x1 = src[(int64_t)(sp + 1) * total], x2 = src[(int64_t)(sp + 2) * total], x3 = src[(int64_t)(sp + 3) * total];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = (((a[e] + x0[e]) + x1[e]) + x2[e]) + x3[e];
    }
    for (; sp < p.ks_n; ++sp) {
        const f32x4 x = src[(int64_t)sp * total];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += x[e];
    }
    const int lane = (int)(g & 63);
    int r = (int)(g >> 6);
    const int vi = r % NV; r /= NV;
    const int wave = r % NW, tl = r / NW;
    const int tile_n = tl / p.tiles_m, tile_m = tl - tile_n * p.tiles_m;
    const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
    int m, n;
    if constexpr (MS == 32) {
        const int i = vi / (TM * 4), j = (vi >> 2) % TM, v = vi & 3, lr = lane & 31, half = lane >> 5;
        m = tile_m * BM + wm * WTM + j * 32 + lr;
        n = tile_n * BN + wn * WTN + i * 32 + 8 * v + 4 * half;
    } else {                                     // 16x16x32 tiles: one f32x4 per tile, lane (pixel l16, quad q) holds channels 4q .. 4q + 3
        const int i = vi / TM, j = vi - i * TM;
        m = tile_m * BM + wm * WTM + j * 16 + (lane & 15);
        n = tile_n * BN + wn * WTN + i * 16 + 4 * (lane >> 4);
    }
    if (m >= p.M || n >= p.Cout) return;
    constexpr int ES = (int)sizeof(T);
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = n + e < p.Cout ? n + e : p.Cout - 1;
        o[e] = a[e] * (p.scale ? p.scale[c] : 1.f) + (p.shift ? p.shift[c] : 0.f);
    }
    if (p.residual) {
        const char* rp = p.residual + ((int64_t)m * p.res_pix_stride + n) * ES;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < p.Cout) o[e] += ElemTraits<T>::to_f(*(const T*)(rp + e * ES));
    }
    if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    const int64_t ob = (int64_t)m * p.out_pix_stride + n;
    if (ES == 4 || p.out_f32) {
        float* op = (float*)p.out + ob;
        if (p.vec_epilogue) *(f32x4*)op = f32x4{o[0], o[1], o[2], o[3]};
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < p.Cout) op[e] = o[e];
        }
    } else {
        T* op = (T*)p.out + ob;
        if (p.vec_epilogue) *(i32x2*)op = i32x2{Fmt16<T>::pack2(o[0], o[1]), Fmt16<T>::pack2(o[2], o[3])};
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < p.Cout) op[e] = ElemTraits<T>::from_f(o[e]);
        }
    }
}

template <typename T, int BN, int WARPS_M, int WARPS_N, bool PIPE, int BM = 128, int MS = 32, int RING = 0, int KG = 1>
int launch_splitk(ConvArgs& a, hipStream_t stream, const SplitPlan& pl) {
    constexpr int NT = WARPS_M * WARPS_N * KG * 64;
    constexpr int LDS = KG * 2 * (BM + BN) * 128 + 2 * BN * 4;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.group_m = 0;
    a.ks_n = pl.splits;
    a.ks_per = (a.nk + pl.splits - 1) / pl.splits;
    a.ks_phase = 1;
    static Vd3dLdsLimit lim;
    auto kern = conv_igemm_dma_kernel<T, BM, BN, WARPS_M, WARPS_N, PIPE, MS, RING, 0, false, 1, 0, KG>;
    if (const int rc = vd3d_raise_lds_limit((const void*)kern, LDS, lim, "hipFuncSetAttribute(conv_igemm split-K)")) return rc;
    const int64_t tiles = (int64_t)a.tiles_m * a.tiles_n;
    if (tiles <= 0 || tiles * pl.splits > 0x7fffffff) return VD3D_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * pl.splits)), dim3(NT), LDS, stream, a);
    const int64_t total = tiles * (BM * BN / 4);                         // f32x4 vectors of one split's partial image
    hipLaunchKernelGGL((splitk_reduce_kernel<T, BN, WARPS_M, WARPS_N, BM, MS>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, total);
    return vd3d_check_launch("conv_igemm split-K");
}

// Split-K on the 256 x 288 column strips (round 4): a layer whose strip tiles fill LESS THAN HALF the chip.  (i) config 2's 1408 -> 576 reg-tower
// output conv at batch 8 (60 x 2 = 120 tiles) ran on 128 x 288 tiles instead (240 tiles, one round, 38 % MFMA busy: 11 bytes of LDS fill per KFLOP
// against 6.7 on the strips); with K split in two the strips fill 240 CUs at their own efficiency and the 70 MB of fp32 partials cost less than
// the difference (246.7 -> 187.1 us).  (ii) the deep layers of a BATCH-1 call (1408 -> 1408 at 1 x 24 x 80: 8 x 5 = 40 strip tiles x 6 splits) -- before:
// 128 x 128 tiles x 3 splits.  Taken when the split strips fill >= 85 % of the CUs with >= 16 slices per split, <= 1/16 of the last strip's columns
// and <= 1/12 of the last tile's pixel rows wasted; 16-bit formats (see dispatch()); partials added in split order.
static SplitPlan plan_splitk_strip(const ConvArgs& a) {
    SplitPlan pl;
    const int cus = vd3d_device_cu_count() > 0 ? vd3d_device_cu_count() : 256;
    const int tm = (a.M + 255) / 256, tn = (a.Cout + 287) / 288;
    const int64_t tiles = (int64_t)tm * tn;
    const int pad_n = tn * 288 - a.Cout, pad_m = tm * 256 - a.M;
    if (a.Cout <= 288 || pad_n * 16 > a.Cout || pad_m * 12 > a.M) return pl;
    int sp;
    if (tiles * 2 > cus) {
        // Round 5, (iii): SEVERAL rounds, the last one badly filled, and a K deep enough that the fp32 partials are noise -- config 3's 2176 -> 576 reg-tower
        // output conv at 32 x 18 x 80: 180 x 2 = 360 strip tiles = 1.4 rounds (70 % of two); it ran on 128 x 288 tiles (720 = 2.81 rounds, but 1 001 TF/s: the
        // half-height tile's LDS fill per KFLOP).  Two splits of the strips are the same 720 workgroups at the strips' efficiency; 2 x 106 MB of partials
        // against a 306-slice K.  MEASURED: profiles/r05_strip_split_two_rounds.txt.
        auto fill = [&](int64_t w) { const int64_t r = (w + cus - 1) / cus; return (double)w / (double)(r * cus); };
        if (!(fill(tiles) < 0.8 && fill(2 * tiles) >= 0.9 && a.nk >= 160)) return pl;
        sp = 2;
    } else {
        if (x_fill_blind() && tiles * 4 >= cus) return pl;
        sp = (int)(cus / tiles);
        if (sp > a.nk / 16) sp = a.nk / 16;
        if (sp > 16) sp = 16;
        if (sp < 2 || tiles * sp * 100 < (int64_t)cus * 85) return pl;
    }
    const int per = (a.nk + sp - 1) / sp;
    pl.splits = (a.nk + per - 1) / per;               // every split is non-empty
    if (pl.splits < 2) { pl.splits = 1; return pl; }
    pl.bn = 288;
    pl.ws_bytes = (int64_t)pl.splits * tiles * 256 * 288 * 4;
    // the fp32 partials are a per-call allocation (and live once per captured graph): bounded -- a layer that would need more stays unsplit
    // (config 3's 2176 -> 576 conv, the largest shape the split was measured on, takes 212 MB)
    if (pl.ws_bytes > (256ll << 20)) return SplitPlan();
    return pl;
}

#ifdef VD3D_TUNING
// EXPERIMENT (round 6, tuning build only; forced ids 145 / 176 / 147): workgroups of two K groups (conv_igemm_dma_kernel, KG = 2) as a general replacement of the
// split over workgroups on 128-pixel tiles -- alone when the tiles fill at least half the chip, as phase 1 of a shallower split otherwise.  The micro-benchmark
// (tools/bench_conv.py, back-to-back launches on dense random operands) liked it everywhere: batch 8: 1408 -> 256 849 (two-way split-K) -> 961 TF/s, 256 -> 144
// 270 -> 397; one frame per call: 256 -> 144 63 -> 82, 288 -> 288 123 -> 145, 384 -> 384 191 -> 228, 1024 -> 256 307 -> 348, 512 -> 512 at 12 x 40 107 -> 136.  IN THE
// MODEL it loses or does nothing: per-launch trace of the headline step 1408 -> 256 96.1 + 8.9 (reduce) -> 127.1 us, 256 -> 144 24.0 + 8.5 -> 38.5 us; whole
// configurations, same box: C1 0.5016 / 0.5000 -> 0.5005 / 0.5036 ms per call, C2_B1_api 1.0929 / 1.0898 -> 1.0913 / 1.1006.  Only the 288 -> 288 neck convs
// (forced id 173) keep their gain per launch (43.3 / 46.9 -> 37.8 / 41.0 us) -- and the step does not move (3.461 / 3.460 against 3.461 / 3.461 ms): that part
// of the neck runs on the side stream under backbone layer3.  Nothing of this is in the product library.  profiles/r06_kgroups_ab.txt.
static SplitPlan plan_kgroups(const ConvArgs& a, int bn, bool forced = false) {
    SplitPlan pl;
    const int cus = vd3d_device_cu_count() > 0 ? vd3d_device_cu_count() : 256;
    const int64_t tiles = (int64_t)((a.M + 127) / 128) * ((a.Cout + bn - 1) / bn);
    pl.bn = bn;
    pl.kg = 2;
    int sp = (int)(cus / tiles);
    if (sp > a.nk / 8) sp = a.nk / 8;             // >= 4 slices per K group
    if (sp > 16) sp = 16;
    if (sp < 2) return pl;
    const int per = (a.nk + sp - 1) / sp;
    const int splits = (a.nk + per - 1) / per;    // every split is non-empty
    if (splits < 2) return pl;
    const int64_t ws = (int64_t)splits * tiles * 128 * bn * 4;
    // a slice of a 128-pixel tile costs ~0.7 us; the reduction launch ~7 us + its partials written and read back at ~3 TB/s
    const double saved_us = 0.7 * (a.nk - per), cost_us = 7.0 + 2.0 * (double)ws / 3.0e6;
    if (!forced && saved_us <= cost_us) return pl;
    pl.splits = splits;
    pl.ws_bytes = ws;
    return pl;
}
// (launches of a K-group plan: 16-bit formats)
template <typename T>
int dispatch_kgroups(ConvArgs& a, hipStream_t stream, const SplitPlan& pl) {
    if constexpr (sizeof(T) == 2) {
        const bool split = pl.splits >= 2 && a.ks_ws && a.ks_ws_bytes >= pl.ws_bytes;      // (no scratch handed over: K groups alone)
        if (pl.bn == 144) {
            if (split) return launch_splitk<T, 144, 4, 1, true, 128, 16, 0, 2>(a, stream, pl);
            return launch<T, 128, 144, 4, 1, true, true, 16, 0, 0, false, 0, 2>(a, stream);
        }
        if (split) return launch_splitk<T, 128, 2, 2, true, 128, 32, 0, 2>(a, stream, pl);
        return launch<T, 128, 128, 2, 2, true, true, 32, 0, 0, false, 0, 2>(a, stream);
    } else {
        vd3d_set_error("conv2d_igemm: K-group tiles are 16-bit only");
        return VD3D_EINVAL;
    }
}
#endif

template <typename T>
int dispatch_splitk(ConvArgs& a, hipStream_t stream, const SplitPlan& pl) {
    if constexpr (sizeof(T) == 2) {
        if (pl.bn == 288) return launch_splitk<T, 288, 4, 2, true, 256, 16, 6>(a, stream, pl);
    }
    if (pl.bn == 64) return launch_splitk<T, 64, 4, 1, false>(a, stream, pl);
    return launch_splitk<T, 128, 2, 2, true>(a, stream, pl);
}

// The 16x16x32 column strips with the STAGGERED DMA schedule (round 4; STG template parameter of conv_igemm_dma_kernel): waves 4-7 issue all
// ten pieces of slice t + 1 at fragment 8 (352 strip: 22 fragments per slice, barrier at 20) / 5 (288 strip with its ring of six: 18
// fragments, barrier at 12) of slice t while waves 0-3 keep the schedule "half right behind the barrier, half at the top of the next
// slice" -- SIMD mates no longer sit in their DMA bursts at the same time.  Same k order: bit-identical results.  Measured, same box,
// round-robin (tools/bench_conv.py): 1408 -> 1408 1315-1373 -> 1413 TF/s (STG 6 / 7 / 8 / 12: 1389-1401 / 1408-1412 / 1413-1425 / 1322-1357),
// 2176 -> 2176 on the 352 strip 1098-1109 -> 1168-1179; 1152 -> 1152 on the 288 strip 1352 -> 1363-1365 (STG 3 / 4 / 5 / 8: 1350 / 1354-1359 /
// 1363-1365 / 1304-1310).  Four phases (wave pairs at 0 / d / 2d / 3d) are 10-25 % SLOWER: a piece issued 8 fragments before the barrier does not
// land in time.  VD3D_CONV_NO_STAGGER=1: every wave on the old schedule (A/B).
template <typename T>
int launch_strip352(ConvArgs& a, hipStream_t stream) {
    if (vd3d_switch(VD3D_SW_CONV_NO_STAGGER) || a.group_m > 0) return launch<T, 256, 352, 4, 2, true, true, 16>(a, stream);
    return launch<T, 256, 352, 4, 2, true, true, 16, 0, 0, false, 8>(a, stream);
}
template <typename T>
int launch_strip288(ConvArgs& a, hipStream_t stream) {
    // (not for the grouped 1x1 GEMMs over an operand streamed from HBM -- config 3's 19 584-deep column GEMM: a later issue point has
    // HBM latency to cover, 3 344 against 3 206 us)
    if (vd3d_switch(VD3D_SW_CONV_NO_STAGGER) || a.group_m > 0) return launch<T, 256, 288, 4, 2, true, true, 16, 6>(a, stream);
    return launch<T, 256, 288, 4, 2, true, true, 16, 6, 0, false, 5>(a, stream);
}
// Round 5: a 320-wide strip for Cout = 2176 (config 3's R50 head: 2176 -> 2176 and the 19 584-deep DCN column GEMM).  On the 288 strip that width is
// 7.56 strips (94.4 % of the columns computed are real) x 180 pixel tiles = 1 440 workgroups = 5.6 rounds of 256 CUs (93.8 %): 88.5 % useful; on 320
// it is 6.8 strips (97.1 %) x 180 = 1 260 workgroups = 4.92 rounds (98.4 %): 95.6 %.  Same kernel: 4 x 2 waves of 64 x 160 on 16x16x32 MFMAs (TN = 10,
// 20 fragments per slice, five weight pieces per wave and slice exactly), 150 KB of LDS.
template <typename T>
int launch_strip320(ConvArgs& a, hipStream_t stream) {
    // MEASURED (tools/bench_conv.py, one box, round-robin; 288 strip -> this one): the column GEMM 19 584 -> 2176 at 32 x 18 x 80 1166 -> 1299 TF/s (+11 %),
    // 2176 -> 2176 3x3 at the same size 1366 -> 1391 (+1.8 %); at 16 x 18 x 80 (90 pixel tiles: 630 workgroups = 2.46 rounds) it LOSES, 1391 -> 1327 -- the
    // heuristic's round count keeps the 288 strip there.  Weight-fragment ring 2 / 4 / 5 / 10: 1391 / 1380 / 1377 / 692 (ten live fragments spill);
    // upper-half issue point 5 / 6 / 7 / 8: flat (1359 - 1377).  Unlike the 288 strip, the staggered schedule also wins on the grouped column GEMM
    // (1299 against 1226 without it).
    if (vd3d_switch(VD3D_SW_CONV_NO_STAGGER)) return launch<T, 256, 320, 4, 2, true, true, 16, 2>(a, stream);
    return launch<T, 256, 320, 4, 2, true, true, 16, 2, 0, false, 6>(a, stream);
}

// Tile override: 0 = heuristic, otherwise a config id (vd3d_test_force_conv_tile, csrc/test_hooks.h: a TEST hook, thread-local,
// not declared in include/vd3d.h).  The PRODUCT build only knows the ids of the
// tiles the heuristic below can pick (tests/test_conv_tiles_gpu.py forces each of them on awkward shapes and compares with
// the oracle); a forced tile that cannot run the given convolution is an error, never a silent fallback.  Experimental
// tiles and the timing ablations (results wrong by construction) exist only in a -DVD3D_TUNING build
// (`python -m visualdet3d_amd.build --tuning` -> libvd3d_hip_tuning.so, used by tools/bench_conv.py).
// Which of the resident-weight / streaming kernels takes a 16-bit shape under natural dispatch (ROUTE_NONE: the tile kernels, possibly
// split over K).  The ONE statement of that choice: dispatch() launches by it and vd3d_conv2d_workspace_bytes() sizes the split-K
// scratch by it, so the two cannot drift apart.
enum SpecialRoute { ROUTE_NONE = 0, ROUTE_REGW, ROUTE_KSPLIT, ROUTE_SMALL, ROUTE_NARROW, ROUTE_PW, ROUTE_RES64, ROUTE_TILE256x128 };
// Round 5: the Cin 256 3x3 layers with SEVERAL well-filled rounds of 256-pixel x 128-channel tiles (config 3's ResNet-50 layer3 at 64 x 18 x 80: 360 x 2 = 720
// tiles = 2.81 rounds of 256 CUs) run on THAT tile -- 8 waves of 64 x 64 on 16x16x32 MFMAs, weight-fragment ring 4, one workgroup per CU.
// MEASURED (profiles/r05_tile256x128_ab.txt).  tools/bench_conv.py, back-to-back launches: 64 x 18 x 80 966 against 798 TF/s for the K-split resident kernel;
// ResNet-34 layer3 at 16 x 24 x 80 (240 tiles, ONE round) 861 - 898 against 790 - 840 -- but inside the model the one-round case LOSES: same-box A/B of the
// headline 3.510 - 3.512 against 3.500 - 3.508 ms per step, and the serial trace shows why (41 - 45 us per launch against 38.5 - 41 for the K-split kernel:
// the micro-benchmark's dense random operands throttle the resident kernel more than the model's post-ReLU activations do); config 3, same box:
// 17.36 / 17.29 -> 17.27 / 17.20 ms per step.  Hence: only with >= 2 rounds.  (The 256 -> 256 cls conv at 8 x 24 x 80, 120 tiles: 504 against 712.)
static bool tile256x128_fills(const ConvArgs& a) {
    if (a.Cout % 128 != 0 || a.out_f32) return false;
    const int cus = vd3d_device_cu_count() > 0 ? vd3d_device_cu_count() : 256;
    const int64_t tm = (a.M + 255) / 256, tiles = tm * (a.Cout / 128), rounds = (tiles + cus - 1) / cus;
    return (rounds >= 2 || x_l3_tile()) && (int64_t)a.M * 100 >= tm * 256 * 95 && tiles * 100 >= rounds * cus * 90;
}
static SpecialRoute special_route(const ConvArgs& a) {
    if (a.Cin == 128 && regw_shape_ok(a)) return ROUTE_REGW;
    if (ksplit_shape_ok(a)) return tile256x128_fills(a) ? ROUTE_TILE256x128 : ROUTE_KSPLIT;
    if (small_shape_ok(a)) return ROUTE_SMALL;
    if (narrow_shape_ok(a) && !vd3d_switch(VD3D_SW_NO_NARROW)) return ROUTE_NARROW;
    if (pw_shape_ok(a)) return ROUTE_PW;
    if (a.Cin == 64 && a.Cout == 64 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.wide_store && !a.out_f32) return ROUTE_RES64;
    return ROUTE_NONE;
}
// The split-K plan of a 16-bit shape under natural dispatch (splits < 2: no split); shared by dispatch() and the workspace query.
static SplitPlan natural_split_plan(const ConvArgs& a) {
    const SplitPlan ps = vd3d_switch(VD3D_SW_NO_STRIP_SPLIT) ? SplitPlan() : plan_splitk_strip(a);
    if (ps.splits >= 2) return ps;
    return plan_splitk(a, false);
}

static thread_local int g_force_cfg = 0;      // per calling thread: a test forcing a tile cannot leak into other threads' launches

static int forced_tile_error(const char* why) {
    char msg[160];
    snprintf(msg, sizeof(msg), "conv2d_igemm: forced tile config %d %s", g_force_cfg, why);
    vd3d_set_error(msg);
    return VD3D_EINVAL;
}

template <typename T>
int dispatch(ConvArgs& a, hipStream_t stream) {
    constexpr bool kBf16 = sizeof(T) == 2;      // a 16-bit format (bf16 or fp16): the 16x16x32 tiles and the halo / resident kernels
    const bool halo_shape = a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.Ho == a.H && a.Wo == a.W &&
                            a.Cin % (128 / (int)sizeof(T)) == 0;
#define VD3D_BF16_ONLY(...) do { if constexpr (kBf16) return __VA_ARGS__; else return forced_tile_error("is a bf16-only tile"); } while (0)
#define VD3D_HALO_ONLY(...) do { if (halo_shape) return __VA_ARGS__; return forced_tile_error("needs 3x3 / stride 1 / pad 1 with Cin a multiple of the K slice"); } while (0)
    switch (g_force_cfg) {
        case 0: break;
        // ---- production tiles (every one of these is reachable from the heuristic below) ----
        case 44: return launch<T, 128, 128, 2, 2, true, true>(a, stream);
        case 42: return launch<T, 256, 256, 2, 4, true, true>(a, stream);
        case 40: return launch<T, 256, 352, 8, 1, true, true>(a, stream);
        case 41: return launch<T, 256, 288, 8, 1, true, true>(a, stream);
        case 50: VD3D_BF16_ONLY(launch_strip352<T>(a, stream));
        case 54: VD3D_BF16_ONLY(launch_strip288<T>(a, stream));
        case 52: VD3D_BF16_ONLY(launch_strip320<T>(a, stream));
        case 53: VD3D_BF16_ONLY(launch<T, 256, 128, 4, 2, true, true, 16, 4>(a, stream));
        case 12: return launch<T, 128, 352, 4, 1, true>(a, stream);
        case 11: return launch<T, 128, 288, 4, 1, true>(a, stream);
        case 76: VD3D_BF16_ONLY(launch<T, 128, 288, 4, 2, true, true, 16>(a, stream));
        case 43: return launch<T, 128, 192, 2, 2, true, true>(a, stream);
        case 79: VD3D_BF16_ONLY(launch<T, 256, 272, 8, 1, true, true, 16>(a, stream));
        case 73: VD3D_BF16_ONLY(launch<T, 128, 144, 4, 1, true, true, 16>(a, stream));
        case 87: return launch<T, 256, 32, 8, 1, true, true, 32, 1>(a, stream);
        case 30: return launch<T, 128, 64, 4, 1, true>(a, stream);
        case 21: VD3D_HALO_ONLY(launch_halo<T, 8, 32, 128, 4, 2, 4>(a, stream));
        case 23: VD3D_HALO_ONLY(launch_halo<T, 8, 16, 256, 2, 4, 3>(a, stream));
        case 27: VD3D_HALO_ONLY(launch_halo<T, 8, 16, 128, 2, 2, 4>(a, stream));
        case 60: break;      // heuristic, but without the resident-weight kernels (A/B of those kernels)
        case 144: case 130: {  // split-K over 128 x 128 / 128 x 64 tiles (the plan's split count, at least 2)
            const SplitPlan pl = plan_splitk(a, true);
            if (pl.splits < 2 || pl.bn != (g_force_cfg == 144 ? 128 : 64)) return forced_tile_error("split-K: needs >= 2 K slices and Cout <= 64 for the 128 x 64 tile / > 64 for 128 x 128");
            if (!a.ks_ws || a.ks_ws_bytes < pl.ws_bytes) return forced_tile_error("split-K: workspace missing or too small (vd3d_conv2d_workspace_bytes)");
            return dispatch_splitk<T>(a, stream, pl);
        }
        case 68:
            if constexpr (kBf16) {
                if (!ksplit_shape_ok(a)) return forced_tile_error("needs a 16-bit 3x3 / stride 1 / pad 1 conv with Cin 256, Cout a multiple of 64");
                return launch_ksplit(a, stream, std::is_same<T, hf16>::value ? VD3D_F16 : VD3D_BF16);
            } else return forced_tile_error("is a 16-bit-only tile");
        case 69:
            if constexpr (kBf16) {
                if (!small_shape_ok(a)) return forced_tile_error("needs a 16-bit 3x3 / stride 1 | 2 conv with Cin 16 | 32 | 64, Cout <= 32, no residual");
                return launch_small(a, stream, std::is_same<T, hf16>::value ? VD3D_F16 : VD3D_BF16);
            } else return forced_tile_error("is a 16-bit-only tile");
        case 70:
            if constexpr (kBf16) {
                if (!narrow_shape_ok(a)) return forced_tile_error("needs a 16-bit 3x3 / stride 1 / pad 1 conv with Cin a multiple of 64 (>= 128), Cout <= 32, weight_frag, no residual");
                return launch_narrow(a, stream, std::is_same<T, hf16>::value ? VD3D_F16 : VD3D_BF16);
            } else return forced_tile_error("is a 16-bit-only tile");
        case 58:
            if constexpr (kBf16) {
                if (!pw_shape_ok(a)) return forced_tile_error("needs a 16-bit 1x1 / stride 1 conv over a dense NHWC input with Cin 64 | 128, Cout a multiple of 256");
                return launch_pw(a, stream, std::is_same<T, hf16>::value ? VD3D_F16 : VD3D_BF16);
            } else return forced_tile_error("is a 16-bit-only tile");
        case 61:
            if constexpr (kBf16) {
                if (!regw_shape_ok(a)) return forced_tile_error("needs bf16 3x3 / stride 1 / pad 1, Cin 128 | 256, Cout a multiple of the channel slice");
                return launch_regw(a, stream, std::is_same<T, hf16>::value ? VD3D_F16 : VD3D_BF16);
            } else return forced_tile_error("is a bf16-only tile");
#ifdef VD3D_TUNING
        case 62: case 63: case 64: case 65: case 66: case 67:
            if constexpr (kBf16) {
                if (!regw_shape_ok(a) || a.Cin != 128) return forced_tile_error("regw experiments: Cin 128 only");
                switch (g_force_cfg) {
                    case 62: return launch_regw(a, stream, VD3D_BF16, 8, 0);
                    case 63: return launch_regw(a, stream, VD3D_BF16, 4, 1);
                    case 64: return launch_regw(a, stream, VD3D_BF16, 4, 2);
                    case 65: return launch_regw(a, stream, VD3D_BF16, 4, 3);
                    case 67: return launch_regw(a, stream, VD3D_BF16, 4, 9);
                    default: return launch_regw(a, stream, VD3D_BF16, 2, 0);
                }
            } else return forced_tile_error("is a bf16-only tile");
        // workgroups of two K groups (KG = 2): 128 x 144 / 128 x 128 tiles alone (173 / 145), and as phase 1 of a split over workgroups (176 / 147)
        case 173: VD3D_BF16_ONLY(launch<T, 128, 144, 4, 1, true, true, 16, 0, 0, false, 0, 2>(a, stream));
        case 145: case 176: case 147: {
            if constexpr (kBf16) {
                const int bn = g_force_cfg == 176 ? 144 : 128;
                SplitPlan pl = plan_kgroups(a, bn, true);
                if (g_force_cfg == 145) { pl.splits = 1; pl.ws_bytes = 0; }
                else {
                    if (pl.splits < 2) return forced_tile_error("K groups + split-K: needs >= 16 K slices and 128-pixel tiles on at most half the CUs");
                    if (!a.ks_ws || a.ks_ws_bytes < pl.ws_bytes) return forced_tile_error("split-K: workspace missing or too small (vd3d_conv2d_workspace_bytes)");
                }
                return dispatch_kgroups<T>(a, stream, pl);
            } else return forced_tile_error("is a 16-bit-only tile");
        }
        // two K groups with the staggered DMA issue of the second group (measured slower than the plain schedule, ids 173 / 145)
        case 174: VD3D_BF16_ONLY(launch<T, 128, 144, 4, 1, true, true, 16, 0, 0, false, 5, 2>(a, stream));
        case 175: VD3D_BF16_ONLY(launch<T, 128, 144, 4, 1, true, true, 16, 0, 0, false, 9, 2>(a, stream));
        case 146: VD3D_BF16_ONLY(launch<T, 128, 128, 2, 2, true, true, 32, 0, 0, false, 3, 2>(a, stream));
        case 1: return launch<T, 128, 128, 2, 2>(a, stream);
        case 2: return launch<T, 128, 128, 2, 2, true>(a, stream);
        case 8: return launch<T, 256, 256, 2, 4, true>(a, stream);
        case 9: return launch<T, 256, 352, 8, 1, true>(a, stream);
        case 10: return launch<T, 256, 288, 8, 1, true>(a, stream);
        case 17: return launch<T, 128, 192, 2, 2, true>(a, stream);
        case 51: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16>(a, stream));
        case 47: VD3D_HALO_ONLY(launch_halo<T, 8, 16, 128, 2, 2, 2>(a, stream));
        case 77: VD3D_HALO_ONLY(launch_halo<T, 8, 16, 256, 2, 2, 3>(a, stream));
        case 84: return launch<T, 256, 32, 4, 1, true, true>(a, stream);
        case 85: return launch<T, 128, 64, 4, 1, true, true, 32, 2>(a, stream);
        case 86: return launch<T, 128, 64, 2, 2, true, true>(a, stream);
        case 78: VD3D_HALO_ONLY(launch_halo<T, 8, 16, 256, 1, 4, 3>(a, stream));
        // timing ablations of the 352 strip: WRONG results by construction (ABL template flag)
        case 91: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 1>(a, stream));
        case 92: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 2>(a, stream));
        case 93: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 3>(a, stream));
        case 94: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 4>(a, stream));
        case 95: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 5>(a, stream));
        case 96: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 6>(a, stream));
        case 97: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 7>(a, stream));
        // staggered DMA issue of the two waves of a SIMD (STG = the fragment at which the upper half issues): sweep variants
        case 31: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 0, false, 6>(a, stream));
        case 55: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 0, false, 7>(a, stream));
        case 56: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 0, false, 8>(a, stream));
        case 32: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 0, false, 104>(a, stream));
        case 36: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 2, 0, false, 6>(a, stream));      // 288 strip, ring 2 (barrier at 16 of 18)
        case 37: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 2, 0, false, 8>(a, stream));
        case 57: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 3, 0, false, 6>(a, stream));      // ring 3 (barrier at 15)
        case 34: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6, 0, false, 6>(a, stream));      // ring 6 (production: STG 5)
        case 33: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6, 0, false, 5>(a, stream));
        case 38: VD3D_BF16_ONLY(launch<T, 128, 288, 4, 2, true, true, 16, 0, 0, false, 6>(a, stream));      // 128 x 288 (1408 -> 576): 18 fragments, barrier at 16
        case 39: VD3D_BF16_ONLY(launch<T, 128, 288, 4, 2, true, true, 16, 0, 0, false, 8>(a, stream));
        // config 3's column GEMM with the blend's VALU load injected into the 288 strip (ring 6, one DMA schedule: what the grouped GEMM runs)
        case 111: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6, 29>(a, stream));   // 9 per fragment (~160 per slice and wave: an fp16 blend)
        case 112: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6, 37>(a, stream));   // 17 (~304: the bf16 blend incl. unpack / pack)
        case 113: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6, 54>(a, stream));   // 34
        // 256 pixels x 128 channels, one workgroup per CU (layer 3 at batch 8: 120 x 2 = 240 tiles for 256 CUs): does a third tile shape move the layer-3 plateau?
        case 150: VD3D_BF16_ONLY(launch<T, 256, 128, 4, 2, true, true, 16>(a, stream));
        case 151: return launch<T, 256, 128, 4, 2, true, true>(a, stream);
        case 152: return launch<T, 256, 128, 8, 1, true, true>(a, stream);
        case 153: VD3D_BF16_ONLY(launch<T, 256, 128, 4, 2, true, true, 16, 4>(a, stream));      // (production: forced id 53)
        case 154: VD3D_BF16_ONLY(launch<T, 256, 128, 4, 2, true, true, 16, 4, 0, false, 2>(a, stream));   // + staggered DMA issue (upper half at fragment 2 of 8)
        case 155: VD3D_BF16_ONLY(launch<T, 256, 128, 4, 2, true, true, 16, 2, 0, false, 3>(a, stream));
        case 156: VD3D_BF16_ONLY(launch<T, 256, 128, 2, 4, true, true, 16>(a, stream));                   // 2 x 4 waves of 128 x 32
        // the 320 strip: ring / stagger sweep (production: ring 2, STG 6 = case 124)
        case 120: VD3D_BF16_ONLY(launch<T, 256, 320, 4, 2, true, true, 16, 5, 0, false, 5>(a, stream));
        case 121: VD3D_BF16_ONLY(launch<T, 256, 320, 4, 2, true, true, 16, 5, 0, false, 7>(a, stream));
        case 122: VD3D_BF16_ONLY(launch<T, 256, 320, 4, 2, true, true, 16, 5, 0, false, 8>(a, stream));
        case 123: VD3D_BF16_ONLY(launch<T, 256, 320, 4, 2, true, true, 16, 10, 0, false, 6>(a, stream));
        case 124: VD3D_BF16_ONLY(launch<T, 256, 320, 4, 2, true, true, 16, 2, 0, false, 6>(a, stream));
        case 125: VD3D_BF16_ONLY(launch<T, 256, 320, 4, 2, true, true, 16, 4, 0, false, 6>(a, stream));
        case 126: VD3D_BF16_ONLY(launch<T, 256, 320, 4, 2, true, true, 16, 2>(a, stream));
        case 101: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6>(a, stream));       // the same instantiation without injection
        case 102: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6, 20>(a, stream));   // empty volatile asm per fragment
        case 103: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6, 21>(a, stream));   // 1 VALU per fragment
        case 104: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6, 23>(a, stream));   // 3
        case 105: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6, 26>(a, stream));   // 6
        case 106: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6, 32>(a, stream));   // 12
        case 107: VD3D_BF16_ONLY(launch<T, 256, 288, 4, 2, true, true, 16, 6, 29, false, 5>(a, stream));   // the production 288 strip (STG 5) + 9 per fragment
        case 108: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 29, false, 8>(a, stream));   // the production 352 strip (STG 8) + 9 per fragment
        case 109: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 26, false, 8>(a, stream));   // ... + 6
        case 110: VD3D_BF16_ONLY(launch<T, 256, 352, 4, 2, true, true, 16, 0, 20, false, 8>(a, stream));   // ... + empty asm
        case 74: VD3D_BF16_ONLY(launch<T, 64, 144, 2, 1, true, true, 16>(a, stream));
        case 75: VD3D_BF16_ONLY(launch<T, 128, 288, 2, 2, true, true, 16>(a, stream));
        case 71: VD3D_HALO_ONLY(launch_halo<T, 8, 16, 128, 4, 1, 2>(a, stream));
        case 72: VD3D_HALO_ONLY(launch_halo<T, 8, 16, 64, 2, 2, 4>(a, stream));
        case 28: VD3D_HALO_ONLY(launch_halo<T, 8, 32, 64, 4, 2, 4>(a, stream));
#endif
        default: return forced_tile_error("is not a tile of this build");
    }
    if constexpr (std::is_same<T, short>::value || std::is_same<T, hf16>::value) {
        constexpr int fmt = std::is_same<T, hf16>::value ? VD3D_F16 : VD3D_BF16;
        // the resident-weight / streaming kernels (ONE predicate, `special_route`, shared with vd3d_conv2d_workspace_bytes: a shape one
        // of these takes never splits over K)
        if (g_force_cfg != 60) switch (special_route(a)) {
            // register-resident weights (v6), Cin 128: +25 % over the 8x32x128 halo tile on ResNet layer2
            case ROUTE_REGW: return launch_regw(a, stream, fmt);
            // Cin 256: the 8-wave K-split resident kernel (v7): +3.5 % over the 8x16x256 halo tiles on layer3 (781 vs 754 TF/s), +50 % on
            // the 256 -> 256 cls conv whose 120 halo tiles leave half the chip idle (690 vs 441; the 4-wave v6 kernel: 641 - 704)
            case ROUTE_KSPLIT: return launch_ksplit(a, stream, fmt);
            case ROUTE_TILE256x128:
                if constexpr (kBf16) return launch<T, 256, 128, 4, 2, true, true, 16, 4>(a, stream);
                else break;
            // small-channel streaming kernel (DLA level 0 / 1, DCN offset convs): input staged once, HBM-bound instead of LDS-fill-bound
            case ROUTE_SMALL: return launch_small(a, stream, fmt);
            // the same tile walked over 64-channel chunks for the deep offset convs (Cin 128 ... 2176 -> 27): 2-4x over the 256 x 32 tiles
            case ROUTE_NARROW: return launch_narrow(a, stream, fmt);
            // point-wise expansions with one or two K slices (ResNet-50 conv3 / down-sample of the first two stages): streaming kernel
            case ROUTE_PW: return launch_pw(a, stream, fmt);
            case ROUTE_RES64: return launch_resident64(a, stream, fmt);
            default: break;
        }
    }
    // low-parallelism shapes with a deep K (batch-1 calls: 1408 -> 1408 at 24 x 80 is 165 tiles for 512 workgroup slots): split-K.
    // 16-bit formats only: the split changes the fp32 summation order with the tile count, i.e. with the batch size -- fp32 is the
    // validation mode, whose results must not depend on how many frames share a call (tests: batch-1 == slice of the batched call)
    if (g_force_cfg == 0 && a.ks_ws && sizeof(T) == 2) {
        const SplitPlan pl = natural_split_plan(a);
        if (pl.splits >= 2 && a.ks_ws_bytes >= pl.ws_bytes) return dispatch_splitk<T>(a, stream, pl);
    }
    // Cout <= 32: 8 waves of 32 pixels x 32 channels, pipelined loop (+8 % on the ghost 24 -> 24 conv, +27 % on KM3D's 64 -> 27
    // offset convs over the 4-wave barrier-per-slice version)
    if (a.Cout <= 32) return launch<T, 256, 32, 8, 1, true, true, 32, 1>(a, stream);
    if (a.Cout <= 64) return launch<T, 128, 64, 4, 1, true>(a, stream);   // 48 KiB LDS -> 3 workgroups / CU (short K: latency bound)
    // 3x3 / stride 1 / pad 1 with 64-channel-aligned input: the halo kernel (each input pixel staged once per channel
    // chunk instead of once per tap) wins on the mid-size layers (measured on MI355X, bf16):
    //   Cout <= 128 (layer2-like): 8x32 patch x 128 channels          +6 %
    //   Cout == 256, short K (layer3-like): 8x16 patch x 256 channels +10 %
    //   Cout == 256, long K (1408 -> 256 cls conv): 8x16 patch x 128  +60 %
    // ragged image edges are fine (halo rows beyond the image are zero-filled, pixels beyond it are dropped) as long as the
    // padded patch grid wastes < 8 % of the work
    auto patch_waste_ok = [&](int th, int tw) {
        const int64_t padded = (int64_t)((a.H + th - 1) / th * th) * ((a.W + tw - 1) / tw * tw);
        return padded * 100 <= (int64_t)a.H * a.W * 108;
    };
    const bool halo_ok = halo_shape && patch_waste_ok(8, 16);
    if (halo_ok && a.Cout <= 128 && a.Cout % 128 == 0 && patch_waste_ok(8, 32)) return launch_halo<T, 8, 32, 128, 4, 2, 4>(a, stream);
    if (halo_ok && a.Cout == 256) {
        if (a.Cin >= 1024) return launch_halo<T, 8, 16, 128, 2, 2, 4>(a, stream);
        return launch_halo<T, 8, 16, 256, 2, 4, 3>(a, stream);
    }
    // Tile choice by modelled throughput = full-occupancy rate of the tile shape (measured on MI355X, bf16 TFLOP/s)
    // x tile-edge waste x last-round occupancy.  Candidates: 128x128 (2 workgroups/CU), 256x256 (8 waves of 128x64,
    // 1/CU), and "column strips" 256xBN / 128xBN with BN = 288 | 352 (= 1152/4, 1408/4: every wave owns 32 pixels x
    // the whole strip) which put the 1152- and 1408-channel layers at B = 8 on the chip in ONE ~94 %-full round.
    // workgroup slots of one round: one (256-row tiles, > 80 KiB of LDS) or two (128-row tiles) workgroups per CU of THIS device
    const int cus = vd3d_device_cu_count() > 0 ? vd3d_device_cu_count() : 256;
    const int kSlots1 = cus, kSlots2 = 2 * cus;
    auto util = [&](int bm, int bn, int slots) {
        const double tm = (a.M + bm - 1) / bm, tn = (a.Cout + bn - 1) / bn;
        const double tiles = tm * tn;
        const double rounds = (double)((int64_t)((tiles + slots - 1) / slots));
        if (x_fill_blind() && rounds == 1.0 && tiles * 4 >= slots) return ((double)a.M * a.Cout) / (tm * bm * tn * bn);
        return ((double)a.M * a.Cout) / (tm * bm * tn * bn) * tiles / (rounds * slots);
    };
    double best = 1000.0 * util(128, 128, kSlots2);
    int pick = 0;
    const double r256 = 1450.0 * util(256, 256, kSlots1);
    if (r256 > best) { best = r256; pick = 1; }
    if (a.Cout % 352 == 0) {
        const double r = 1470.0 * util(256, 352, kSlots1);
        if (r > best) { best = r; pick = 2; }
        const double r2 = 900.0 * util(128, 352, kSlots1);
        if (r2 > best) { best = r2; pick = 4; }
    }
    if (sizeof(T) == 2 && a.Cout % 272 == 0) {   // 2176 = 8 x 272 (Stereo3D R50 head): 8 waves of 32 x 272 on 16x16x32 MFMAs, +20 %
        const double r = 1320.0 * util(256, 272, kSlots1);
        if (r > best) { best = r; pick = 7; }
    }
    if (a.Cout % 192 == 0) {     // 80 KiB LDS: two workgroups per CU
        const double r = 1390.0 * util(128, 192, kSlots2);
        if (r > best) { best = r; pick = 6; }
    }
    if (a.Cout % 288 == 0 || (sizeof(T) == 2 && a.Cout > 1152)) {
        // (a partial last strip is fine: 2176 = 7.56 x 288 runs 1298 TF/s on the 288 strips against 1246 on 8 x 272 -- the 4 x 2-wave
        //  16x16x32 layout reads fewer fragment bytes per MFMA than the 8 x 1 layout of the 272 strip)
        const double r = 1455.0 * util(256, 288, kSlots1);
        if (r > best) { best = r; pick = 3; }
    }
    if (sizeof(T) == 2 && a.Cout > 1152 && a.Cout % 352 != 0 && a.Cout % 288 != 0) {
        // 2176 = 6.8 x 320: 95.6 % of the chip-rounds useful against 88.5 % on the 288 strips (launch_strip320)
        const double r = 1460.0 * util(256, 320, kSlots1);
        if (r > best) { best = r; pick = 9; }
    }
    if (a.Cout % 288 == 0) {
        // bf16: 8 waves of 32 x 144 on 16x16x32 MFMAs: 1109 vs 929 TF/s (128x192 tiles) on the 1408 -> 576 reg output conv
        // (only with >= 2 strips per pixel tile: on the 288 -> 288 neck convs the 120 tiles leave half the chip idle, 69 vs 55 us)
        const double r2 = (sizeof(T) == 2 && a.Cout >= 576 ? 1200.0 : 850.0) * util(128, 288, kSlots1);
        if (r2 > best) { best = r2; pick = 5; }
        if (sizeof(T) == 2 && a.Cout == 288) {     // 128 x 144 tiles on 16x16x32 MFMAs, two workgroups per CU: 544 vs 484 TF/s on 288 -> 288
            const double r3 = 1160.0 * util(128, 144, kSlots2);
            if (r3 > best) { best = r3; pick = 8; }
        }
    }
    if (vd3d_switch(VD3D_SW_CONV_DEBUG)) fprintf(stderr, "[vd3d conv] M=%d N=%d pick=%d best=%.0f\n", a.M, a.Cout, pick, best);
    switch (pick) {
        // the software-pipelined main loop (PIPE) measured +2.5 % (256x352) ... +17 % (256x288 on Cout 576) over the
        // barrier-per-slice loop on every shape of the hot path
        case 1: return launch<T, 256, 256, 2, 4, true, true>(a, stream);
        // strips: bf16 uses 16x16x32 MFMAs so that a wave owns 64 pixels x half the strip (0.68 KB of fragment reads per
        // 32x32x16-equivalent instead of 1.09 for 32 x the whole strip): +11 % (352) / +8 % (288) measured
        case 2:
            if constexpr (sizeof(T) == 2) return launch_strip352<T>(a, stream);
            else return launch<T, 256, 352, 8, 1, true, true>(a, stream);
        case 3:
            if constexpr (sizeof(T) == 2) return launch_strip288<T>(a, stream);   // ring of 6: +4 % over 2
            else return launch<T, 256, 288, 8, 1, true, true>(a, stream);
        case 9:
            if constexpr (sizeof(T) == 2) return launch_strip320<T>(a, stream);
            else return launch<T, 256, 256, 2, 4, true, true>(a, stream);
        case 4: return launch<T, 128, 352, 4, 1, true>(a, stream);
        case 5:     // bf16: 8 waves of 32 x 144 on 16x16x32 MFMAs (+20 % over 128x192 tiles on the 1408 -> 576 reg output conv)
            if constexpr (sizeof(T) == 2) return launch<T, 128, 288, 4, 2, true, true, 16>(a, stream);
            else return launch<T, 128, 288, 4, 1, true>(a, stream);
        case 6: return launch<T, 128, 192, 2, 2, true, true>(a, stream);
        case 7:
            if constexpr (sizeof(T) == 2) return launch<T, 256, 272, 8, 1, true, true, 16>(a, stream);
            else return launch<T, 256, 256, 2, 4, true, true>(a, stream);
        case 8:
            if constexpr (sizeof(T) == 2) return launch<T, 128, 144, 4, 1, true, true, 16>(a, stream);
            else return launch<T, 128, 128, 2, 2, true, true>(a, stream);
        default: return launch<T, 128, 128, 2, 2, true, true>(a, stream);
    }
#undef VD3D_BF16_ONLY
#undef VD3D_HALO_ONLY
}

}  // namespace

extern "C" int vd3d_test_force_conv_tile(int cfg) {
    g_force_cfg = cfg;
    return VD3D_OK;
}


extern "C" int vd3d_conv2d_production_tiles(int32_t* ids, int cap) {
    // keep in step with the "production tiles" block of dispatch()
    static const int32_t kIds[] = {44, 42, 40, 41, 50, 54, 52, 53, 12, 11, 76, 43, 79, 73, 87, 30, 21, 23, 27, 61, 68, 69, 58, 70, 144, 130};
    const int n = (int)(sizeof(kIds) / sizeof(kIds[0]));
    for (int i = 0; i < n && i < cap; ++i) ids[i] = kIds[i];
    return n;
}

static int fill_conv_args(const vd3d_conv_params* p, ConvArgs& a) {
    if (!p || !p->in || !p->weight || !p->out) { vd3d_set_error("conv2d_igemm: null pointer"); return VD3D_EINVAL; }
    const int es = (p->dtype == VD3D_BF16 || p->dtype == VD3D_F16) ? 2 : (p->dtype == VD3D_F32 ? 4 : 0);
    if (!es) { vd3d_set_error("conv2d_igemm: bad dtype"); return VD3D_EINVAL; }
    const int ve = 16 / es, bke = 128 / es;
    // every 16-byte vector must start on a 16-byte boundary: normally the pixel stride is a vector multiple; the
    // stem (NHWC4, pad 0, kw 1) only needs stride * pix_stride to be one
    const bool pix_ok = (p->in_pix_stride % ve == 0) ||
                        (p->pad == 0 && p->kw == 1 && (p->stride * p->in_pix_stride) % ve == 0);
    if (p->Cin % ve || !pix_ok || p->in_row_stride % ve || p->in_batch_stride % ve ||
        ((uintptr_t)p->in & 15) || ((uintptr_t)p->weight & 15)) {
        vd3d_set_error("conv2d_igemm: input channels / strides / pointers must be 16-byte aligned");
        return VD3D_EINVAL;
    }
    if (p->Kpad % bke || p->Kpad < p->kh * p->kw * p->Cin || p->CoutPad % 128 || p->CoutPad < p->Cout) {
        vd3d_set_error("conv2d_igemm: packed weight padding does not match");
        return VD3D_EINVAL;
    }
    if (p->in_bytes <= 0 || p->in_bytes > 0x7ffffff0ll || (int64_t)p->CoutPad * p->Kpad * es > 0x7ffffff0ll) {
        vd3d_set_error("conv2d_igemm: tensor exceeds 2 GiB (32-bit buffer offsets); split the batch");
        return VD3D_ERANGE;
    }
    a.in = (const char*)p->in; a.weight = (const char*)p->weight; a.scale = p->scale; a.shift = p->shift;
    a.wfrag = (const char*)p->weight_frag;
    if (p->weight_frag && ((uintptr_t)p->weight_frag & 15)) { vd3d_set_error("conv2d_igemm: weight_frag must be 16-byte aligned"); return VD3D_EINVAL; }
    a.residual = (const char*)p->residual; a.out = (char*)p->out;
    a.ks_ws = (float*)p->splitk_ws; a.ks_ws_bytes = p->splitk_ws ? p->splitk_ws_bytes : 0;
    if (p->splitk_ws && ((uintptr_t)p->splitk_ws & 15)) { vd3d_set_error("conv2d_igemm: splitk_ws must be 16-byte aligned"); return VD3D_EINVAL; }
    a.B = p->B; a.H = p->H; a.W = p->W; a.Cin = p->Cin;
    a.in_pix_stride = p->in_pix_stride; a.in_row_stride = p->in_row_stride; a.in_batch_stride = p->in_batch_stride;
    a.in_bytes = (uint32_t)p->in_bytes; a.w_bytes = (uint32_t)((int64_t)p->CoutPad * p->Kpad * es);
    a.Ho = p->Ho; a.Wo = p->Wo; a.Cout = p->Cout;
    a.out_pix_stride = p->out_pix_stride; a.res_pix_stride = p->res_pix_stride;
    a.kh = p->kh; a.kw = p->kw; a.stride = p->stride; a.pad = p->pad; a.dil = p->dil;
    a.Kpad = p->Kpad; a.relu = p->relu; a.out_f32 = p->out_f32;
    const int64_t M = (int64_t)p->B * p->Ho * p->Wo;
    if (M <= 0 || M > 0x7fffffff) return VD3D_ERANGE;
    a.M = (int)M;
    a.fd_howo = make_fastdiv((uint32_t)(a.Ho * a.Wo));
    a.fd_wo = make_fastdiv((uint32_t)a.Wo);
    a.ntaps = p->kh * p->kw;
    a.nk = (p->kh * p->kw * p->Cin + bke - 1) / bke;
    const int oes = (es == 4 || p->out_f32) ? 4 : 2;
    // vector epilogue needs 4-channel groups fully inside Cout and 4-element aligned rows
    a.vec_epilogue = (p->Cout % 4 == 0) && (p->out_pix_stride % 4 == 0) && (((uintptr_t)p->out % (4 * oes)) == 0) &&
                     (!p->residual || (p->res_pix_stride % 4 == 0 && ((uintptr_t)p->residual % (4 * es)) == 0)) &&
                     (!p->scale || ((uintptr_t)p->scale % 16) == 0) && (!p->shift || ((uintptr_t)p->shift % 16) == 0);
    // 16-byte bf16 stores (half-wave pairing) need 16-channel groups inside Cout and 16-byte aligned rows
    a.wide_store = a.vec_epilogue && oes == 2 && (p->Cout % 16 == 0) && (p->out_pix_stride % 8 == 0) && (((uintptr_t)p->out & 15) == 0);
    a.chunk_major = (a.ntaps > 1) && (p->Cin % bke == 0);
    // grouped tile order for 1x1 GEMMs whose pixel matrix exceeds the on-die caches (> 512 MB)
    a.group_m = (a.ntaps == 1 && ((int64_t)a.M * p->Cin * es > (512ll << 20) || vd3d_switch(VD3D_SW_FORCE_GROUP_M)) && !vd3d_switch(VD3D_SW_NO_GROUP_M)) ? 4 : 0;
    // whole-line stores through LDS (conv_epilogue_lines): short-K layers (the output stream is a large share of their bytes) and
    // long-K layers alike (their few rounds of tiles end as synchronised store bursts)
    // strips (16x16x32 tiles): whole-line stores through LDS whenever the 16-bit output rows are 16-byte aligned (the wave's strip must
    // also lie inside Cout: checked per wave)
    a.strip_lines = a.wide_store && !p->out_f32 && !vd3d_switch(VD3D_SW_NO_LINE_STORE);
    a.line_store = a.wide_store && !p->out_f32 && p->Cout % 64 == 0 && (p->out_pix_stride % 64 == 0) && (((uintptr_t)p->out & 127) == 0) &&
                   !vd3d_switch(VD3D_SW_NO_LINE_STORE);
    return VD3D_OK;
}

extern "C" int vd3d_conv2d_igemm(const vd3d_conv_params* p, void* stream) {
    ConvArgs a;
    const int rc = fill_conv_args(p, a);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    return p->dtype == VD3D_BF16 ? dispatch<short>(a, s) : (p->dtype == VD3D_F16 ? dispatch<hf16>(a, s) : dispatch<float>(a, s));
}

extern "C" int64_t vd3d_conv2d_workspace_bytes(const vd3d_conv_params* p) {
    ConvArgs a;
    vd3d_conv_params q = *p;
    q.splitk_ws = nullptr;
    if (fill_conv_args(&q, a)) return -1;
    if (g_force_cfg == 144 || g_force_cfg == 130) return plan_splitk(a, true).ws_bytes;
#ifdef VD3D_TUNING
    if (g_force_cfg == 147 || g_force_cfg == 176) return plan_kgroups(a, g_force_cfg == 147 ? 128 : 144, true).ws_bytes;
#endif
    if (g_force_cfg != 0) return 0;
    // fp32 never splits under natural dispatch (see dispatch()); a shape one of the resident-weight / streaming kernels takes never splits
    if (p->dtype == VD3D_F32 || special_route(a) != ROUTE_NONE) return 0;
    const SplitPlan pl = natural_split_plan(a);
    return pl.splits >= 2 ? pl.ws_bytes : 0;
}

extern "C" int vd3d_conv2d_pair(const vd3d_conv_params* pa, const vd3d_conv_params* pb, void* stream) {
    if (!pa || !pb) { vd3d_set_error("conv2d_pair: null pointer"); return VD3D_EINVAL; }
    if ((pa->dtype != VD3D_BF16 && pa->dtype != VD3D_F16) || pb->dtype != pa->dtype) { vd3d_set_error("conv2d_pair: both convs in the same 16-bit format"); return VD3D_EINVAL; }
    vd3d_conv_params qa = *pa, qb = *pb;
    qa.out = pb->out;                              // (unused: keeps the generic checks of the parameter block happy)
    qa.out_pix_stride = pa->Cout;
    qb.in = pa->in;
    qb.in_pix_stride = pb->Cin; qb.in_row_stride = pb->W * pb->Cin; qb.in_batch_stride = (int64_t)pb->H * pb->W * pb->Cin;
    qb.in_bytes = pa->in_bytes;
    ConvArgs a, b;
    int rc = fill_conv_args(&qa, a);
    if (rc) return rc;
    rc = fill_conv_args(&qb, b);
    if (rc) return rc;
    if (!pair_shape_ok(a, b)) {
        vd3d_set_error("conv2d_pair: needs conv A 3x3/s1/p1 16 -> 16 and conv B 3x3/s2/p1 16 -> 16 | 32 continuing A's geometry, 16-bit, no residual");
        return VD3D_EINVAL;
    }
    return launch_pair(a, b, (hipStream_t)stream, pa->dtype);
}

extern "C" int vd3d_conv2d_bottleneck(const vd3d_conv_params* p1, const vd3d_conv_params* p2, const vd3d_conv_params* p3, const vd3d_conv_params* pd,
                                      void* stream) {
    if (!p1 || !p2 || !p3) { vd3d_set_error("conv2d_bottleneck: null pointer"); return VD3D_EINVAL; }
    if ((p1->dtype != VD3D_BF16 && p1->dtype != VD3D_F16) || p2->dtype != p1->dtype || p3->dtype != p1->dtype || (pd && pd->dtype != p1->dtype)) {
        vd3d_set_error("conv2d_bottleneck: every conv in the same 16-bit format");
        return VD3D_EINVAL;
    }
    // the intermediates exist in LDS only: conv1.out / conv2.in / conv2.out / conv3.in (and the downsample conv's out) are ignored
    vd3d_conv_params q1 = *p1, q2 = *p2, q3 = *p3, qd = pd ? *pd : *p3;
    const int64_t mid_bytes = (int64_t)p2->B * p2->H * p2->W * p2->Cin * 2;
    q1.out = p3->out; q1.out_pix_stride = p1->Cout;
    q2.in = p1->in; q2.in_pix_stride = p2->Cin; q2.in_row_stride = p2->W * p2->Cin; q2.in_batch_stride = (int64_t)p2->H * p2->W * p2->Cin; q2.in_bytes = mid_bytes;
    q2.out = p3->out; q2.out_pix_stride = p2->Cout;
    q3.in = p1->in; q3.in_pix_stride = p3->Cin; q3.in_row_stride = p3->W * p3->Cin; q3.in_batch_stride = (int64_t)p3->H * p3->W * p3->Cin; q3.in_bytes = mid_bytes;
    if (pd) { qd.out = p3->out; qd.out_pix_stride = pd->Cout; }
    ConvArgs a1, a2, a3, ad;
    if (const int rc = fill_conv_args(&q1, a1)) return rc;
    if (const int rc = fill_conv_args(&q2, a2)) return rc;
    if (const int rc = fill_conv_args(&q3, a3)) return rc;
    if (pd) { if (const int rc = fill_conv_args(&qd, ad)) return rc; }
    if (!bottleneck_shape_ok(a1, a2, a3, pd ? &ad : nullptr)) {
        vd3d_set_error("conv2d_bottleneck: needs conv1 1x1 256 | 64 -> 64 + relu, conv2 3x3/s1/p1 64 -> 64 + relu, conv3 1x1 64 -> 256 with residual = conv1's "
                       "(dense) input, or with the 1x1 64 -> 256 downsample conv of the same input; 16-bit");
        return VD3D_EINVAL;
    }
    return launch_bottleneck(a1, a2, a3, pd ? &ad : nullptr, (hipStream_t)stream, p1->dtype);
}

extern "C" int vd3d_km3d_head_fused(const vd3d_conv_params* p, const void* w2_packed, const float* b2, void* const* outs,
                                    const int32_t* n_out, int n_heads, void* stream) {
    if (!p || !w2_packed || !b2 || !outs || !n_out) { vd3d_set_error("km3d_head_fused: null pointer"); return VD3D_EINVAL; }
    if (n_heads < 1 || n_heads > 9 || (p->dtype != VD3D_BF16 && p->dtype != VD3D_F16) || p->Cout != 256 * n_heads || p->kh != 3 || p->kw != 3 || p->stride != 1 ||
        p->pad != 1 || p->dil != 1 || p->Cin % 64 || !p->shift || p->scale || p->residual || ((uintptr_t)w2_packed & 15)) {
        vd3d_set_error("km3d_head_fused: needs bf16, 3x3/s1/p1, Cin % 64 == 0, Cout = 256 x heads (<= 9), bias only");
        return VD3D_EINVAL;
    }
    ConvArgs a;
    vd3d_conv_params q = *p;
    q.out = (void*)outs[0];                       // unused by the fused epilogue; keeps the generic checks happy
    const int rc = fill_conv_args(&q, a);
    if (rc) return rc;
    if (!a.chunk_major) { vd3d_set_error("km3d_head_fused: channel count must be chunk aligned"); return VD3D_EINVAL; }
    a.h_w2 = (const char*)w2_packed;
    a.h_b2 = b2;
    for (int h = 0; h < n_heads; ++h) {
        if (!outs[h] || n_out[h] < 1 || n_out[h] > 32) { vd3d_set_error("km3d_head_fused: 1 <= n_out <= 32 per head"); return VD3D_EINVAL; }
        a.h_out[h] = (float*)outs[h];
        a.h_n[h] = n_out[h];
    }
    if (vd3d_switch(VD3D_SW_HEAD_PARKED)) {        // the first version: tile parked in LDS, second GEMM from there (A/B)
        if (p->dtype == VD3D_F16) return launch<hf16, 256, 256, 2, 4, true, true, 32, 0, 0, true>(a, (hipStream_t)stream);
        return launch<short, 256, 256, 2, 4, true, true, 32, 0, 0, true>(a, (hipStream_t)stream);
    }
    if (!km3d_head_shape_ok(a)) { vd3d_set_error("km3d_head_fused: shape not supported by the persistent head kernel"); return VD3D_EINVAL; }
    return launch_km3d_head(a, (hipStream_t)stream, p->dtype);
}
