// conv_strip1w.hip -- EXPERIMENT (tile ids 52 / 53, forced through the test hook only): the 256 x 352 / 256 x 288 strips of
// conv_igemm_dma_kernel<.., 4, 2, PIPE, 16> with ONE wave per SIMD (gfx950).
//
// The production strips run 4 x 2 waves of 64 x 176 (two per SIMD, 255 registers each): per 32-deep sub-step a wave reads 15 fragments
// for 44 MFMAs and keeps a ring of only two weight fragments (no third fits), so every weight fragment is consumed 64 cycles after its
// ds_read was issued and the MFMA pipe sits at 57-60 %.  Here a workgroup is 2 x 2 waves of 128 x 176 with the 512-register
// budget of a lone wave: 352 accumulators, pixel fragments double buffered per sub-step (64), a ring of two weight fragments whose
// look-ahead is now EIGHT MFMAs = 128 cycles; 19 fragments per 88 MFMAs (1.25x less LDS traffic per MFMA), and no second wave to
// share the issue port with: per MFMA (16 cycles of pipe) three issue slots are free for the ~0.4 other instructions it needs.
// Everything else -- LDS-DMA operand stages, source-side swizzle, chunk-major K walk, barrier before the last fragments of a slice,
// whole-line epilogue -- is the production kernel's.  Instantiating the production template at <2, 2> does not test this design point:
// its accumulator array lands in scratch (an epilogue loop over 88 tiles stays rolled) or, unrolled, it spills 140 registers.
#include "conv_common.h"

namespace vd3d_conv {
namespace {

template <int V> struct SI { static constexpr int value = V; };
template <int... Is, class F> VD3D_DEV void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(SI<Is>{}), ...); }
template <int N, class F> VD3D_DEV void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

template <typename T, int BN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) conv_strip1w_kernel(const ConvArgs p) {
    constexpr int BM = 256, NW = 4, WTM = 128, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
    constexpr int A_PIECES = BM / 8 / NW, W_PIECES = BN / 8 / NW, NPIECE = A_PIECES + W_PIECES;
    constexpr int A_STAGE = BM * 128, STAGE = (BM + BN) * 128;
    static_assert(BN % 32 == 0 && (BN / 8) % NW == 0, "strip width");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tile_n = tile / p.tiles_m, tile_m = tile - tile_n * p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l16 = lane & 15, q4 = lane >> 4;

    // folded-BN constants of this tile's channels behind the operand stages
    float* ltab = (float*)(smem + 2 * STAGE);
    float tsc[2], tsh[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 256;
        tsc[u] = 1.f, tsh[u] = 0.f;
        if (i < BN && n0 + i < p.Cout) {
            if (p.scale) tsc[u] = p.scale[n0 + i];
            if (p.shift) tsh[u] = p.shift[n0 + i];
        }
    }
    // ---- loader state -----------------------------------------------------------------------------------------------------------
    const int prow = lane >> 3;
    const int slot = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);     // = (L % 8) ^ ((row / 2) % 8): 8 NW = 32 rows per round keep the phase
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
    int kc = slot * 8, tap = 0, dy = 0, dx = 0;                        // chunk-major / tap-minor K walk (Cin % 64 == 0)
    const uint32_t w_row = (uint32_t)(((n0 + 8 * wave + prow) * p.Kpad + slot * 8) * 2);
    uint32_t w_off = w_row;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.weight, 0, p.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto issue_group = [&](int st, int g, bool enable) {
        char* base = smem + st * STAGE + wave * 1024;
        const bool kvalid = enable && tap < p.ntaps;
        const int ddy = dy * p.dil, ddx = dx * p.dil;
        const int tap_off = ddy * p.in_row_stride + ddx * p.in_pix_stride + kc;
#pragma unroll
        for (int pi = 0; pi < NPIECE; ++pi) {
            if ((pi & 3) != g) continue;
            if (pi < A_PIECES) {
                const bool v = kvalid && (unsigned)(a_iy[pi] + ddy) < (unsigned)p.H && (unsigned)(a_ix[pi] + ddx) < (unsigned)p.W;
                const uint32_t off = v ? (uint32_t)(a_off[pi] + tap_off) * 2u : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(base + pi * NW * 1024), 16, off, 0, 0, 0);
            } else {
                const int it = pi - A_PIECES;
                const uint32_t off = enable ? w_off + (uint32_t)(it * NW * 8 * p.Kpad * 2) : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(base + A_STAGE + it * NW * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    auto advance_k = [&]() {
        ++tap;
        if (++dx == p.kw) { dx = 0; ++dy; }
        if (tap == p.ntaps) { tap = 0; dx = 0; dy = 0; kc += 64; }
        w_off = w_row + (uint32_t)((tap * p.Cin + kc - slot * 8) * 2);
    };
    // fragment addresses: row r, 16-byte slot (4 ks + q4) ^ ((r / 2) % 8); rows of one wave differ by multiples of 16, so the key is
    // the lane's own and block i / j is an immediate offset of 2048 bytes
    const int a_row = wm * WTM + l16, w_rowl = wn * WTN + l16;
    const int a_base0 = a_row * 128 + (((0 + q4) ^ ((a_row >> 1) & 7)) << 4), a_base1 = a_row * 128 + (((4 + q4) ^ ((a_row >> 1) & 7)) << 4);
    const int w_base0 = A_STAGE + w_rowl * 128 + (((0 + q4) ^ ((w_rowl >> 1) & 7)) << 4), w_base1 = A_STAGE + w_rowl * 128 + (((4 + q4) ^ ((w_rowl >> 1) & 7)) << 4);
    auto ld_a = [&](int st, int ks, int j) { return *(const i32x4*)(smem + st * STAGE + (ks ? a_base1 : a_base0) + j * 2048); };
    auto ld_w = [&](int st, int ks, int i) { return *(const i32x4*)(smem + st * STAGE + (ks ? w_base1 : w_base0) + i * 2048); };

    // ---- main loop, hand-ordered: MFMAs, fragment reads and their waits are volatile inline asm in program order; the compiler only
    // allocates registers and fills the gaps with the loader's address arithmetic.  Accumulator tiles 0 .. NA-1 are pinned to AGPRs
    // ("+a"), the rest to VGPRs: left to itself the allocator shuffled ~200 registers per slice between the two files.
    constexpr int NA = 64;                             // 64 tiles x 4 = all 256 AGPRs
    f32x4 acc[TN][TM];
    sfor<TN>([&](auto ic) {
        sfor<TM>([&](auto jc) { acc[decltype(ic)::value][decltype(jc)::value] = f32x4{0.f, 0.f, 0.f, 0.f}; });
    });
    auto mfma = [&](auto tc, const i32x4& wa, const i32x4& px, f32x4& c) {
        constexpr int t = decltype(tc)::value;
        if constexpr (std::is_same<T, hf16>::value) {
            if constexpr (t < NA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(wa), "v"(px));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(wa), "v"(px));
        } else {
            if constexpr (t < NA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(wa), "v"(px));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(wa), "v"(px));
        }
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    // fragment base addresses per (stage, sub-step); block i / j is the instruction's immediate offset (2048 i)
    uint32_t adA[2][2], adW[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        adA[st][0] = lds0 + st * STAGE + a_base0; adA[st][1] = lds0 + st * STAGE + a_base1;
        adW[st][0] = lds0 + st * STAGE + w_base0; adW[st][1] = lds0 + st * STAGE + w_base1;
    }
    auto rd = [&](uint32_t addr, auto oc) {
        i32x4 r;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(decltype(oc)::value * 2048));
        return r;
    };

#pragma unroll
    for (int g = 0; g < 4; ++g) issue_group(0, g, true);
    advance_k();
    issue_group(1, 0, p.nk > 1);
    issue_group(1, 1, p.nk > 1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 256;
        if (i < BN) { ltab[i] = tsc[u]; ltab[BN + i] = tsh[u]; }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    constexpr int NF = 2 * TN;                         // weight fragments per slice
    constexpr int R = 2, F0 = NF - R;                  // ring of two; the barrier sits before the last two fragments of a slice
    i32x4 fa[R], fb[2][TM];
    // LDS queue order (returns are in order): [pixel burst of sub-step 0] [fragment 0] [fragment 1] -- the steady-state order
    sfor<TM>([&](auto jc) { fb[0][decltype(jc)::value] = rd(adA[0][0], jc); });
    fa[0] = rd(adW[0][0], SI<0>{});
    fa[1] = rd(adW[0][0], SI<1>{});
    auto slice = [&](auto stc, int kt) {
        constexpr int st = decltype(stc)::value;
        const bool more1 = kt + 1 < p.nk, more2 = kt + 2 < p.nk;
        sfor<NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value, ks = f / TN, i = f - ks * TN;
            if constexpr (f == F0) {
                // every read of this stage has completed, the next slice has landed -- for everybody
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                issue_group(st, 0, more2);
                issue_group(st, 1, more2);
                // pixel fragments of the next slice's first sub-step (past the last slice they read a dead stage; never consumed)
                sfor<TM>([&](auto jc) { fb[0][decltype(jc)::value] = rd(adA[st ^ 1][0], jc); });
            } else if constexpr (f == 1 || f == 2 || f == NF - 1) {
                asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");        // the wanted fragment is older than a pixel burst + one refill
            } else {
                asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");        // ... than one refill
            }
            sfor<TM>([&](auto jc) { mfma(SI<i * TM + decltype(jc)::value>{}, fa[f % R], fb[ks][decltype(jc)::value], acc[i][decltype(jc)::value]); });
            constexpr int nf = f + R, nks = nf / TN, ni = nf - nks * TN;
            if constexpr (nks < 2) fa[f % R] = rd(adW[st][nks], SI<ni>{});
            else fa[f % R] = rd(adW[st ^ 1][0], SI<nf - NF>{});
            if constexpr (f == 0) {
                sfor<TM>([&](auto jc) { fb[1][decltype(jc)::value] = rd(adA[st][1], jc); });
                issue_group(st ^ 1, 2, more1);
                issue_group(st ^ 1, 3, more1);
                advance_k();
            }
        });
    };
    for (int kt = 0; kt < p.nk; kt += 2) {
        slice(SI<0>{}, kt);
        if (kt + 1 < p.nk) slice(SI<1>{}, kt + 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // dead look-ahead reads; MFMA results -> VALU
    // ---- epilogue: BN (+ residual) (+ ReLU); whole pixel runs through a per-wave LDS tile when the strip lies inside Cout ---------
    __syncthreads();                                   // every wave is done with the operand stages
    const int nw = n0 + wn * WTN;
    constexpr int ROWB = WTN * 2 + 16, CPR = WTN / 8;
    char* park = smem + wave * (16 * ROWB);
    const bool lines = !p.out_f32 && nw + WTN <= p.Cout && p.out_pix_stride % 8 == 0 && ((uintptr_t)p.out & 15) == 0 && (nw % 8) == 0;
    const float relu_lo = p.relu ? 0.f : -3.0e38f;
    sfor<TM>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int m = m0 + wm * WTM + j * 16 + l16;
        const bool mvalid = m < p.M;
        const int64_t rbase = (int64_t)(mvalid ? m : 0) * p.res_pix_stride;
        const int64_t obase = (int64_t)(mvalid ? m : 0) * p.out_pix_stride;
        sfor<TN>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int nb = nw + i * 16 + 4 * q4;
            const f32x4 s = *(const f32x4*)(ltab + (nb - n0)), t = *(const f32x4*)(ltab + BN + (nb - n0));
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * s[e] + t[e];
            if (p.residual && mvalid && nb < p.Cout) {
                const i32x2 rr = *(const i32x2*)(p.residual + (rbase + nb) * 2);
                const uint32_t r0 = (uint32_t)(int)rr[0], r1 = (uint32_t)(int)rr[1];
                v[0] += Fmt16<T>::lo(r0);
                v[1] += Fmt16<T>::hi(r0);
                v[2] += Fmt16<T>::lo(r1);
                v[3] += Fmt16<T>::hi(r1);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], relu_lo);
            if (lines) {
                *(i32x2*)(park + l16 * ROWB + (i * 16 + 4 * q4) * 2) = i32x2{Fmt16<T>::pack2(v[0], v[1]), Fmt16<T>::pack2(v[2], v[3])};
            } else if (mvalid) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (nb + e < p.Cout) {
                        if (p.out_f32) ((float*)p.out)[obase + nb + e] = v[e];
                        else ((T*)p.out)[obase + nb + e] = Fmt16<T>::one(v[e]);
                    }
                }
            }
        });
        if (lines) {
#pragma unroll
            for (int it = 0; it < (16 * CPR + 63) / 64; ++it) {
                const int c = it * 64 + lane;
                if (c < 16 * CPR) {
                    const int r = c / CPR, ch = c - r * CPR;
                    const i32x4 o = *(const i32x4*)(park + r * ROWB + ch * 16);
                    const int mm = m0 + wm * WTM + j * 16 + r;
                    if (mm < p.M) *(i32x4*)(p.out + ((int64_t)mm * p.out_pix_stride + nw) * 2 + ch * 16) = o;
                }
            }
        }
    });
}

}  // namespace

bool strip1w_shape_ok(const ConvArgs& a) { return a.chunk_major && a.Cin % 64 == 0 && !a.group_m; }

template <typename T, int BN>
static int launch_strip1w_t(ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = 2 * (256 + BN) * 128 + 2 * BN * 4;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)conv_strip1w_kernel<T, BN>, LDS, lim, "hipFuncSetAttribute(conv_strip1w)")) return rc;
    const int64_t grid = (int64_t)a.tiles_m * a.tiles_n;
    if (grid <= 0 || grid > 0x7fffffff) return VD3D_EINVAL;
    hipLaunchKernelGGL((conv_strip1w_kernel<T, BN>), dim3((unsigned)grid), dim3(256), LDS, stream, a);
    return vd3d_check_launch("conv_strip1w");
}

int launch_strip1w(ConvArgs& a, hipStream_t stream, int fmt, int bn) {
    if (fmt == VD3D_F16) return bn == 352 ? launch_strip1w_t<hf16, 352>(a, stream) : launch_strip1w_t<hf16, 288>(a, stream);
    return bn == 352 ? launch_strip1w_t<short, 352>(a, stream) : launch_strip1w_t<short, 288>(a, stream);
}

}  // namespace vd3d_conv
