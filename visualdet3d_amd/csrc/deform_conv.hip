// deform_conv.hip -- deformable convolution v1 / v2 (modulated) forward for gfx950.
//
// Replaces deform_conv_ext.{deform_conv_forward, modulated_deform_conv_forward}
// (lib/ops/dcn/src/deform_conv_ext.cpp:149-163; host deform_conv_cuda.cpp:152-260,491-570; kernels
// deform_conv_cuda_kernel.cu:190-243 (v1 im2col), :570-633 (v2 im2col), bilinear :84-115 / :467-497).
// The reference materialises `columns` (C*kh*kw x Ho*Wo fp32 per image) in HBM, then calls a GEMM per image and per
// group.  Here the bilinear-sampled (and mask-modulated) column tile is produced straight into LDS and consumed by
// MFMA in the same workgroup: columns never touch HBM, one launch covers the whole batch, bias / folded BN / ReLU are
// applied in the epilogue.
//
//   out[b,o,y,x] = bias[o] + sum_{tap,c} W[o,c,tap] * m[b,tap,y,x] * bilinear(in[b,c], y*s - p + ty*d + dy, x*s - p + tx*d + dx)
//   a sample is taken iff -1 < h < H and -1 < w < W; corners outside the image contribute 0 (exactly the reference).
//
// Tensors are addressed through element strides, so the same kernel serves the reference's NCHW fp32 extension ABI
// and the engine's NHWC bf16/fp32 activations.  K is ordered tap-major (k = tap*Cg + c): one K slice is one tap x a
// run of channels, so the sampling geometry (4 corner offsets + 4 weights) is computed once per (pixel, tap) and reused
// for the whole channel run.  fp32: v_mfma_f32_32x32x2_f32 (bit-exact fp32 FMA chains); bf16: v_mfma_f32_32x32x16_bf16.
#pragma clang fp contract(off)
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

// compile-time loop (a `#pragma unroll` over a register array indexed by the loop variable may stay rolled and move the array to scratch)
template <int I> struct DIntC { static constexpr int value = I; };
template <int... Is, class F> VD3D_DEV void d_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(DIntC<Is>{}), ...); }
template <int N, class F> VD3D_DEV void static_for(F&& f) { d_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct DcnArgs {
    const void* in; const void* w; const float* bias; const float* scale; const float* shift;
    const float* offset; const float* mask; void* out;
    int B, C, H, W, O, Ho, Wo;
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int groups, dgroups, Cg, Og, Kg, Kpad;   // per-group channels, K = kh*kw*Cg, padded K of the packed weight rows
    int64_t in_sb, in_sc, in_sy, in_sx;      // element strides
    int64_t off_sb, off_sc, off_sy, off_sx;
    int64_t msk_sb, msk_sc, msk_sy, msk_sx;
    int64_t out_sb, out_sc, out_sy, out_sx;
    int mask_sigmoid, relu;
    int no_lstage = 0;                       // VD3D_DCN_NO_LSTAGE=1: logits read lane = pixel from global memory (A/B; same values)
};

template <typename T> VD3D_DEV float ld(const void* p, int64_t i);
template <> VD3D_DEV float ld<float>(const void* p, int64_t i) { return ((const float*)p)[i]; }
template <> VD3D_DEV float ld<short>(const void* p, int64_t i) { return bf2f(((const short*)p)[i]); }
template <> VD3D_DEV float ld<hf16>(const void* p, int64_t i) { return h2f(((const hf16*)p)[i]); }
template <typename T> VD3D_DEV void st(void* p, int64_t i, float v);
template <> VD3D_DEV void st<float>(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
template <> VD3D_DEV void st<short>(void* p, int64_t i, float v) { ((short*)p)[i] = f2bf(v); }
template <> VD3D_DEV void st<hf16>(void* p, int64_t i, float v) { ((hf16*)p)[i] = f2h(v); }

template <typename T> struct DMma {         // 16-bit formats (bf16 | fp16)
    static VD3D_DEV void run(const i32x4& a, const i32x4& b, f32x16& acc) { Fmt16<T>::mfma32(a, b, acc); }
};
template <> struct DMma<float> {
    static VD3D_DEV void run(const i32x4& a, const i32x4& b, f32x16& acc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int aj = a[j], bj = b[j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(i2f(aj), i2f(bj), acc, 0, 0, 0);
        }
    }
};

struct Sample {  // bilinear geometry of one (pixel, tap)
    int64_t o1, o2, o3, o4;   // element offsets of the 4 corners inside one channel plane
    float w1, w2, w3, w4;     // hh*hw, hh*lw, lh*hw, lh*lw -- zeroed for corners outside the image
    float m;                  // modulation (1 for DCNv1); 0 when the sample point itself is out of range
};

template <typename T>
__global__ void __launch_bounds__(256) dcn_kernel(const DcnArgs p) {
    constexpr int ES = (int)sizeof(T);
    constexpr int VE = 16 / ES, BKE = 128 / ES;
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128];
    char* Cs = smem;             // sampled columns  [64 pixels][128 B]
    char* Ws = smem + 64 * 128;  // weights          [64 out-channels][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z;
    const int o0 = blockIdx.y * 64;
    const int grp = o0 / p.Og;           // conv group of this tile (host guarantees a tile never straddles groups)
    const int HoWo = p.Ho * p.Wo;
    const int pix0 = blockIdx.x * 64;
    const int KK = p.kh * p.kw;

    // this thread samples pixel (tid & 63), vectors (tid >> 6) and (tid >> 6) + 4 of each K slice
    const int mypix = pix0 + (tid & 63);
    const bool pvalid = mypix < HoWo;
    const int ho = pvalid ? mypix / p.Wo : 0, wo = pvalid ? mypix - (mypix / p.Wo) * p.Wo : 0;
    const int h_in = ho * p.sh - p.ph, w_in = wo * p.sw - p.pw;
    const int srow = tid & 63;

    auto geometry = [&](int tap, int dgi) {
        Sample s;
        s.o1 = s.o2 = s.o3 = s.o4 = 0;
        s.w1 = s.w2 = s.w3 = s.w4 = 0.f;
        s.m = 0.f;
        if (!pvalid) return s;
        const int ti = tap / p.kw, tj = tap - ti * p.kw;
        const int64_t ob = b * p.off_sb + ho * p.off_sy + wo * p.off_sx;
        const float off_h = p.offset[ob + (int64_t)(dgi * 2 * KK + 2 * tap) * p.off_sc];
        const float off_w = p.offset[ob + (int64_t)(dgi * 2 * KK + 2 * tap + 1) * p.off_sc];
        float m = 1.f;
        if (p.mask) {
            m = p.mask[b * p.msk_sb + ho * p.msk_sy + wo * p.msk_sx + (int64_t)(dgi * KK + tap) * p.msk_sc];
            if (p.mask_sigmoid) m = 1.0f / (1.0f + expf(-m));
        }
        const float h_im = (float)(h_in + ti * p.dh) + off_h;
        const float w_im = (float)(w_in + tj * p.dw) + off_w;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
            const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const bool t_ok = h_low >= 0, b_ok = h_high <= p.H - 1, l_ok = w_low >= 0, r_ok = w_high <= p.W - 1;
            if (t_ok && l_ok) { s.w1 = hh * hw; s.o1 = h_low * p.in_sy + w_low * p.in_sx; }
            if (t_ok && r_ok) { s.w2 = hh * lw; s.o2 = h_low * p.in_sy + w_high * p.in_sx; }
            if (b_ok && l_ok) { s.w3 = lh * hw; s.o3 = h_high * p.in_sy + w_low * p.in_sx; }
            if (b_ok && r_ok) { s.w4 = lh * lw; s.o4 = h_high * p.in_sy + w_high * p.in_sx; }
            s.m = m;
        }
        return s;
    };

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int wn = wave & 1, wm = wave >> 1;     // 2 x 2 waves over (out-channels, pixels)
    const int lr = lane & 31, half = lane >> 5;
    const int cpd = p.C / p.dgroups;             // channels per deformable group

    const int nk = (p.Kg + BKE - 1) / BKE;
    for (int kt = 0; kt < nk; ++kt) {
        // ---- sampled column tile -> LDS -------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int vec = (tid >> 6) + 4 * i;
            const int k0 = kt * BKE + vec * VE;
            float vals[VE];
#pragma unroll
            for (int e = 0; e < VE; ++e) vals[e] = 0.f;
            if (k0 < p.Kg) {
                int tap = k0 / p.Cg, c = k0 - tap * p.Cg;
                int cabs = grp * p.Cg + c;
                Sample s = geometry(tap, cabs / cpd);
                int cur_dg = cabs / cpd;
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    if (k0 + e < p.Kg) {
                        if (c >= p.Cg) {            // vector crosses into the next tap
                            c = 0; ++tap; cabs = grp * p.Cg;
                            s = geometry(tap, cabs / cpd); cur_dg = cabs / cpd;
                        } else if (cabs / cpd != cur_dg) {
                            cur_dg = cabs / cpd; s = geometry(tap, cur_dg);
                        }
                        const int64_t base = b * p.in_sb + (int64_t)cabs * p.in_sc;
                        const float v1 = s.w1 != 0.f ? ld<T>(p.in, base + s.o1) : 0.f;
                        const float v2 = s.w2 != 0.f ? ld<T>(p.in, base + s.o2) : 0.f;
                        const float v3 = s.w3 != 0.f ? ld<T>(p.in, base + s.o3) : 0.f;
                        const float v4 = s.w4 != 0.f ? ld<T>(p.in, base + s.o4) : 0.f;
                        const float val = s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4;
                        vals[e] = val * s.m;
                        ++c; ++cabs;
                    }
                }
            }
            Vec16<T> o;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o.set2(e, vals[2 * e], vals[2 * e + 1]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o.set(e, vals[e]);
            }
            *(i32x4*)(Cs + srow * 128 + ((vec ^ ((srow >> 1) & 7)) << 4)) = o.raw;
        }
        // ---- weight tile -> LDS (packed [O][Kpad], tap-major K) --------------------------------------------
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + 256 * i;
            const int row = v >> 3, slot = v & 7;
            i32x4 wv = {0, 0, 0, 0};
            if (o0 + row < p.O) wv = *(const i32x4*)((const char*)p.w + ((int64_t)(o0 + row) * p.Kpad + kt * BKE + slot * VE) * ES);
            *(i32x4*)(Ws + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4)) = wv;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int sk = 2 * ks + half;
            const int rw = wn * 32 + lr, rc = wm * 32 + lr;
            const i32x4 fa = *(const i32x4*)(Ws + rw * 128 + ((sk ^ ((rw >> 1) & 7)) << 4));
            const i32x4 fb = *(const i32x4*)(Cs + rc * 128 + ((sk ^ ((rc >> 1) & 7)) << 4));
            DMma<T>::run(fa, fb, acc);
        }
        __syncthreads();
    }

    // ---- epilogue: bias, optional folded BN, ReLU; strided store ------------------------------------------
    const int pix = pix0 + wm * 32 + lr;
    if (pix >= HoWo) return;
    const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
    const int64_t ob = b * p.out_sb + oy * p.out_sy + ox * p.out_sx;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = o0 + wn * 32 + 8 * g + 4 * half + e;
            if (o >= p.O) continue;
            float v = acc[4 * g + e];
            if (p.bias) v += p.bias[o];
            if (p.scale) v = v * p.scale[o];
            if (p.shift) v = v + p.shift[o];
            if (p.relu) v = fmaxf(v, 0.f);
            st<T>(p.out, ob + (int64_t)o * p.out_sc, v);
        }
}

template <typename T> constexpr int kDcnGeoBytes = sizeof(T) == 2 ? 32 : 48;   // LDS bytes per (pixel, tap) geometry entry

// ---- NHWC fast path --------------------------------------------------------------------------------------------------
// Channel-contiguous input / output (the engine's activations), groups == deformable_groups == 1, Cg a multiple of the
// 128-byte K slice.  Differences from the generic kernel above, which gathers element by element through arbitrary strides
// (50 ms per KM3D step at 16 x 512 x 1760: 78 % of that model's time):
//   * one K slice = one tap x 64 (bf16) / 32 (fp32) channels: the bilinear geometry is computed once per (pixel, slice)
//     and each corner is ONE 16-byte load per 8 / 4 channels;
//   * the workgroup owns 64 pixels x BN (up to 256) output channels, so the sampled columns are produced once for all
//     output channels instead of once per 64-channel tile;
//   * column and weight tiles are double buffered in LDS: sampling of slice k+1 overlaps the MFMAs of slice k, one barrier
//     per slice.
// v_fma_mix_f32: fp32 fma whose first operand is the low / high fp16 half of a dword -- the unpack costs no instruction
VD3D_DEV float mix_mul_lo(int x, float w) { float d; asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(x), "v"(w)); return d; }
VD3D_DEV float mix_mul_hi(int x, float w) { float d; asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(x), "v"(w)); return d; }
VD3D_DEV float mix_fma_lo(int x, float w, float c) { float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(x), "v"(w), "v"(c)); return d; }
VD3D_DEV float mix_fma_hi(int x, float w, float c) { float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(x), "v"(w), "v"(c)); return d; }

// (Rounds 4 - 5: an opt-in packed fp16 blend on v_pk_fma_f16 -- the reference's own half path blends in scalar_t = half -- measured -2 % on config 5 at
// 0.54 - 1.35 x the 2-ulp bar of the stage-tap tests: outside the parity bar, so it was removed from the library in round 6.)
template <typename T, int BN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 256 ? 2 : 4, 8))) dcn_nhwc_kernel(const DcnArgs p) {
    constexpr int ES = (int)sizeof(T);
    constexpr int VE = 16 / ES, BKE = 128 / ES;
    constexpr int TN = BN / 64;                      // 32-channel MFMA blocks per wave (2 x 2 waves: 32 px x BN/2 ch)
    constexpr int STAGE = (64 + BN) * 128;
    // geometry entry: int32 off[4] (byte offset of the corner's channel run; kDcnOOB = corner outside the image: the buffer load
    // returns zeros without a branch) + float w[4].  16-bit formats: w already carries the modulation (the blended value is
    // rounded to 16 bits next, so (sum w x) m and sum (w m) x agree except on rounding ties); fp32 keeps the reference order
    // with m in a 9th word.
    constexpr bool FOLD = ES == 2;
    constexpr int GE = kDcnGeoBytes<T>;
    constexpr uint32_t kDcnOOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* geo_tab = smem + 2 * STAGE;                // [tap][64 pixels] x entry
    float* ctab = (float*)(geo_tab + 64 * (p.kh * p.kw) * GE);   // bias | scale | shift of this workgroup's BN output channels
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z;
    const int o0 = blockIdx.y * BN;
    const int HoWo = p.Ho * p.Wo;
    const int pix0 = blockIdx.x * 64;
    const int KK = p.kh * p.kw;
    for (int i = tid; i < BN; i += 256) {
        const int o = o0 + i;
        ctab[i] = (p.bias && o < p.O) ? p.bias[o] : 0.f;
        ctab[BN + i] = (p.scale && o < p.O) ? p.scale[o] : 1.f;
        ctab[2 * BN + i] = (p.shift && o < p.O) ? p.shift[o] : 0.f;
    }
    // Sampling map: 8 consecutive lanes fetch the 8 16-byte vectors of ONE pixel's 128-byte channel run, so a corner load of a
    // wave touches 8 cache lines (one per pixel) instead of 64; a thread handles pixels prow0 and prow0 + 32, vector vslot.
    const int prow0 = tid >> 3, vslot = tid & 7;
    // one image / one BN-row weight panel per workgroup: 32-bit offsets, out-of-range rows read as zeros
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.in + (int64_t)b * p.in_sb * ES), 0, 0x7fffffff, 0x00020000);
    const int wrows = p.O - o0 < BN ? p.O - o0 : BN;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w + (int64_t)o0 * p.Kpad * ES), 0, wrows * p.Kpad * ES, 0x00020000);
    const int chunks = p.Cg / BKE;                   // K slices per tap
    const int nk = KK * chunks;

    // ---- phase 0: the sampling geometry of every (pixel, tap) of the tile, once, into LDS.  In the K loop a thread then
    // needs no dependent global load (offset -> address -> corner): the corner loads of slice k+1 are issued straight away
    // and overlap the MFMAs of slice k.
    // a thread owns ONE pixel (lane) and every fourth tap (wave): the pixel's row / column is divided out once and the tap's
    // (ti, tj) is wave-uniform (scalar)
    // Round 4: the logits in the engine's own layout (one pixel = 32 contiguous fp32: offsets 0..17 | mask 18..26 | pad -- what the offset conv
    // writes) are STAGED through LDS first: two fully coalesced 16-byte loads per thread (8 lanes = one pixel's 128 bytes) into the first
    // operand stage (idle until the K loop), rows padded to 33 floats so that the per-pixel reads below are bank-conflict free.  Read from global
    // memory lane = pixel, every one of the nine 4-byte loads per thread touched 64 different cache lines (timing ablation of the same phase in
    // the round-4 LDS-window variant of this kernel: 51 of 372 us).  Any other layout (NCHW offsets of the reference's extension entry points, 27-channel tensors) keeps the old path.
    const bool packed_logits = !p.no_lstage && KK == 9 && p.mask && p.off_sc == 1 && p.msk_sc == 1 && p.mask == p.offset + 18 && p.msk_sb == p.off_sb &&
                               p.msk_sy == p.off_sy && p.msk_sx == p.off_sx && (p.off_sx & 3) == 0 && (p.off_sy & 3) == 0 && (p.off_sb & 3) == 0 &&
                               p.off_sx >= 28 && ((uintptr_t)p.offset & 15) == 0;
    constexpr int LROW = 33;                         // floats per staged pixel row
    float* lstage = (float*)smem;
    if (packed_logits) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int spx = (tid >> 3) + 32 * r, v = tid & 7, spix = pix0 + spx;
            if (spix < HoWo && v < 7) {
                const int soy = spix / p.Wo, sox = spix - soy * p.Wo;
                const f32x4 val = *(const f32x4*)(p.offset + b * p.off_sb + soy * p.off_sy + sox * p.off_sx + 4 * v);
#pragma unroll
                for (int c = 0; c < 4; ++c) lstage[spx * LROW + 4 * v + c] = val[c];
            }
        }
        __syncthreads();
    }
    {
    const int px = lane, pix = pix0 + px;
    const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
    const int64_t ob = b * p.off_sb + oy * p.off_sy + ox * p.off_sx;
    const int64_t mb = b * p.msk_sb + oy * p.msk_sy + ox * p.msk_sx;
    // the logits of this lane's taps (every fourth: wave, wave + 4, ...) are requested up front, three taps at a time, offsets AND
    // mask -- read one tap after the other, with the mask read only once the offsets say the sample is inside the image, the
    // phase is six dependent memory round trips (12.4k of a workgroup's 39.6k cycles, cycle stamps)
    constexpr int TPW = 3;
    for (int t0 = wave; t0 < KK; t0 += 4 * TPW) {
        float oh[TPW], ow[TPW], ml[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int tap = t0 + 4 * u;
            oh[u] = ow[u] = ml[u] = 0.f;
            if (tap < KK && pix < HoWo) {
                if (packed_logits) {
                    oh[u] = lstage[px * LROW + 2 * tap];
                    ow[u] = lstage[px * LROW + 2 * tap + 1];
                    ml[u] = lstage[px * LROW + 18 + tap];
                } else {
                    oh[u] = p.offset[ob + (int64_t)(2 * tap) * p.off_sc];
                    ow[u] = p.offset[ob + (int64_t)(2 * tap + 1) * p.off_sc];
                    if (p.mask) ml[u] = p.mask[mb + (int64_t)tap * p.msk_sc];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
        const int tap = t0 + 4 * u;
        if (tap >= KK) break;
        const int it = tap * 64 + px;
        uint32_t go[4] = {kDcnOOB, kDcnOOB, kDcnOOB, kDcnOOB};
        float gw[4] = {0.f, 0.f, 0.f, 0.f};
        float m = 0.f;
        if (pix < HoWo) {
            const int ti = tap / p.kw, tj = tap - ti * p.kw;
            const float off_h = oh[u];
            const float off_w = ow[u];
            const float h_im = (float)(oy * p.sh - p.ph + ti * p.dh) + off_h;
            const float w_im = (float)(ox * p.sw - p.pw + tj * p.dw) + off_w;
            if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                m = 1.f;
                if (p.mask) {
                    m = ml[u];
                    // 16-bit formats: v_exp / v_rcp (1 ulp each; the blended value is rounded to 11 / 8 bits next)
                    if (p.mask_sigmoid) m = FOLD ? __frcp_rn(1.0f + __expf(-m)) : 1.0f / (1.0f + expf(-m));
                }
                const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                const int h_high = h_low + 1, w_high = w_low + 1;
                const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const bool t_ok = h_low >= 0, b_ok = h_high <= p.H - 1, l_ok = w_low >= 0, r_ok = w_high <= p.W - 1;
                if (t_ok && l_ok) { gw[0] = hh * hw; go[0] = (uint32_t)((h_low * p.in_sy + w_low * p.in_sx) * ES); }
                if (t_ok && r_ok) { gw[1] = hh * lw; go[1] = (uint32_t)((h_low * p.in_sy + w_high * p.in_sx) * ES); }
                if (b_ok && l_ok) { gw[2] = lh * hw; go[2] = (uint32_t)((h_high * p.in_sy + w_low * p.in_sx) * ES); }
                if (b_ok && r_ok) { gw[3] = lh * lw; go[3] = (uint32_t)((h_high * p.in_sy + w_high * p.in_sx) * ES); }
            }
        }
        if constexpr (FOLD) {
#pragma unroll
            for (int c = 0; c < 4; ++c) gw[c] *= m;
        } else {
            *(float*)(geo_tab + (size_t)it * GE + 32) = m;   // weights stay unmodulated: the modulation multiplies the blended value
        }
        *(i32x4*)(geo_tab + (size_t)it * GE) = i32x4{(int)go[0], (int)go[1], (int)go[2], (int)go[3]};
        *(f32x4*)(geo_tab + (size_t)it * GE + 16) = f32x4{gw[0], gw[1], gw[2], gw[3]};
        }
    }
    }
    __syncthreads();

    struct Geo { int32_t o[4]; float w[4]; float m; };
    auto geometry = [&](int tap, int prow) {
        Geo g;
        const char* e = geo_tab + (size_t)(tap * 64 + prow) * GE;
        const i32x4 go = *(const i32x4*)e;
        const f32x4 gw = *(const f32x4*)(e + 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) { g.o[c] = go[c]; g.w[c] = gw[c]; }
        g.m = FOLD ? 1.f : *(const float*)(e + 32);
        return g;
    };
    // global -> registers for slice kt (sampled column vectors + this thread's share of the weight tile)
    constexpr int WV = BN * 8 / 256;                 // 16-byte weight vectors per thread per slice
    int w_voff[WV];
#pragma unroll
    for (int i = 0; i < WV; ++i) {
        const int v = tid + 256 * i;
        w_voff[i] = ((v >> 3) * p.Kpad + (v & 7) * VE) * ES;
    }
    Geo geo[2] = {geometry(0, prow0), geometry(0, prow0 + 32)};
    int geo_tap = 0;
    auto fetch = [&](int kt, i32x4 (&cv)[2][4], i32x4 (&wv)[WV]) {
        const int tap = kt / chunks, c0 = (kt - tap * chunks) * BKE;
        if (tap != geo_tap) { geo[0] = geometry(tap, prow0); geo[1] = geometry(tap, prow0 + 32); geo_tap = tap; }
        const int coff = (c0 + vslot * VE) * ES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                cv[i][c] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, geo[i].o[c] + coff, 0, 0));
#pragma unroll
        for (int i = 0; i < WV; ++i)
            wv[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff[i], kt * (BKE * ES), 0));
    };
    // registers -> LDS stage: blend the four corners in fp32 (reference order), modulate, round, store swizzled
    auto stash = [&](int st, const i32x4 (&cv)[2][4], const i32x4 (&wv)[WV]) {
        char* Cs = smem + st * STAGE;
        char* Ws = Cs + 64 * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            Vec16<T> c1, c2, c3, c4, o;
            c1.raw = cv[i][0]; c2.raw = cv[i][1]; c3.raw = cv[i][2]; c4.raw = cv[i][3];
            float vals[VE];
            if constexpr (std::is_same<T, hf16>::value) {
                // fp16: the four-term fma chain on v_fma_mix_f32 (fp16 operand read in place, fp32 weight and accumulator)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    vals[2 * d] = mix_fma_lo(cv[i][3][d], geo[i].w[3], mix_fma_lo(cv[i][2][d], geo[i].w[2], mix_fma_lo(cv[i][1][d], geo[i].w[1], mix_mul_lo(cv[i][0][d], geo[i].w[0]))));
                    vals[2 * d + 1] = mix_fma_hi(cv[i][3][d], geo[i].w[3], mix_fma_hi(cv[i][2][d], geo[i].w[2], mix_fma_hi(cv[i][1][d], geo[i].w[1], mix_mul_hi(cv[i][0][d], geo[i].w[0]))));
                }
            } else {
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    if constexpr (sizeof(T) == 2) {
                        // bf16 mode: the result is rounded to bf16 anyway -- fused multiply-adds (4 ops instead of 7)
                        vals[e] = fmaf(geo[i].w[3], c4.get(e), fmaf(geo[i].w[2], c3.get(e), fmaf(geo[i].w[1], c2.get(e), geo[i].w[0] * c1.get(e))));
                    } else {
                        vals[e] = (geo[i].w[0] * c1.get(e) + geo[i].w[1] * c2.get(e) + geo[i].w[2] * c3.get(e) + geo[i].w[3] * c4.get(e)) * geo[i].m;
                    }
                }
            }
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o.set2(e, vals[2 * e], vals[2 * e + 1]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o.set(e, vals[e]);
            }
            const int row = prow0 + 32 * i;
            *(i32x4*)(Cs + row * 128 + ((vslot ^ ((row >> 1) & 7)) << 4)) = o.raw;
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + 256 * i, row = v >> 3, slot = v & 7;
            *(i32x4*)(Ws + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4)) = wv[i];
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int wn = wave & 1, wm = wave >> 1;         // 2 x 2 waves over (out-channel halves, pixel halves)
    const int lr = lane & 31, half = lane >> 5;

    i32x4 cv[2][4], wv[WV];
    fetch(0, cv, wv);
    stash(0, cv, wv);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) fetch(kt + 1, cv, wv);             // global loads in flight under the MFMAs below (`geo` now = slice kt+1's)
        const char* Cs = smem + st * STAGE;
        const char* Ws = Cs + 64 * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int sk = 2 * ks + half;
            const int rc = wm * 32 + lr;
            const i32x4 fb = *(const i32x4*)(Cs + rc * 128 + ((sk ^ ((rc >> 1) & 7)) << 4));
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int rw = wn * (BN / 2) + i * 32 + lr;
                const i32x4 fa = *(const i32x4*)(Ws + rw * 128 + ((sk ^ ((rw >> 1) & 7)) << 4));
                DMma<T>::run(fa, fb, acc[i]);
            }
        }
        if (more) stash(st ^ 1, cv, wv);             // the other stage was last read in slice kt-1 (barrier below covers it)
        __syncthreads();
    }
    // ---- epilogue: bias, optional folded BN, ReLU; 4 consecutive channels per accumulator quad -> NHWC vector stores ----
    const int pix = pix0 + wm * 32 + lr;
    if (pix >= HoWo) return;
    const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
    const int64_t ob = b * p.out_sb + oy * p.out_sy + ox * p.out_sx;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int oc = o0 + wn * (BN / 2) + i * 32 + 8 * g + 4 * half;
            float v[4];
            {
                // the quad's per-channel constants from the LDS table written at kernel start (bias 0 / scale 1 / shift 0 where the
                // caller passed none: exact identities); read from global here they were 12 more 1 KiB loads per thread on the L1
                // path that already bounds this kernel, exposed at the end of the workgroup
                const int lc = wn * (BN / 2) + i * 32 + 8 * g + 4 * half;
                const f32x4 bz = *(const f32x4*)(ctab + lc), sc = *(const f32x4*)(ctab + BN + lc), sh = *(const f32x4*)(ctab + 2 * BN + lc);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = (acc[i][4 * g + e] + bz[e]) * sc[e] + sh[e];
                    if (p.relu) x = fmaxf(x, 0.f);
                    v[e] = x;
                }
            }
            if (oc + 3 < p.O && ((ob + oc) * ES) % (4 * ES) == 0) {
                if constexpr (sizeof(T) == 2) {
                    i32x2 o2;
                    o2[0] = Fmt16<T>::pack2(v[0], v[1]);
                    o2[1] = Fmt16<T>::pack2(v[2], v[3]);
                    *(i32x2*)((char*)p.out + (ob + oc) * 2) = o2;
                } else {
                    *(f32x4*)((char*)p.out + (ob + oc) * 4) = f32x4{v[0], v[1], v[2], v[3]};
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (oc + e < p.O) st<T>(p.out, ob + oc + e, v[e]);
            }
        }
}

// ---- 64 -> 64 channel DCNv2 (KM3D's five full-resolution DLA-Up nodes): what was built, measured and REMOVED from the product library ----
// Four alternatives to dcn_nhwc_kernel<T, 64> were built in rounds 3 - 4, each bit-identical (or within 1 ulp) to it and each measured
// SLOWER or equal inside the model; they shipped as opt-in switches until round 5 and live on in the history (last present in commit
// 3b99961, csrc/deform_conv.hip: dcn_win64_kernel -- corners from an LDS-staged 15 x 15 window, weights resident in LDS, persistent;
// dcn_ks64_kernel -- the same with the weights K-split over the registers of 8 waves; dcn_lw64_kernel -- the gather kernel with its corners
// from an LDS window; dcn_bf64_kernel -- barrier-free K loop, the blended vector is the MFMA B fragment).  What they established
// (DESIGN.md sections 9, 10, 11): the operator is bound by VALU issue, not by the L1 request path (taking 70 % of the gathers off that path
// changed nothing); the exact fp32 blend of 16-bit data is 32 v_fma_mix per 8 channels, corner and tap -- 36 wave-instructions per output pixel
// before any geometry, address arithmetic or epilogue; every variant runs 73 - 78 wave-instructions per pixel at 40 - 75 % VALU utilisation.
// The packed fp16 blend (dcn_blend8_pk above, OPT-IN, halves the blend) stays outside the 2-ulp bar the teacher-forced tests hold this path to.

// Round 5, a fifth one, built and removed in the same round (source as measured + numbers: profiles/r05_dcn_geo64_experiment.txt): the barrier-free kernel WITHOUT the LDS
// window -- lane (pixel n, k-group g) gathers its own four corners straight from global memory and the blended vector is the MFMA B fragment; evenly
// spread geometry phase, 63 wave-instructions per pixel (ISA count; the tap-by-tap kernel: 78), 51 KB of LDS and 137 - 165 registers: three workgroups per
// CU; bit-identical.  MEASURED (16 x 128 x 440 fp16, one box): 698 / 720 / 646 us at offsets sigma 0.5 / 1.5 / 40 px against 358 / 375 / 385 us for
// dcn_nhwc_kernel<T, 64>; config 5 11.0 against 9.1 ms per step.  TWICE as slow with 20 % fewer instructions: in the fragment mapping a corner load is
// 16 pixels x 64 contiguous bytes (and an A-fragment load 16 weight rows x 64 bytes) -- 16 half-used cache lines per instruction where the gather mapping
// of dcn_nhwc_kernel (eight lanes = one pixel's 128-byte run) touches 8 whole ones; the vector memory path serves scattered 16-byte lanes at about half
// the rate.  The round-4 barrier-free kernel only looked viable because its corners came from LDS.  Lesson: for this operator the 8-lanes-per-pixel gather
// mapping is not negotiable, so the blended columns MUST change lanes before the MFMA (an LDS tile), and with them comes the per-tap skeleton.

template <typename T>
__global__ void __launch_bounds__(256) dcn_columns_kernel(const DcnArgs p, T* __restrict__ cols) {
    constexpr int ES = (int)sizeof(T);
    constexpr int VE = 16 / ES;
    const int KK = p.kh * p.kw;
    const int cv = p.C / VE;
    const int64_t total = (int64_t)p.B * p.Ho * p.Wo * KK * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VE;
        int64_t r = i / cv;
        const int tap = (int)(r % KK);
        r /= KK;
        const int ox = (int)(r % p.Wo), oy = (int)((r / p.Wo) % p.Ho), b = (int)(r / ((int64_t)p.Wo * p.Ho));
        const int ti = tap / p.kw, tj = tap - ti * p.kw;
        const int64_t ob = b * p.off_sb + oy * p.off_sy + ox * p.off_sx;
        const float off_h = p.offset[ob + (int64_t)(2 * tap) * p.off_sc];
        const float off_w = p.offset[ob + (int64_t)(2 * tap + 1) * p.off_sc];
        float m = 1.f;
        if (p.mask) {
            m = p.mask[b * p.msk_sb + oy * p.msk_sy + ox * p.msk_sx + (int64_t)tap * p.msk_sc];
            if (p.mask_sigmoid) m = 1.0f / (1.0f + expf(-m));
        }
        const float h_im = (float)(oy * p.sh - p.ph + ti * p.dh) + off_h;
        const float w_im = (float)(ox * p.sw - p.pw + tj * p.dw) + off_w;
        float vals[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) vals[e] = 0.f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
            const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const bool t_ok = h_low >= 0, b_ok = h_high <= p.H - 1, l_ok = w_low >= 0, r_ok = w_high <= p.W - 1;
            const float gw[4] = {t_ok && l_ok ? hh * hw : 0.f, t_ok && r_ok ? hh * lw : 0.f, b_ok && l_ok ? lh * hw : 0.f, b_ok && r_ok ? lh * lw : 0.f};
            const int64_t go[4] = {h_low * p.in_sy + w_low * p.in_sx, h_low * p.in_sy + w_high * p.in_sx,
                                   h_high * p.in_sy + w_low * p.in_sx, h_high * p.in_sy + w_high * p.in_sx};
            const char* in_b = (const char*)p.in + ((int64_t)b * p.in_sb + c) * ES;
            Vec16<T> cr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cr[k].raw = i32x4{0, 0, 0, 0};
                if (gw[k] != 0.f) cr[k].raw = *(const i32x4*)(in_b + go[k] * ES);
            }
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                float val;
                if constexpr (sizeof(T) == 2)
                    val = fmaf(gw[3], cr[3].get(e), fmaf(gw[2], cr[2].get(e), fmaf(gw[1], cr[1].get(e), gw[0] * cr[0].get(e))));
                else
                    val = gw[0] * cr[0].get(e) + gw[1] * cr[1].get(e) + gw[2] * cr[2].get(e) + gw[3] * cr[3].get(e);
                vals[e] = val * m;
            }
        }
        Vec16<T> o;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, vals[2 * e], vals[2 * e + 1]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, vals[e]);
        }
        *(i32x4*)((char*)cols + i * 16) = o.raw;
    }
}

// Wave-per-sample variant of the columns kernel (inputs below 2 GiB: 32-bit offsets): a lane computes the geometry of ONE (pixel,
// tap) sample, then the wave walks its 64 samples: the sample's four corner offsets and weights are broadcast (v_readlane) and
// the 64 lanes sweep the channel run -- four 1 KiB corner loads, the fp32 blend, one 1 KiB store per 512 | 256 channels.
// The thread-per-vector kernel above recomputes the geometry (three strided logit loads, sigmoid, floor, 64-bit index
// divisions) for every 8 channels: at C = 2176 that is 272 times per sample, twice the blend work itself.
template <typename T>
__global__ void __launch_bounds__(256) dcn_columns_wave_kernel(const DcnArgs p, T* __restrict__ cols, uint32_t in_bytes) {
    constexpr int ES = (int)sizeof(T);
    constexpr int VE = 16 / ES;
    constexpr bool FOLD = ES == 2;                   // (16-bit formats: modulation folded into the weights, as in dcn_nhwc_kernel)
    constexpr uint32_t kDcnOOB = 0x80000000u;
    const int KK = p.kh * p.kw;
    const int lane = threadIdx.x & 63;
    const int64_t nsamp = (int64_t)p.B * p.Ho * p.Wo * KK;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (int64_t)gridDim.x * 4;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, in_bytes, 0x00020000);
    const int cbytes = p.C * ES;
    // work item = (64 samples, one 1 KiB channel step): the wave keeps ONE channel step and walks its samples, so the corner runs
    // of neighbouring taps / pixels (the same few input lines) are re-read while they are still in L1 -- sweeping all channels
    // of a sample before the next sample streams 17 KB per sample through a 32 KB L1 and re-fetches the input 36 times from
    // beyond L2 (measured: 6 TB/s of reads, 1.19 ms at 32 x 18 x 80 x 2176)
    const int nsteps = (cbytes + 1023) / 1024;
    const int64_t nitems = ((nsamp + 63) / 64) * nsteps;
    for (int64_t item = wave0; item < nitems; item += nwave) {
        const int64_t base = (item / nsteps) * 64;
        const int cb = (int)(item % nsteps) * 1024 + lane * 16;
        uint32_t go[4] = {kDcnOOB, kDcnOOB, kDcnOOB, kDcnOOB};
        float gw[4] = {0.f, 0.f, 0.f, 0.f};
        float m = 0.f;
        const int64_t sidx = base + lane;
        if (sidx < nsamp) {
            const int tap = (int)(sidx % KK);
            const int64_t r = sidx / KK;
            const int ox = (int)(r % p.Wo), oy = (int)((r / p.Wo) % p.Ho), b = (int)(r / ((int64_t)p.Wo * p.Ho));
            const int ti = tap / p.kw, tj = tap - ti * p.kw;
            const int64_t ob = b * p.off_sb + oy * p.off_sy + ox * p.off_sx;
            const float off_h = p.offset[ob + (int64_t)(2 * tap) * p.off_sc];
            const float off_w = p.offset[ob + (int64_t)(2 * tap + 1) * p.off_sc];
            const float h_im = (float)(oy * p.sh - p.ph + ti * p.dh) + off_h;
            const float w_im = (float)(ox * p.sw - p.pw + tj * p.dw) + off_w;
            if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                m = 1.f;
                if (p.mask) {
                    m = p.mask[b * p.msk_sb + oy * p.msk_sy + ox * p.msk_sx + (int64_t)tap * p.msk_sc];
                    if (p.mask_sigmoid) m = FOLD ? __frcp_rn(1.0f + __expf(-m)) : 1.0f / (1.0f + expf(-m));
                }
                const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                const int h_high = h_low + 1, w_high = w_low + 1;
                const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const bool t_ok = h_low >= 0, b_ok = h_high <= p.H - 1, l_ok = w_low >= 0, r_ok = w_high <= p.W - 1;
                const int64_t ib = (int64_t)b * p.in_sb;
                if (t_ok && l_ok) { gw[0] = hh * hw; go[0] = (uint32_t)((ib + h_low * p.in_sy + w_low * p.in_sx) * ES); }
                if (t_ok && r_ok) { gw[1] = hh * lw; go[1] = (uint32_t)((ib + h_low * p.in_sy + w_high * p.in_sx) * ES); }
                if (b_ok && l_ok) { gw[2] = lh * hw; go[2] = (uint32_t)((ib + h_high * p.in_sy + w_low * p.in_sx) * ES); }
                if (b_ok && r_ok) { gw[3] = lh * lw; go[3] = (uint32_t)((ib + h_high * p.in_sy + w_high * p.in_sx) * ES); }
            }
        }
        if constexpr (FOLD) {
#pragma unroll
            for (int c = 0; c < 4; ++c) gw[c] *= m;
        }
        const int nj = nsamp - base < 64 ? (int)(nsamp - base) : 64;
#pragma unroll 4
        for (int j = 0; j < nj; ++j) {
            uint32_t o[4];
            float w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                o[c] = (uint32_t)__builtin_amdgcn_readlane((int)go[c], j);
                w[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gw[c]), j));
            }
            const float mj = FOLD ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), j));
            char* dst = (char*)cols + (base + j) * cbytes;
            {
                i32x4 cv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) cv[c] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, cb < cbytes ? o[c] + cb : kDcnOOB, 0, 0));
                float vals[VE];
                if constexpr (std::is_same<T, hf16>::value) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        vals[2 * d] = mix_fma_lo(cv[3][d], w[3], mix_fma_lo(cv[2][d], w[2], mix_fma_lo(cv[1][d], w[1], mix_mul_lo(cv[0][d], w[0]))));
                        vals[2 * d + 1] = mix_fma_hi(cv[3][d], w[3], mix_fma_hi(cv[2][d], w[2], mix_fma_hi(cv[1][d], w[1], mix_mul_hi(cv[0][d], w[0]))));
                    }
                } else {
                    Vec16<T> c1, c2, c3, c4;
                    c1.raw = cv[0]; c2.raw = cv[1]; c3.raw = cv[2]; c4.raw = cv[3];
#pragma unroll
                    for (int e = 0; e < VE; ++e) {
                        if constexpr (sizeof(T) == 2) vals[e] = fmaf(w[3], c4.get(e), fmaf(w[2], c3.get(e), fmaf(w[1], c2.get(e), w[0] * c1.get(e))));
                        else vals[e] = (w[0] * c1.get(e) + w[1] * c2.get(e) + w[2] * c3.get(e) + w[3] * c4.get(e)) * mj;
                    }
                }
                Vec16<T> ov;
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov.set2(e, vals[2 * e], vals[2 * e + 1]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov.set(e, vals[e]);
                }
                if (cb < cbytes) *(i32x4*)(dst + cb) = ov.raw;
            }
        }
    }
}

template <typename T, int BN>
int launch_dcn_nhwc(const DcnArgs& a, hipStream_t s) {
    const int LDS = 2 * (64 + BN) * 128 + 64 * a.kh * a.kw * kDcnGeoBytes<T> + 3 * BN * 4;   // two stages + the geometry table + the epilogue constants
    if (LDS > 160 * 1024) { vd3d_set_error("deform_conv: kernel window too large for the NHWC path"); return VD3D_EINVAL; }
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)dcn_nhwc_kernel<T, BN>, 160 * 1024, lim, "hipFuncSetAttribute(dcn_nhwc)")) return rc;
    dim3 grid((a.Ho * a.Wo + 63) / 64, (a.O + BN - 1) / BN, a.B);
    hipLaunchKernelGGL((dcn_nhwc_kernel<T, BN>), grid, dim3(256), LDS, s, a);
    return vd3d_check_launch("deform_conv(nhwc)");
}

template <typename T>
int dispatch_dcn_nhwc(const DcnArgs& a, hipStream_t s) {
    if (a.O > 128) return launch_dcn_nhwc<T, 256>(a, s);
    if (a.O > 64) return launch_dcn_nhwc<T, 128>(a, s);
    return launch_dcn_nhwc<T, 64>(a, s);
}

int launch_dcn(const vd3d_dcn_params* q, hipStream_t s) {
    if (!q || !q->in || !q->weight || !q->offset || !q->out) { vd3d_set_error("deform_conv: null pointer"); return VD3D_EINVAL; }
    if (q->dtype != VD3D_BF16 && q->dtype != VD3D_F16 && q->dtype != VD3D_F32) { vd3d_set_error("deform_conv: bad dtype"); return VD3D_EINVAL; }
    if (q->groups < 1 || q->deformable_groups < 1 || q->C % q->groups || q->O % q->groups || q->C % q->deformable_groups) {
        vd3d_set_error("deform_conv: channels must divide by groups / deformable_groups");
        return VD3D_EINVAL;
    }
    const int Og = q->O / q->groups;
    if (q->groups > 1 && Og % 64) { vd3d_set_error("deform_conv: groups > 1 needs out-channels per group to be a multiple of 64"); return VD3D_EINVAL; }
    const int es = q->dtype == VD3D_F32 ? 4 : 2, bke = 128 / es;
    DcnArgs a;
    a.in = q->in; a.w = q->weight; a.bias = q->bias; a.scale = q->scale; a.shift = q->shift; a.offset = q->offset; a.mask = q->mask; a.out = q->out;
    a.B = q->B; a.C = q->C; a.H = q->H; a.W = q->W; a.O = q->O;
    a.kh = q->kh; a.kw = q->kw; a.sh = q->stride_h; a.sw = q->stride_w; a.ph = q->pad_h; a.pw = q->pad_w; a.dh = q->dil_h; a.dw = q->dil_w;
    a.Ho = (q->H + 2 * q->pad_h - (q->dil_h * (q->kh - 1) + 1)) / q->stride_h + 1;
    a.Wo = (q->W + 2 * q->pad_w - (q->dil_w * (q->kw - 1) + 1)) / q->stride_w + 1;
    a.groups = q->groups; a.dgroups = q->deformable_groups; a.Cg = q->C / q->groups; a.Og = Og;
    a.Kg = q->kh * q->kw * a.Cg; a.Kpad = q->Kpad;
    if (a.Kpad % bke || a.Kpad < a.Kg || ((uintptr_t)q->weight & 15)) { vd3d_set_error("deform_conv: packed weight padding / alignment"); return VD3D_EINVAL; }
    a.in_sb = q->in_strides[0]; a.in_sc = q->in_strides[1]; a.in_sy = q->in_strides[2]; a.in_sx = q->in_strides[3];
    a.off_sb = q->offset_strides[0]; a.off_sc = q->offset_strides[1]; a.off_sy = q->offset_strides[2]; a.off_sx = q->offset_strides[3];
    a.msk_sb = q->mask_strides[0]; a.msk_sc = q->mask_strides[1]; a.msk_sy = q->mask_strides[2]; a.msk_sx = q->mask_strides[3];
    a.out_sb = q->out_strides[0]; a.out_sc = q->out_strides[1]; a.out_sy = q->out_strides[2]; a.out_sx = q->out_strides[3];
    a.mask_sigmoid = q->mask_sigmoid; a.relu = q->relu;
    a.no_lstage = vd3d_switch(VD3D_SW_DCN_NO_LSTAGE) ? 1 : 0;
    if (a.Ho <= 0 || a.Wo <= 0 || q->B <= 0) { vd3d_set_error("deform_conv: empty output"); return VD3D_EINVAL; }
    // channel-contiguous activations, one group: the NHWC fast path (everything the detectors launch)
    if (a.groups == 1 && a.dgroups == 1 && a.in_sc == 1 && a.out_sc == 1 && a.Cg % bke == 0 && ((uintptr_t)q->in & 15) == 0 &&
        (int64_t)a.H * a.in_sy * es < 0x7fffffffll && 64 * a.kh * a.kw * 48 + 2 * (64 + 256) * 128 <= 160 * 1024 &&
        a.in_sx % (16 / es) == 0 && a.in_sy % (16 / es) == 0 && a.in_sb % (16 / es) == 0 && !vd3d_switch(VD3D_SW_DCN_GENERIC))
        return q->dtype == VD3D_BF16 ? dispatch_dcn_nhwc<short>(a, s) : (q->dtype == VD3D_F16 ? dispatch_dcn_nhwc<hf16>(a, s) : dispatch_dcn_nhwc<float>(a, s));
    dim3 grid((a.Ho * a.Wo + 63) / 64, (q->O + 63) / 64, q->B);
    if (q->dtype == VD3D_BF16) hipLaunchKernelGGL(dcn_kernel<short>, grid, dim3(256), 0, s, a);
    else if (q->dtype == VD3D_F16) hipLaunchKernelGGL(dcn_kernel<hf16>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(dcn_kernel<float>, grid, dim3(256), 0, s, a);
    return vd3d_check_launch("deform_conv");
}

int launch_dcn_columns(const vd3d_dcn_params* q, void* columns, hipStream_t s) {
    if (!q || !q->in || !q->offset || !columns) { vd3d_set_error("deform_columns: null pointer"); return VD3D_EINVAL; }
    if (q->dtype != VD3D_BF16 && q->dtype != VD3D_F16 && q->dtype != VD3D_F32) { vd3d_set_error("deform_columns: bad dtype"); return VD3D_EINVAL; }
    const int es = q->dtype == VD3D_F32 ? 4 : 2, ve = 16 / es;
    if (q->groups != 1 || q->deformable_groups != 1 || q->in_strides[1] != 1 || q->C % ve || ((uintptr_t)q->in & 15) || ((uintptr_t)columns & 15) ||
        q->in_strides[0] % ve || q->in_strides[2] % ve || q->in_strides[3] % ve) {
        vd3d_set_error("deform_columns: needs channel-contiguous (NHWC) 16-byte aligned input, groups = deformable_groups = 1");
        return VD3D_EINVAL;
    }
    DcnArgs a;
    a.in = q->in; a.w = nullptr; a.bias = nullptr; a.scale = nullptr; a.shift = nullptr; a.offset = q->offset; a.mask = q->mask; a.out = nullptr;
    a.B = q->B; a.C = q->C; a.H = q->H; a.W = q->W; a.O = q->O;
    a.kh = q->kh; a.kw = q->kw; a.sh = q->stride_h; a.sw = q->stride_w; a.ph = q->pad_h; a.pw = q->pad_w; a.dh = q->dil_h; a.dw = q->dil_w;
    a.Ho = (q->H + 2 * q->pad_h - (q->dil_h * (q->kh - 1) + 1)) / q->stride_h + 1;
    a.Wo = (q->W + 2 * q->pad_w - (q->dil_w * (q->kw - 1) + 1)) / q->stride_w + 1;
    a.groups = 1; a.dgroups = 1; a.Cg = q->C; a.Og = q->O; a.Kg = q->kh * q->kw * q->C; a.Kpad = a.Kg;
    a.in_sb = q->in_strides[0]; a.in_sc = 1; a.in_sy = q->in_strides[2]; a.in_sx = q->in_strides[3];
    a.off_sb = q->offset_strides[0]; a.off_sc = q->offset_strides[1]; a.off_sy = q->offset_strides[2]; a.off_sx = q->offset_strides[3];
    a.msk_sb = q->mask_strides[0]; a.msk_sc = q->mask_strides[1]; a.msk_sy = q->mask_strides[2]; a.msk_sx = q->mask_strides[3];
    a.out_sb = a.out_sc = a.out_sy = a.out_sx = 0;
    a.mask_sigmoid = q->mask_sigmoid; a.relu = 0;
    if (a.Ho <= 0 || a.Wo <= 0 || q->B <= 0) { vd3d_set_error("deform_columns: empty output"); return VD3D_EINVAL; }
    const int64_t in_span = ((int64_t)(q->B - 1) * a.in_sb + (int64_t)(q->H - 1) * a.in_sy + (int64_t)(q->W - 1) * a.in_sx + q->C) * es;
    if (in_span < 0x7ffffff0ll && !vd3d_switch(VD3D_SW_DCN_COLUMNS_GENERIC)) {
        // wave-per-sample kernel: geometry once per (pixel, tap) instead of once per 16-byte vector
        const int64_t nsamp = (int64_t)q->B * a.Ho * a.Wo * q->kh * q->kw;
        const int64_t nitems = ((nsamp + 63) / 64) * (((int64_t)q->C * es + 1023) / 1024);   // (64 samples, 1 KiB channel step) per wave
        const int64_t want = (nitems + 3) / 4;
        const int cus = vd3d_device_cu_count();
        const int gridw = (int)(want < (int64_t)cus * 8 ? want : (int64_t)cus * 8);
        if (gridw <= 0) return VD3D_ELAUNCH;
        if (q->dtype == VD3D_BF16) hipLaunchKernelGGL(dcn_columns_wave_kernel<short>, dim3(gridw), dim3(256), 0, s, a, (short*)columns, (uint32_t)in_span);
        else if (q->dtype == VD3D_F16) hipLaunchKernelGGL(dcn_columns_wave_kernel<hf16>, dim3(gridw), dim3(256), 0, s, a, (hf16*)columns, (uint32_t)in_span);
        else hipLaunchKernelGGL(dcn_columns_wave_kernel<float>, dim3(gridw), dim3(256), 0, s, a, (float*)columns, (uint32_t)in_span);
        return vd3d_check_launch("deform_columns");
    }
    const int64_t total = (int64_t)q->B * a.Ho * a.Wo * q->kh * q->kw * (q->C / ve);
    const int grid = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    if (q->dtype == VD3D_BF16) hipLaunchKernelGGL(dcn_columns_kernel<short>, dim3(grid), dim3(256), 0, s, a, (short*)columns);
    else if (q->dtype == VD3D_F16) hipLaunchKernelGGL(dcn_columns_kernel<hf16>, dim3(grid), dim3(256), 0, s, a, (hf16*)columns);
    else hipLaunchKernelGGL(dcn_columns_kernel<float>, dim3(grid), dim3(256), 0, s, a, (float*)columns);
    return vd3d_check_launch("deform_columns");
}

// OIHW fp32 -> packed [O][Kpad] (tap-major K) in the compute dtype
template <typename T>
__global__ void dcn_pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int O, int Cg, int KK, int Kpad) {
    const int64_t total = (int64_t)O * Kpad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad), o = (int)(i / Kpad);
        float v = 0.f;
        if (k < KK * Cg) {
            const int tap = k / Cg, c = k - tap * Cg;
            v = w[((int64_t)o * Cg + c) * KK + tap];
        }
        out[i] = ElemTraits<T>::from_f(v);
    }
}

}  // namespace

extern "C" int vd3d_dcn_pack_weight(const float* w_oihw, void* packed, int O, int Cg, int kh, int kw, int Kpad, int dtype, void* stream) {
    if (!w_oihw || !packed) { vd3d_set_error("dcn_pack_weight: null pointer"); return VD3D_EINVAL; }
    const int64_t total = (int64_t)O * Kpad;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == VD3D_BF16) hipLaunchKernelGGL(dcn_pack_weight_kernel<short>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, (short*)packed, O, Cg, kh * kw, Kpad);
    else if (dtype == VD3D_F16) hipLaunchKernelGGL(dcn_pack_weight_kernel<hf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, (hf16*)packed, O, Cg, kh * kw, Kpad);
    else if (dtype == VD3D_F32) hipLaunchKernelGGL(dcn_pack_weight_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, (float*)packed, O, Cg, kh * kw, Kpad);
    else { vd3d_set_error("bad dtype"); return VD3D_EINVAL; }
    return vd3d_check_launch("dcn_pack_weight");
}

extern "C" int vd3d_deform_conv(const vd3d_dcn_params* p, void* stream) { return launch_dcn(p, (hipStream_t)stream); }

extern "C" int64_t vd3d_deform_conv_workspace_bytes(int O, int C, int groups, int kh, int kw) {
    const int Kg = kh * kw * (C / (groups > 0 ? groups : 1));
    return (int64_t)O * ((Kg + 31) / 32 * 32) * 4 + 256;
}

extern "C" int vd3d_deform_columns(const vd3d_dcn_params* p, void* columns, void* stream) {
    return launch_dcn_columns(p, columns, (hipStream_t)stream);
}

extern "C" int vd3d_deform_conv_forward(const float* input, const float* weight, const float* bias, const float* offset,
                                        const float* mask, float* output, void* workspace, int B, int C, int H, int W, int O,
                                        int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                                        int dil_h, int dil_w, int groups, int deformable_groups, void* stream) {
    if (!workspace || groups < 1 || C % groups) { vd3d_set_error("deform_conv_forward: bad workspace / groups"); return VD3D_EINVAL; }
    const int Cg = C / groups, Kpad = (kh * kw * Cg + 31) / 32 * 32;
    int rc = vd3d_dcn_pack_weight(weight, workspace, O, Cg, kh, kw, Kpad, VD3D_F32, stream);
    if (rc) return rc;
    vd3d_dcn_params q;
    q.in = input; q.weight = workspace; q.bias = bias; q.scale = nullptr; q.shift = nullptr; q.offset = offset; q.mask = mask; q.out = output;
    q.B = B; q.C = C; q.H = H; q.W = W; q.O = O; q.kh = kh; q.kw = kw;
    q.stride_h = stride_h; q.stride_w = stride_w; q.pad_h = pad_h; q.pad_w = pad_w; q.dil_h = dil_h; q.dil_w = dil_w;
    q.groups = groups; q.deformable_groups = deformable_groups; q.Kpad = Kpad; q.dtype = VD3D_F32; q.mask_sigmoid = 0; q.relu = 0;
    const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1, Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
    const int KK = kh * kw;
    // NCHW contiguous, exactly the reference extension's tensors
    q.in_strides[0] = (int64_t)C * H * W; q.in_strides[1] = (int64_t)H * W; q.in_strides[2] = W; q.in_strides[3] = 1;
    q.offset_strides[0] = (int64_t)deformable_groups * 2 * KK * Ho * Wo; q.offset_strides[1] = (int64_t)Ho * Wo; q.offset_strides[2] = Wo; q.offset_strides[3] = 1;
    q.mask_strides[0] = (int64_t)deformable_groups * KK * Ho * Wo; q.mask_strides[1] = (int64_t)Ho * Wo; q.mask_strides[2] = Wo; q.mask_strides[3] = 1;
    q.out_strides[0] = (int64_t)O * Ho * Wo; q.out_strides[1] = (int64_t)Ho * Wo; q.out_strides[2] = Wo; q.out_strides[3] = 1;
    return launch_dcn(&q, (hipStream_t)stream);
}
