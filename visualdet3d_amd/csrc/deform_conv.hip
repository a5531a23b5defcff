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
    int pk16 = 0;                            // fp16, OPT-IN (VD3D_DCN_PK16=1): blend on v_pk_fma_f16 (below); 0 = fp32 blend on v_fma_mix_f32
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

// fp16, packed blend (round 4, OPT-IN with VD3D_DCN_PK16=1): the reference's own half path blends in scalar_t = half
// (deform_conv_cuda_kernel.cu:467-497 dmcn_im2col_bilinear instantiated by AT_DISPATCH_FLOATING_TYPES_AND_HALF).  Here: the four
// modulated bilinear weights rounded ONCE to fp16 (RNE) and the four-term chain w0 x0 -> fma(w1, x1, .) -> fma(w2, x2, .) ->
// fma(w3, x3, .) on v_pk_fma_f16: TWO channels per instruction and the result is already the packed fp16 pair the MFMA operand wants --
// 16 VALU instructions per 8 channels instead of 32 v_fma_mix_f32 + 4 conversions.
// MEASURED (16 x 128 x 440 fp16, same box): 64 -> 64 386 -> 365 us (-5.5 %), 128 -> 128 219 -> 194 (-11 %), 256 -> 256 216 -> 196 (-9 %);
// config 5 8.94 -> 8.77 ms per step (-1.9 %).  PARITY: three more roundings of <= 1/2 ulp plus the weights' own rounding -- the 16
// DCNv2 blocks of config 5 teacher-forced at size read 0.54 - 1.35 x (2 fp16 ulp + 5e-4 scale) against 0.37 x for the fp32 blend: outside
// the bar the tests hold this path to, for 2 % of one configuration.  NOT the default; tests run it under the switch with a 4-ulp bar.
VD3D_DEV int pk_mul_f16(int a, int b) { int d; asm("v_pk_mul_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
VD3D_DEV int pk_fma_f16(int a, int b, int c) { int d; asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
VD3D_DEV int dcn_pk_weight(float w) { return Fmt16<hf16>::pack2_1(w, w); }
VD3D_DEV i32x4 dcn_blend8_pk(const i32x4& c0, const i32x4& c1, const i32x4& c2, const i32x4& c3, int w0, int w1, int w2, int w3) {
    i32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = pk_fma_f16(c3[d], w3, pk_fma_f16(c2[d], w2, pk_fma_f16(c1[d], w1, pk_mul_f16(c0[d], w0))));
    return o;
}

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
    // dcn_lw64_kernel: 51 of 372 us).  Any other layout (NCHW offsets of the reference's extension entry points, 27-channel tensors) keeps the old path.
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
        if (std::is_same<T, hf16>::value && p.pk16)       // packed blend: the weights as fp16 pairs (w, w), same 16 bytes of the entry
            *(i32x4*)(geo_tab + (size_t)it * GE + 16) = i32x4{dcn_pk_weight(gw[0]), dcn_pk_weight(gw[1]), dcn_pk_weight(gw[2]), dcn_pk_weight(gw[3])};
        else
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
              if (p.pk16) {
                o.raw = dcn_blend8_pk(cv[i][0], cv[i][1], cv[i][2], cv[i][3], __builtin_bit_cast(int, geo[i].w[0]), __builtin_bit_cast(int, geo[i].w[1]),
                                      __builtin_bit_cast(int, geo[i].w[2]), __builtin_bit_cast(int, geo[i].w[3]));
              } else {
                // fp16: the four-term fma chain on v_fma_mix_f32 (fp16 operand read in place, fp32 weight and accumulator)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    vals[2 * d] = mix_fma_lo(cv[i][3][d], geo[i].w[3], mix_fma_lo(cv[i][2][d], geo[i].w[2], mix_fma_lo(cv[i][1][d], geo[i].w[1], mix_mul_lo(cv[i][0][d], geo[i].w[0]))));
                    vals[2 * d + 1] = mix_fma_hi(cv[i][3][d], geo[i].w[3], mix_fma_hi(cv[i][2][d], geo[i].w[2], mix_fma_hi(cv[i][1][d], geo[i].w[1], mix_mul_hi(cv[i][0][d], geo[i].w[0]))));
                }
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
                if (!(std::is_same<T, hf16>::value && p.pk16)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.set2(e, vals[2 * e], vals[2 * e + 1]);
                }
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

// ---- 64 -> 64 channel DCNv2 3x3 / s1 / p1 (round 4, OPT-IN VD3D_DCN_LWIN=1): dcn_nhwc_kernel with its corners read from LDS ----------------
// MEASURED (16 x 128 x 440 fp16, same box): 389 us against 396 us for the gather kernel at offsets sigma 0.3, 398 / 402 at 1.5, 456 / 410 at 4.0;
// config 5 9.59 against 9.44 ms per step.  Taking 260 KB of gathers per workgroup off the L1 request path changed NOTHING: the hypothesis of
// round 2 / 3 ("bound by the 64 B / clk texture path") is refuted -- see the ablations in front of dcn_bf64_kernel.
// dcn_nhwc_kernel<T, 64> is bound by the CU's L1 / texture request path (64 B / clk): per 64-pixel workgroup and tap every thread issues
// 8 corner gathers of 16 B, i.e. 295 KB of gathers + 74 KB of weights per workgroup, 5.1 GB per launch at 16 x 128 x 440 = 150 us of
// request-path time under a 350 us kernel -- every input pixel passes through L1 ~36 times.  Two earlier attempts moved the corners to an
// LDS window (dcn_win64 / dcn_ks64 below, opt-in): both persistent, one wave or two per SIMD, and both lost to their own per-tile
// synchronisation (3.4 us of barrier / geometry / bookkeeping per 64-pixel tile with nothing else on the CU to run meanwhile).
// This kernel changes ONE thing about the gather kernel and keeps everything that hides its latencies -- one tile per workgroup, four
// waves, several workgroups per CU running out of phase, geometry table in LDS, the weight tile of a tap through registers into LDS, the
// blended columns as an LDS tile feeding 32x32x16 MFMAs, one barrier per tap:
//   * the workgroup's 64 pixels are an 8 x 8 TILE; its (8 + 2 * 3 + 1)^2 = 15 x 15-pixel input window (28.1 KB) is staged ONCE by LDS-DMA at
//     kernel start (29 pieces, under the geometry phase; out-of-image pixels arrive as zeros);
//   * a corner is one ds_read_b128 from the window (XOR-swizzled on the 16-byte slot by the window pixel: the eight lanes of a pixel read
//     its 128 contiguous bytes) -- the geometry entry holds the corner's LDS offset;
//   * a corner that lies outside the window (a learned offset beyond ~2 pixels) keeps its GLOBAL offset in the entry (sign bit set) and is
//     fetched exactly as before, per lane, behind a wave-uniform "any such corner in this wave?" test: the cost grows with the number of
//     far samples instead of falling off a cliff (config 5's five full-resolution launches: 0 - 17 % of the samples beyond 2 pixels).
// Request-path bytes per workgroup: 29 KB window + 74 KB weights + 7 KB logits instead of 370 KB.  LDS: 2 x 16 KB stages + 18 KB geometry +
// 29 KB window + constants = 79.75 KB: two workgroups per CU (the gather kernel: three).  Same blend order, same modulation fold, same k
// order on the matrix cores: BIT-IDENTICAL to dcn_nhwc_kernel (tests/test_dcn_gpu.py).
constexpr int kLwP = 3, kLwD = 8 + 2 * kLwP + 1, kLwPix = kLwD * kLwD;                   // 15 x 15 window pixels
constexpr int kLwPieces = (kLwPix * 8 + 63) / 64, kLwWin = kLwPieces * 1024;              // 29 DMA pieces of 1 KiB
constexpr int kLwStage = (64 + 64) * 128, kLwGeo = 2 * kLwStage, kLwCtab = kLwGeo + 9 * 64 * 32, kLwWinOff = kLwCtab + 3 * 64 * 4;
constexpr int kLwLds = kLwWinOff + kLwWin;                                                // 81 664 B

template <typename T>
__global__ void __launch_bounds__(256) dcn_lw64_kernel(const DcnArgs p, int tiles_x) {
    static_assert(sizeof(T) == 2, "16-bit formats only");
    constexpr int ES = 2, VE = 8, BN = 64, KK = 9, GE = 32, WV = 2;
    constexpr uint32_t kDcnOOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* geo_tab = smem + kLwGeo;                   // [tap][64 pixels] x {int32 a[4], float w[4]}
    float* ctab = (float*)(smem + kLwCtab);
    char* win = smem + kLwWinOff;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int ty0 = tyi * 8, tx0 = txi * 8;
    const int wy0 = ty0 - kLwP, wx0 = tx0 - kLwP;    // image coordinates of window pixel (0, 0)
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.in + (int64_t)b * p.in_sb * ES), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 64 * p.Kpad * ES, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    // ---- the window: piece q = wave + 4 it, chunk c = 64 q + lane -> window pixel pp = c >> 3, physical slot c & 7 holds the logical
    // vector (slot ^ key(pp)), key(pp) = (pp >> 1) & 7 (the swizzle is applied to the SOURCE address: the DMA writes lane-linear)
#pragma unroll
    for (int it = 0; it < (kLwPieces + 3) / 4; ++it) {
        const int q = wave + 4 * it;
        if (q < kLwPieces) {                         // wave-uniform
            const int c = q * 64 + lane, pp = c >> 3, v = (c & 7) ^ ((pp >> 1) & 7);
            const int wy = pp / kLwD, wx = pp - wy * kLwD;
            const int iy = wy0 + wy, ix = wx0 + wx;
            const bool ok = pp < kLwPix && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t off = ok ? (uint32_t)((iy * p.in_sy + ix * p.in_sx + v * VE) * ES) : kDcnOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(win + q * 1024), 16, off, 0, 0, 0);
        }
    }
    for (int i = tid; i < BN; i += 256) {
        ctab[i] = p.bias ? p.bias[i] : 0.f;
        ctab[BN + i] = p.scale ? p.scale[i] : 1.f;
        ctab[2 * BN + i] = p.shift ? p.shift[i] : 0.f;
    }
    // ---- phase 0: the sampling geometry of every (pixel, tap) of the tile, once, into LDS (as dcn_nhwc_kernel: a thread owns pixel `lane`
    // and every fourth tap; the logits of three taps are requested up front)
    {
    const int px = lane;
    const int oy = ty0 + (px >> 3), ox = tx0 + (px & 7);
    const bool pvalid = oy < p.Ho && ox < p.Wo;
    const int64_t ob = b * p.off_sb + oy * p.off_sy + ox * p.off_sx;
    const int64_t mb = b * p.msk_sb + oy * p.msk_sy + ox * p.msk_sx;
    constexpr int TPW = 3;
    float oh[TPW], ow[TPW], ml[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int tap = wave + 4 * u;
        oh[u] = ow[u] = ml[u] = 0.f;
        if (tap < KK && pvalid) {
            oh[u] = p.offset[ob + (int64_t)(2 * tap) * p.off_sc];
            ow[u] = p.offset[ob + (int64_t)(2 * tap + 1) * p.off_sc];
            if (p.mask) ml[u] = p.mask[mb + (int64_t)tap * p.msk_sc];
        }
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int tap = wave + 4 * u;
        if (tap >= KK) break;
        int ga[4] = {0, 0, 0, 0};                    // LDS offset 0: finite data, weight 0
        float gw[4] = {0.f, 0.f, 0.f, 0.f};
        if (pvalid) {
            const int ti = tap / 3, tj = tap - ti * 3;
            const float h_im = (float)(oy - 1 + ti) + oh[u];
            const float w_im = (float)(ox - 1 + tj) + ow[u];
            if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                float m = 1.f;
                if (p.mask) {
                    m = ml[u];
                    if (p.mask_sigmoid) m = __frcp_rn(1.0f + __expf(-m));
                }
                const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float cw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int hc = h_low + (c >> 1), wc = w_low + (c & 1);
                    if ((unsigned)hc < (unsigned)p.H && (unsigned)wc < (unsigned)p.W) {
                        gw[c] = cw[c] * m;           // the modulation folded into the weights (16-bit formats), as dcn_nhwc_kernel
                        const int wy = hc - wy0, wx = wc - wx0;
                        if ((unsigned)wy < (unsigned)kLwD && (unsigned)wx < (unsigned)kLwD) {
                            const int pp = wy * kLwD + wx;
                            ga[c] = pp * 128 + (((pp >> 1) & 7) << 4);
                        } else {
                            ga[c] = (int)(0x80000000u | ((uint32_t)((hc * p.in_sy + wc * p.in_sx) * ES) >> 4));   // global: byte offset / 16
                        }
                    }
                }
            }
        }
        const int it = tap * 64 + px;
        *(i32x4*)(geo_tab + (size_t)it * GE) = i32x4{ga[0], ga[1], ga[2], ga[3]};
        *(f32x4*)(geo_tab + (size_t)it * GE + 16) = f32x4{gw[0], gw[1], gw[2], gw[3]};
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's window pieces have landed
    __syncthreads();                                        // ... everybody's; the geometry table is complete

    // Sampling map of the K loop (as dcn_nhwc_kernel): 8 consecutive lanes own the 8 16-byte vectors of ONE pixel's 128-byte channel run;
    // a thread handles pixels prow0 and prow0 + 32, vector vslot
    const int prow0 = tid >> 3, vslot = tid & 7, vs = vslot << 4;
    struct Geo { int32_t a[4]; float w[4]; };
    auto geometry = [&](int tap, int prow) {
        Geo g;
        const char* e = geo_tab + (size_t)(tap * 64 + prow) * GE;
        const i32x4 ga = *(const i32x4*)e;
        const f32x4 gw = *(const f32x4*)(e + 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) { g.a[c] = ga[c]; g.w[c] = gw[c]; }
        return g;
    };
    int w_voff[WV];
#pragma unroll
    for (int i = 0; i < WV; ++i) {
        const int v = tid + 256 * i;
        w_voff[i] = ((v >> 3) * p.Kpad + (v & 7) * VE) * ES;
    }
    Geo geo[2];
    auto fetch = [&](int kt, i32x4 (&cv)[2][4], i32x4 (&wv)[WV]) {
        geo[0] = geometry(kt, prow0);
        geo[1] = geometry(kt, prow0 + 32);
#pragma unroll
        for (int i = 0; i < WV; ++i)
            wv[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff[i], kt * 128, 0));
        int far = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int a = geo[i].a[c];
                far |= a;
                cv[i][c] = *(const i32x4*)(win + (a < 0 ? 0 : (a ^ vs)));
            }
        if (__builtin_amdgcn_ballot_w64(far < 0) != 0) {       // wave-uniform: some corner of this wave's samples lies outside the window
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int a = geo[i].a[c];
                    if (a < 0) cv[i][c] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)(((uint32_t)a & 0x7fffffffu) << 4) + vs, 0, 0));
                }
        }
    };
    // registers -> LDS stage: blend the four corners (fp32, the gather kernel's order), round, store swizzled
    auto stash = [&](int st, const i32x4 (&cv)[2][4], const i32x4 (&wv)[WV]) {
        char* Cs = smem + st * kLwStage;
        char* Ws = Cs + 64 * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            Vec16<T> c1, c2, c3, c4, o;
            c1.raw = cv[i][0]; c2.raw = cv[i][1]; c3.raw = cv[i][2]; c4.raw = cv[i][3];
            float vals[VE];
            bool packed = false;
            if constexpr (std::is_same<T, hf16>::value) {
                if (p.pk16) {
                    o.raw = dcn_blend8_pk(cv[i][0], cv[i][1], cv[i][2], cv[i][3], dcn_pk_weight(geo[i].w[0]), dcn_pk_weight(geo[i].w[1]),
                                          dcn_pk_weight(geo[i].w[2]), dcn_pk_weight(geo[i].w[3]));
                    packed = true;
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        vals[2 * d] = mix_fma_lo(cv[i][3][d], geo[i].w[3], mix_fma_lo(cv[i][2][d], geo[i].w[2], mix_fma_lo(cv[i][1][d], geo[i].w[1], mix_mul_lo(cv[i][0][d], geo[i].w[0]))));
                        vals[2 * d + 1] = mix_fma_hi(cv[i][3][d], geo[i].w[3], mix_fma_hi(cv[i][2][d], geo[i].w[2], mix_fma_hi(cv[i][1][d], geo[i].w[1], mix_mul_hi(cv[i][0][d], geo[i].w[0]))));
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < VE; ++e)
                    vals[e] = fmaf(geo[i].w[3], c4.get(e), fmaf(geo[i].w[2], c3.get(e), fmaf(geo[i].w[1], c2.get(e), geo[i].w[0] * c1.get(e))));
            }
            if (!packed) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o.set2(e, vals[2 * e], vals[2 * e + 1]);
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

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int wn = wave & 1, wm = wave >> 1;         // 2 x 2 waves over (out-channel halves, pixel halves)
    const int lr = lane & 31, half = lane >> 5;
    i32x4 cv[2][4], wv[WV];
    fetch(0, cv, wv);
    stash(0, cv, wv);
    __syncthreads();
    for (int kt = 0; kt < KK; ++kt) {
        const int st = kt & 1;
        const bool more = kt + 1 < KK;
        if (more) fetch(kt + 1, cv, wv);
        const char* Cs = smem + st * kLwStage;
        const char* Ws = Cs + 64 * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int sk = 2 * ks + half;
            const int rc = wm * 32 + lr, rw = wn * 32 + lr;
            const i32x4 fb = *(const i32x4*)(Cs + rc * 128 + ((sk ^ ((rc >> 1) & 7)) << 4));
            const i32x4 fa = *(const i32x4*)(Ws + rw * 128 + ((sk ^ ((rw >> 1) & 7)) << 4));
            DMma<T>::run(fa, fb, acc);
        }
        if (more) stash(st ^ 1, cv, wv);
        __syncthreads();
    }
    // ---- epilogue: bias, folded BN, ReLU; 4 consecutive channels per accumulator quad -> 8-byte NHWC stores ----
    const int px = wm * 32 + lr;
    const int oy = ty0 + (px >> 3), ox = tx0 + (px & 7);
    if (oy >= p.Ho || ox >= p.Wo) return;
    const int64_t ob = b * p.out_sb + oy * p.out_sy + ox * p.out_sx;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int lc = wn * 32 + 8 * g + 4 * half;
        const f32x4 bz = *(const f32x4*)(ctab + lc), sc = *(const f32x4*)(ctab + BN + lc), sh = *(const f32x4*)(ctab + 2 * BN + lc);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = (acc[4 * g + e] + bz[e]) * sc[e] + sh[e];
            if (p.relu) x = fmaxf(x, 0.f);
            v[e] = x;
        }
        i32x2 o2;
        o2[0] = Fmt16<T>::pack2(v[0], v[1]);
        o2[1] = Fmt16<T>::pack2(v[2], v[3]);
        *(i32x2*)((char*)p.out + (ob + lc) * 2) = o2;
    }
}

// ---- 64 -> 64 channel DCNv2 3x3 / s1 / p1, barrier-free K loop (round 4, OPT-IN VD3D_DCN_BF=1) ---------------------------------------------
// Timing ablations of dcn_lw64_kernel above (16 x 128 x 440 fp16, 372 us; results wrong by construction, removed again): no corner reads
// -25 us, no weight loads -24, no blend -40, no logit loads -51, no MFMA -23, ALL of them off: still 232 us -- the skeleton of "one tap at a
// time: registers -> LDS column tile -> barrier -> fragments -> 4 MFMAs per wave" (ten barriers and ~0.3 MB of LDS staging traffic per 64-pixel
// workgroup, two or three workgroups per CU) is 62 % of the kernel, whichever way the corners arrive.  This kernel has ONE barrier:
//   * a workgroup owns an 8 x 16 tile; window (15 x 23 pixels, 43 KB, LDS-DMA) and geometry table (128 pixels x 9 taps x 32 B) as above;
//   * a wave owns 32 pixels (two rows of the tile) x ALL 64 output channels on v_mfma_f32_16x16x32: lane (pixel n, k-group g) blends exactly
//     the 8 channels it multiplies -- the blended vector IS the B fragment, no column tile, no barrier, the waves of a workgroup never meet
//     again after the geometry phase (the operand mapping of dcn_win64_kernel);
//   * the A fragments (weights, 16 output channels x 32 k) come straight from the packed [O][Kpad] matrix through the vector cache: one
//     16-byte load per lane, shared by the wave's two pixel blocks; every wave streams the 72 KB panel once per 32 pixels (2.3 KB per
//     pixel through the request path, 2.6 with the window, against 5.8 in the gather kernel);
//   * corners outside the window: per lane from global memory, as above; outputs leave as whole 128-byte lines through the wave's own (dead)
//     slice of the geometry table.
// Same blend, same fold, same k order: bit-identical to the gather kernel.  LDS 81 664 B: two workgroups (8 waves) per CU.
// MEASURED (16 x 128 x 440 fp16, same box, gather kernel 372 / 390 / 402 us at offsets sigma 0.3 / 1.5 / 4.0): first version (nine taps fully
// unrolled, logits as fifteen strided 4-byte loads per thread) 351 / 384 / 494 us; + rolled tap loop (the unrolled kernel was 80 KB of code), coalesced
// logits, software-pipelined taps: 345 / 388 us; config 5 9.21 against 9.06 ms per step -- NOT faster inside the model, hence opt-in: under rocprofv3 the
// model's five 64 -> 64 launches take 319 - 331 us on the gather kernel and 331 - 384 us here (profiles/r04_c5_timeline_*.txt) -- the model's offsets are smooth, so the
// gathers of neighbouring pixels hit the same cache lines; the micro-benchmark's independent random offsets (372 us) flatter every window design.  Cycle stamps of
// one workgroup (tools/dcn_stamps.py, -DVD3D_STAMPS): 50k cycles = logits + window issue 5k, geometry 12-13k, drain + barrier 6k, nine taps 2.0-2.4k each
// (21k; 2-4k before the pipelining), epilogue 4k.  Instruction count: ~12k wave-instructions of VALU per 128-pixel workgroup (geometry ~110 per
// (pixel, tap), run on 5 of 8 lanes; blend 128 v_fma_mix + 16 conversions per tap and lane) = 157 us per launch at 100 % VALU issue: every variant of
// this operator -- gather, LDS window, K-split, barrier-free -- sits at 40-45 % of THAT bound with 8-12 waves per CU.  What is left: a geometry fast
// path (all four corners inside image and window: one test instead of four, ~50 instead of ~110 instructions), the ninth tap spread over the idle
// lanes, and the packed fp16 blend (halves the K loop's VALU; outside the 2-ulp bar, see dcn_blend8_pk).
#ifdef VD3D_STAMPS
__device__ unsigned long long g_stamps[4][32];
#define STAMP(i) do { if (blockIdx.x == 200 && blockIdx.z == 5 && lane == 0) g_stamps[wave][i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
constexpr int kBfTH = 8, kBfTW = 16, kBfWR = kBfTH + 2 * kLwP + 1, kBfWC = kBfTW + 2 * kLwP + 1;     // window 15 rows x 23 columns
constexpr int kBfPieces = 43, kBfWinPix = kBfPieces * 8, kBfWin = kBfPieces * 1024;                     // 344 of the 345 pixels (the last one: "far")
constexpr int kBfGeoWave = 9 * 32 * 32, kBfGeo = 4 * kBfGeoWave;                                        // [wave][tap][32 pixels] x 32 B
constexpr int kBfWinOff = kBfGeo, kBfLds = kBfWinOff + kBfWin + 768;                                    // 81 664 B

template <typename T, bool PK>
__global__ void __launch_bounds__(256) dcn_bf64_kernel(const DcnArgs p, int tiles_x) {
    static_assert(sizeof(T) == 2, "16-bit formats only");
    constexpr int ES = 2, KK = 9;
    constexpr uint32_t kDcnOOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* geo_tab = smem;
    char* win = smem + kBfWinOff;
    float* ctab = (float*)(smem + kBfWinOff + kBfWin);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int ty0 = tyi * kBfTH, tx0 = txi * kBfTW;
    const int wy0 = ty0 - kLwP, wx0 = tx0 - kLwP;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.in + (int64_t)b * p.in_sb * ES), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 64 * p.Kpad * ES, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    STAMP(0);

    // One geometry entry: (pixel of the tile, tap, its three logits) -> {LDS offset | global offset of the four corners, modulated weights}
    auto write_entry = [&](int prow, int pcol, int tap, bool pvalid, float off_h, float off_w, float mlog) {
        int ga[4] = {0, 0, 0, 0};                    // LDS offset 0: finite data, weight 0
        float gw[4] = {0.f, 0.f, 0.f, 0.f};
        if (pvalid) {
            const int ti = tap / 3, tj = tap - ti * 3;
            const float h_im = (float)(ty0 + prow - 1 + ti) + off_h;
            const float w_im = (float)(tx0 + pcol - 1 + tj) + off_w;
            if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                float m = 1.f;
                if (p.mask) {
                    m = mlog;
                    if (p.mask_sigmoid) m = __frcp_rn(1.0f + __expf(-m));
                }
                const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float cw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int hc = h_low + (c >> 1), wc = w_low + (c & 1);
                    if ((unsigned)hc < (unsigned)p.H && (unsigned)wc < (unsigned)p.W) {
                        gw[c] = cw[c] * m;           // the modulation folded into the weights (16-bit formats), as dcn_nhwc_kernel
                        const int wy = hc - wy0, wx = wc - wx0;
                        const int pp = wy * kBfWC + wx;
                        if ((unsigned)wy < (unsigned)kBfWR && (unsigned)wx < (unsigned)kBfWC && pp < kBfWinPix)
                            ga[c] = pp * 128 + (((pp >> 1) & 7) << 4);
                        else
                            ga[c] = (int)(0x80000000u | ((uint32_t)((hc * p.in_sy + wc * p.in_sx) * ES) >> 4));   // global: byte offset / 16
                    }
                }
            }
        }
        // entry of (wave prow >> 1, tap, pixel (prow & 1) * 16 + pcol)
        char* e = geo_tab + (prow >> 1) * kBfGeoWave + (tap * 32 + (prow & 1) * 16 + pcol) * 32;
        *(i32x4*)e = i32x4{ga[0], ga[1], ga[2], ga[3]};
        *(f32x4*)(e + 16) = f32x4{gw[0], gw[1], gw[2], gw[3]};
    };
    // The logits of the tile.  Fast layout (what the offset conv of the engine writes: one pixel = 32 contiguous fp32 = offsets 0..17 | mask
    // 18..26 | pad, 16-byte aligned): FOUR fully coalesced 16-byte loads per thread (8 lanes = one pixel's 128 bytes) instead of fifteen
    // 4-byte loads at a 128-byte stride (64 cache lines per instruction: the geometry phase was 13k of a workgroup's 54k cycles, cycle stamps).
    const bool packed_logits = p.mask && p.off_sc == 1 && p.msk_sc == 1 && p.mask == p.offset + 18 && p.msk_sb == p.off_sb && p.msk_sy == p.off_sy &&
                               p.msk_sx == p.off_sx && (p.off_sx & 3) == 0 && (p.off_sy & 3) == 0 && (p.off_sb & 3) == 0 && p.off_sx >= 28 &&
                               ((uintptr_t)p.offset & 15) == 0;
    f32x4 lg[4];
    float oh[5], ow[5], ml[5];
    if (packed_logits) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int px = (tid >> 3) + 32 * r, v = tid & 7;
            const int oy = ty0 + (px >> 4), ox = tx0 + (px & 15);
            lg[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (oy < p.Ho && ox < p.Wo && v < 7) lg[r] = *(const f32x4*)(p.offset + b * p.off_sb + oy * p.off_sy + ox * p.off_sx + 4 * v);
        }
    } else {
        const int px = tid & 127, par = tid >> 7;
        const int oy = ty0 + (px >> 4), ox = tx0 + (px & 15);
        const bool pvalid = oy < p.Ho && ox < p.Wo;
        const int64_t ob = b * p.off_sb + oy * p.off_sy + ox * p.off_sx;
        const int64_t mb = b * p.msk_sb + oy * p.msk_sy + ox * p.msk_sx;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int tap = par + 2 * u;
            oh[u] = ow[u] = ml[u] = 0.f;
            if (tap < KK && pvalid) {
                oh[u] = p.offset[ob + (int64_t)(2 * tap) * p.off_sc];
                ow[u] = p.offset[ob + (int64_t)(2 * tap + 1) * p.off_sc];
                if (p.mask) ml[u] = p.mask[mb + (int64_t)tap * p.msk_sc];
            }
        }
    }
    STAMP(1);
    // ---- the window (as dcn_lw64_kernel: physical slot c & 7 of window pixel pp holds the logical vector (c & 7) ^ key(pp)); issued while
    // the logits travel ----
#pragma unroll
    for (int it = 0; it < (kBfPieces + 3) / 4; ++it) {
        const int q = wave + 4 * it;
        if (q < kBfPieces) {
            const int c = q * 64 + lane, pp = c >> 3, v = (c & 7) ^ ((pp >> 1) & 7);
            const int wy = pp / kBfWC, wx = pp - wy * kBfWC;
            const int iy = wy0 + wy, ix = wx0 + wx;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t off = ok ? (uint32_t)((iy * p.in_sy + ix * p.in_sx + v * 8) * ES) : kDcnOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(win + q * 1024), 16, off, 0, 0, 0);
        }
    }
    if (tid < 64) {
        ctab[tid] = p.bias ? p.bias[tid] : 0.f;
        ctab[64 + tid] = p.scale ? p.scale[tid] : 1.f;
        ctab[128 + tid] = p.shift ? p.shift[tid] : 0.f;
    }
    STAMP(2);
    if (packed_logits) {
        // lane v (0..4) of a pixel's 8-lane group owns taps 2v and 2v + 1: their offsets are components (0, 1) / (2, 3) of its own vector, their
        // mask logits (18 + tap) sit in the vector of lane 4 + (v + 1) / 2 of the same group, components (2, 3) for even v, (0, 1) for odd v
        const int v = tid & 7;
        const int src = ((lane & ~7) + 4 + ((v + 1) >> 1)) << 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int px = (tid >> 3) + 32 * r;
            const int prow = px >> 4, pcol = px & 15;
            const bool pvalid = ty0 + prow < p.Ho && tx0 + pcol < p.Wo;
            float x[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = i2f(__builtin_amdgcn_ds_bpermute(src, f2i(lg[r][c])));
            const float m0 = (v & 1) ? x[0] : x[2], m1 = (v & 1) ? x[1] : x[3];
            if (v < 5) write_entry(prow, pcol, 2 * v, pvalid, lg[r][0], lg[r][1], m0);
            if (v < 4) write_entry(prow, pcol, 2 * v + 1, pvalid, lg[r][2], lg[r][3], m1);
        }
    } else {
        const int px = tid & 127, par = tid >> 7;
        const int prow = px >> 4, pcol = px & 15;
        const bool pvalid = ty0 + prow < p.Ho && tx0 + pcol < p.Wo;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int tap = par + 2 * u;
            if (tap < KK) write_entry(prow, pcol, tap, pvalid, oh[u], ow[u], ml[u]);
        }
    }
    STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(4);
    __syncthreads();                                  // the ONLY barrier: window, geometry table and constants are in LDS
    STAMP(5);

    // ---- K loop: tap-major, two 32-channel k-steps per tap; lane (n, g): pixel n of a 16-pixel block, k-group g ----
    const int n = lane & 15, g = lane >> 4;
    const char* G0 = geo_tab + wave * kBfGeoWave;
    const int wlane = ((lane & 15) * p.Kpad + g * 8) * ES;        // A fragment: row blk * 16 + (lane & 15), k = tap * 64 + s * 32 + g * 8 ..
    const int wblk = 16 * p.Kpad * ES;
    f32x4 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) acc[j][blk] = f32x4{0.f, 0.f, 0.f, 0.f};
    // does ANY sample of this wave's 32 pixels leave the window?  (wave-uniform; the common case runs without a branch per corner)
    bool wave_far = false;
    {
        int f = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int e = r * 64 + lane;                       // 288 entries of this wave
            if (e < 288) {
                const i32x4 a = *(const i32x4*)(G0 + e * 32);
                f |= a[0] | a[1] | a[2] | a[3];
            }
        }
        wave_far = __builtin_amdgcn_ballot_w64(f < 0) != 0;
    }
    STAMP(6);
    // The taps as a ROLLED loop, two taps per trip (fully unrolled with both corner paths and both blends the kernel was 9.8k instructions,
    // 80 KB: more than the instruction cache two CUs share), SOFTWARE-PIPELINED by hand: nothing a (k-step, pixel block) unit consumes is
    // requested inside that unit -- the A fragments and the geometry entries of tap t + 1 are requested at the top of tap t, the four corner
    // vectors of unit u + 1 before the blend of unit u (cycle stamps of the first version: 2.1 - 4.1k cycles per tap against ~0.7k of VALU
    // issue: every unit waited for its own LDS reads and every tap for its own weight loads).
    auto taps = [&](auto slow_c) {
    constexpr bool SLOW = decltype(slow_c)::value;
    struct TapRegs { i32x4 ga[2]; f32x4 gw[2]; i32x4 fa[2][4]; };
    auto load_tap = [&](int tap, TapRegs& r) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            r.ga[j] = *(const i32x4*)(G0 + (tap * 32 + j * 16 + n) * 32);
            r.gw[j] = *(const f32x4*)(G0 + (tap * 32 + j * 16 + n) * 32 + 16);
        }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
                r.fa[sx][blk] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wlane + blk * wblk, (tap * 64 + sx * 32) * ES, 0));
    };
    auto corners = [&](const i32x4& ga, int sx, i32x4 (&cv)[4]) {
        const int vs = (sx * 4 + g) << 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int a = ga[c];
            if constexpr (SLOW) cv[c] = *(const i32x4*)(win + (a < 0 ? 0 : (a ^ vs)));
            else cv[c] = *(const i32x4*)(win + (a ^ vs));
        }
        if constexpr (SLOW) {
            if (__builtin_amdgcn_ballot_w64((ga[0] | ga[1] | ga[2] | ga[3]) < 0) != 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int a = ga[c];
                    if (a < 0) cv[c] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)(((uint32_t)a & 0x7fffffffu) << 4) + vs, 0, 0));
                }
            }
        }
    };
    auto blend_mfma = [&](const i32x4 (&cv)[4], const f32x4& gw, const i32x4 (&fa)[4], f32x4 (&ac)[4]) {
        Vec16<T> o;
        if constexpr (PK) {                       // (compile time: a run-time branch per unit would fence the scheduler in)
            o.raw = dcn_blend8_pk(cv[0], cv[1], cv[2], cv[3], dcn_pk_weight(gw[0]), dcn_pk_weight(gw[1]), dcn_pk_weight(gw[2]), dcn_pk_weight(gw[3]));
        } else {
            float vals[8];
            if constexpr (std::is_same<T, hf16>::value) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    vals[2 * d] = mix_fma_lo(cv[3][d], gw[3], mix_fma_lo(cv[2][d], gw[2], mix_fma_lo(cv[1][d], gw[1], mix_mul_lo(cv[0][d], gw[0]))));
                    vals[2 * d + 1] = mix_fma_hi(cv[3][d], gw[3], mix_fma_hi(cv[2][d], gw[2], mix_fma_hi(cv[1][d], gw[1], mix_mul_hi(cv[0][d], gw[0]))));
                }
            } else {
                Vec16<T> c1, c2, c3, c4;
                c1.raw = cv[0]; c2.raw = cv[1]; c3.raw = cv[2]; c4.raw = cv[3];
#pragma unroll
                for (int e = 0; e < 8; ++e) vals[e] = fmaf(gw[3], c4.get(e), fmaf(gw[2], c3.get(e), fmaf(gw[1], c2.get(e), gw[0] * c1.get(e))));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, vals[2 * e], vals[2 * e + 1]);
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) Fmt16<T>::mfma16(fa[blk], o.raw, ac[blk]);
    };
    // one tap: units u = 0..3 = (k-step u >> 1, pixel block u & 1); `first` = corners of unit 0 (already requested), `nxt` = the next tap's
    // registers (requested at the top), whose unit-0 corners are requested before this tap's last blend and returned in `first`
    auto one_tap = [&](const TapRegs& cur, const TapRegs& nxt, i32x4 (&first)[4]) {
        i32x4 cva[4], cvb[4];
        corners(cur.ga[1], 0, cva);                                  // unit 1
        blend_mfma(first, cur.gw[0], cur.fa[0], acc[0]);             // unit 0
        corners(cur.ga[0], 1, cvb);                                  // unit 2
        blend_mfma(cva, cur.gw[1], cur.fa[0], acc[1]);               // unit 1
        corners(cur.ga[1], 1, cva);                                  // unit 3
        blend_mfma(cvb, cur.gw[0], cur.fa[1], acc[0]);               // unit 2
        corners(nxt.ga[0], 0, first);                                // unit 0 of the next tap
        blend_mfma(cva, cur.gw[1], cur.fa[1], acc[1]);               // unit 3
    };
    TapRegs ra, rb;
    i32x4 first[4];
    load_tap(0, ra);
    corners(ra.ga[0], 0, first);
#pragma unroll 1
    for (int tap = 0; tap < KK - 1; tap += 2) {                      // taps 0 .. 7 in pairs (a -> b -> a); tap 8 below
        load_tap(tap + 1, rb);
        one_tap(ra, rb, first);
        STAMP(7 + tap);
        load_tap(tap + 2, ra);
        one_tap(rb, ra, first);
        STAMP(8 + tap);
    }
    one_tap(ra, ra, first);                                          // (its look-ahead re-reads tap 8's own first corners: never consumed)
    STAMP(15);
    };
    if (wave_far) taps(std::true_type{});
    else taps(std::false_type{});
    // ---- epilogue: bias, folded BN, ReLU; lane (n, g) holds channels blk * 16 + g * 4 + e of pixel n.  The wave's 32 pixels x 64 channels
    // (4 KB) are parked in its own slice of the geometry table (dead: only this wave ever read it) and leave as whole 128-byte lines ----
    char* park = geo_tab + wave * kBfGeoWave;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int lc = blk * 16 + g * 4;
            const f32x4 bz = *(const f32x4*)(ctab + lc), sc = *(const f32x4*)(ctab + 64 + lc), sh = *(const f32x4*)(ctab + 128 + lc);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = (acc[j][blk][e] + bz[e]) * sc[e] + sh[e];
                if (p.relu) x = fmaxf(x, 0.f);
                v[e] = x;
            }
            i32x2 o2;
            o2[0] = Fmt16<T>::pack2(v[0], v[1]);
            o2[1] = Fmt16<T>::pack2(v[2], v[3]);
            *(i32x2*)(park + (j * 16 + n) * 128 + lc * 2) = o2;
        }
    STAMP(16);
    // (same wave wrote and reads: program order + the compiler's lgkmcnt is all the synchronisation there is to do)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pl = r * 8 + (lane >> 3), vsl = lane & 7;          // pixel (j = pl >> 4, column pl & 15), 16-byte vector vsl
        const i32x4 o = *(const i32x4*)(park + pl * 128 + vsl * 16);
        const int oy = ty0 + 2 * wave + (pl >> 4), ox = tx0 + (pl & 15);
        if (oy < p.Ho && ox < p.Wo)
            *(i32x4*)((char*)p.out + ((int64_t)b * p.out_sb + (int64_t)oy * p.out_sy + (int64_t)ox * p.out_sx + vsl * 8) * 2) = o;
    }
    STAMP(17);
}

// ---- 64 -> 64 channel DCNv2 3x3 / s1 / p1: corners gathered from an LDS-staged input window, weights resident in LDS ---------------
// The kernel above is bound by the L1 request path for C = O = 64 (the ten full-resolution launches of KM3D's DLA-Up, 2.2 ms of a
// 9.8 ms step at 8.8 % of the MFMA peak): per 64-pixel workgroup and tap every thread issues 8 corner gathers + 2 weight vectors of
// 1 KiB per wave, every needed input pixel passes through L1 ~36 times, and every workgroup streams the whole 74 KB weight panel
// (1 GB per launch at 16 x 128 x 440).  Here:
//   * PERSISTENT workgroups (one per CU, 4 waves); the 64 x 576 weights are loaded ONCE per workgroup into LDS as MFMA A-fragment
//     images (72 x 1 KiB, lane-linear: conflict-free ds_read_b128);
//   * a tile is 8 x 8 output pixels; its input WINDOW of (8 + 2 * 3 + 1)^2 = 15 x 15 pixels x 128 B is staged once by LDS-DMA
//     (out-of-image pixels arrive as zeros = the reference's "corner outside the image contributes 0"), double buffered: tile k+1's
//     window and offset logits travel while tile k computes; the four corners of a sample are four ds_read_b128 (256 B/clk/CU
//     instead of the 64 B/clk L1 path), XOR-swizzled on the 16-byte slot by the window pixel;
//   * a wave owns 16 pixels x all 64 output channels on v_mfma_f32_16x16x32: the blended values ARE the B fragment (lane (pixel,
//     k-group) blends the 2 x 8 channels it multiplies) -- no column tile in LDS, no barrier inside a tile, ONE barrier per tile;
//   * samples whose corners leave the window (learned offsets beyond +-2 px) take the global-gather path per (wave, tap): same
//     arithmetic, only slower -- results are bit-identical to dcn_nhwc_kernel in every case (same blend order, same modulation fold);
//   * outputs leave as whole 128-byte lines (parked per wave in its own geometry slot of LDS).
// MEASURED (round 3, 16 x 128 x 440 fp16, offsets sigma 0.5 px): 419 us against 354 us for dcn_nhwc_kernel -- the L1 path is gone, but
// with 151 KB of LDS per workgroup only ONE wave runs per SIMD and nothing hides its LDS / VALU latencies (the blend alone is ~1050
// VALU instructions per tile and wave = a 113 us floor for this launch; the gather kernel keeps 12 waves per CU busy at ~32 % VALU
// utilisation).  The kernel is therefore NOT the default: it runs only with VD3D_DCN_WINDOW=1 (tests exercise it that way and hold it
// bit-identical to the gather kernel).  What would make it pay is two waves per SIMD: weights split over K into registers (8 waves,
// 144 VGPRs each, partial sums reduced in LDS) instead of 74 KB of LDS -- DESIGN.md section 9.
constexpr int kWinP = 3, kWinD = 8 + 2 * kWinP + 1, kWinPix = kWinD * kWinD;            // 15 x 15 window pixels
constexpr int kWinPieces = (kWinPix * 8 + 63) / 64, kWinBytes = kWinPieces * 1024;       // 29 DMA pieces of 1 KiB
constexpr int kWinWts = 9 * 2 * 4 * 1024;                                                // (tap, k-step, 16-channel block) fragments
constexpr int kWinGeoWave = 9 * 16 * 16, kWinGeo = 4 * kWinGeoWave;                      // per wave: 9 taps x 16 pixels x 16 B
constexpr int kWinLds = kWinWts + 2 * kWinBytes + 2 * kWinGeo;                           // 151 552 B

template <typename T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) dcn_win64_kernel(const DcnArgs p, int ntiles, int tiles_x, int tiles_y) {
    static_assert(sizeof(T) == 2, "16-bit formats only");
    constexpr uint32_t kOOBw = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wts = smem;
    char* win = smem + kWinWts;
    char* geo = win + 2 * kWinBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int prow = 2 * wave + (n >> 3), pcol = n & 7;                 // this lane's pixel inside the tile
    // XCD-aware persistent walk: the workgroups of one XCD (private L2) take a contiguous run of tiles, neighbours share window halos
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_xcd = nwg >> 3;
    const int chunk = (ntiles + 7) >> 3;
    auto tile_of = [&](int k) { const int i = jx + k * per_xcd; return i < chunk ? xcd * chunk + i : ntiles; };
    const int tiles_img = tiles_x * tiles_y;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0x7fffffff, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    // ---- weights -> LDS fragment images (once per workgroup): fragment f = (tap*2 + s)*4 + blk, lane l: row blk*16 + (l & 15),
    // k = tap*64 + s*32 + (l >> 4)*8 .. +7
    for (int idx = tid; idx < 72 * 64; idx += 256) {
        const int f = idx >> 6, l = idx & 63;
        const int tap = f >> 3, sx = (f >> 2) & 1, blk = f & 3;
        const int o = blk * 16 + (l & 15), kk = tap * 64 + sx * 32 + (l >> 4) * 8;
        *(i32x4*)(wts + idx * 16) = *(const i32x4*)((const char*)p.w + ((size_t)o * p.Kpad + kk) * 2);
    }
    // per-channel constants of this lane's 16 output channels (blk*16 + g*4 + e)
    float cb[4][4], cs[4][4], ct[4][4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = blk * 16 + g * 4 + e;
            cb[blk][e] = p.bias ? p.bias[o] : 0.f;
            cs[blk][e] = p.scale ? p.scale[o] : 1.f;
            ct[blk][e] = p.shift ? p.shift[o] : 0.f;
        }
    // DMA lane constants: piece q = wave + 4*it, chunk c = 64 q + lane -> window pixel pp = c >> 3, LDS slot c & 7 holds vector
    // (slot ^ key(pp)) of that pixel (the swizzle is applied to the SOURCE: the DMA writes lane-linear)
    constexpr int NIT = (kWinPieces + 3) / 4;
    int d_rel[NIT], d_yx[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = wave + 4 * it, c = q * 64 + lane, pp = c >> 3, sl = c & 7;
        const int v = sl ^ ((pp >> 1) & 7);
        const int wy = pp / kWinD, wx = pp - wy * kWinD;
        d_rel[it] = (int)((wy * p.in_sy + wx * p.in_sx + v * 8) * 2);
        d_yx[it] = (q < kWinPieces && pp < kWinPix) ? (wy | (wx << 8)) : -1;
    }
    auto decode = [&](int t, int& b, int& ty0, int& tx0) {
        b = t / tiles_img;
        const int r = t - b * tiles_img, ty = r / tiles_x;
        ty0 = ty * 8;
        tx0 = (r - ty * tiles_x) * 8;
    };
    auto issue_window = [&](int t, int buf) {
        int b, ty0, tx0;
        decode(t, b, ty0, tx0);
        const int base = (int)((b * p.in_sb + (ty0 - kWinP) * p.in_sy + (tx0 - kWinP) * p.in_sx) * 2);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = wave + 4 * it;
            if (q < kWinPieces) {                                   // wave-uniform
                const int wy = d_yx[it] & 255, wx = (d_yx[it] >> 8) & 255;
                const bool ok = d_yx[it] >= 0 && (unsigned)(ty0 - kWinP + wy) < (unsigned)p.H && (unsigned)(tx0 - kWinP + wx) < (unsigned)p.W;
                const uint32_t off = ok ? (uint32_t)(base + d_rel[it]) : kOOBw;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(win + buf * kWinBytes + q * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    // the offset / mask logits of this wave's 9 x 16 geometry entries (entry e = r*64 + lane: tap e >> 4, pixel e & 15)
    float l_oh[3], l_ow[3], l_ml[3];
    auto issue_logits = [&](int t) {
        int b, ty0, tx0;
        decode(t, b, ty0, tx0);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = r * 64 + lane, tap = e >> 4, nn = e & 15;
            const int y = ty0 + 2 * wave + (nn >> 3), x = tx0 + (nn & 7);
            l_oh[r] = l_ow[r] = l_ml[r] = 0.f;
            if (e < 144 && y < p.Ho && x < p.Wo) {
                const int64_t ob = b * p.off_sb + y * p.off_sy + x * p.off_sx;
                l_oh[r] = p.offset[ob + (int64_t)(2 * tap) * p.off_sc];
                l_ow[r] = p.offset[ob + (int64_t)(2 * tap + 1) * p.off_sc];
                if (p.mask) l_ml[r] = p.mask[b * p.msk_sb + y * p.msk_sy + x * p.msk_sx + (int64_t)tap * p.msk_sc];
            }
        }
    };
    // geometry entry: { (h_low + 1) | (w_low + 1) << 16, lh, lw, m }; an invalid sample (outside (-1, H) x (-1, W), or a pixel
    // beyond the image edge of a ragged tile) carries m = 0 and the tile origin (always inside the window)
    auto write_geometry = [&](int t, int buf) -> bool {
        int b, ty0, tx0;
        decode(t, b, ty0, tx0);
        bool leaves = false;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = r * 64 + lane, tap = e >> 4, nn = e & 15;
            if (e < 144) {
                const int y = ty0 + 2 * wave + (nn >> 3), x = tx0 + (nn & 7);
                const int ti = tap / 3, tj = tap - ti * 3;
                int hl = ty0, wl = tx0;
                float lh = 0.f, lw = 0.f, m = 0.f;
                if (y < p.Ho && x < p.Wo) {
                    const float h_im = (float)(y - 1 + ti) + l_oh[r];
                    const float w_im = (float)(x - 1 + tj) + l_ow[r];
                    if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                        m = 1.f;
                        if (p.mask) {
                            m = l_ml[r];
                            if (p.mask_sigmoid) m = __frcp_rn(1.0f + __expf(-m));
                        }
                        hl = (int)floorf(h_im);
                        wl = (int)floorf(w_im);
                        lh = h_im - (float)hl;
                        lw = w_im - (float)wl;
                    }
                }
                *(i32x4*)(geo + buf * kWinGeo + wave * kWinGeoWave + e * 16) = i32x4{(hl + 1) | ((wl + 1) << 16), f2i(lh), f2i(lw), f2i(m)};
                leaves |= !((unsigned)(hl - (ty0 - kWinP)) <= (unsigned)(kWinD - 2) && (unsigned)(wl - (tx0 - kWinP)) <= (unsigned)(kWinD - 2));
            }
        }
        return __builtin_amdgcn_ballot_w64(leaves) != 0;         // wave-uniform: this wave's tile needs the global-gather path somewhere
    };

    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x7fffffff, 0x00020000);
    int t = tile_of(0);
    bool slow = false, slow_next = false;
    if (t < ntiles) {
        issue_window(t, 0);
        issue_logits(t);
        slow = write_geometry(t, 0);
    }
    for (int k = 0; t < ntiles; ++k) {
        const int buf = k & 1;
        // the window of this tile (DMA issued a whole tile ago) must have landed; the only younger VMEM operations of this wave are
        // the previous tile's two (unconditional) output stores, which need not be waited for
        if (k == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int tn = tile_of(k + 1);
        if (tn < ntiles) {
            issue_window(tn, buf ^ 1);
            issue_logits(tn);
        }
        int b, ty0, tx0;
        decode(t, b, ty0, tx0);
        const char* W0 = win + buf * kWinBytes;
        char* G0 = geo + buf * kWinGeo + wave * kWinGeoWave;
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // The nine taps as straight-line code (FAST: every sample of this wave's tile inside the window -- known since the geometry
        // was computed -- so no branch separates the taps and the next tap's LDS reads are scheduled under this tap's blend), or with
        // the per-tap window test and the global-gather fallback (SLOW).
        auto taps = [&](auto slow_c) {
        constexpr bool SLOW = decltype(slow_c)::value;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const i32x4 ge = *(const i32x4*)(G0 + (tap * 16 + n) * 16);
            const int hl = (ge[0] & 0xffff) - 1, wl = (int)((uint32_t)ge[0] >> 16) - 1;
            const float lh = i2f(ge[1]), lw = i2f(ge[2]), m = i2f(ge[3]);
            const float hh = 1.f - lh, hw = 1.f - lw;
            float w[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c] *= m;                  // the modulation folded into the weights, as in dcn_nhwc_kernel
            const int wy = hl - (ty0 - kWinP), wx = wl - (tx0 - kWinP);
            const bool inwin = (unsigned)wy <= (unsigned)(kWinD - 2) && (unsigned)wx <= (unsigned)(kWinD - 2);
            i32x4 cv[2][4];
            if (!SLOW || __builtin_amdgcn_ballot_w64(!inwin) == 0) {
                const int pw = wy * kWinD + wx;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int q = pw + (c >> 1) * kWinD + (c & 1);
                    const int a0 = q * 128 + ((g ^ ((q >> 1) & 7)) << 4);
                    cv[0][c] = *(const i32x4*)(W0 + a0);
                    cv[1][c] = *(const i32x4*)(W0 + (a0 ^ 64));              // vector g + 4: the swizzled slot differs in bit 2 only
                }
            } else {
                // some sample of this (wave, tap) leaves the window: gather every corner from global memory (out-of-image -> zeros)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int y = hl + (c >> 1), x = wl + (c & 1);
                    const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                    const uint32_t off = ok ? (uint32_t)((b * p.in_sb + y * p.in_sy + x * p.in_sx + g * 8) * 2) : kOOBw;
                    cv[0][c] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0));
                    cv[1][c] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? off + 64 : kOOBw, 0, 0));
                }
            }
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                float vals[8];
                Vec16<T> o;
                bool packed = false;
                if constexpr (std::is_same<T, hf16>::value) {
                  if (p.pk16) {
                    o.raw = dcn_blend8_pk(cv[sx][0], cv[sx][1], cv[sx][2], cv[sx][3], dcn_pk_weight(w[0]), dcn_pk_weight(w[1]), dcn_pk_weight(w[2]), dcn_pk_weight(w[3]));
                    packed = true;
                  } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        vals[2 * d] = mix_fma_lo(cv[sx][3][d], w[3], mix_fma_lo(cv[sx][2][d], w[2], mix_fma_lo(cv[sx][1][d], w[1], mix_mul_lo(cv[sx][0][d], w[0]))));
                        vals[2 * d + 1] = mix_fma_hi(cv[sx][3][d], w[3], mix_fma_hi(cv[sx][2][d], w[2], mix_fma_hi(cv[sx][1][d], w[1], mix_mul_hi(cv[sx][0][d], w[0]))));
                    }
                  }
                } else {
                    Vec16<T> c1, c2, c3, c4;
                    c1.raw = cv[sx][0]; c2.raw = cv[sx][1]; c3.raw = cv[sx][2]; c4.raw = cv[sx][3];
#pragma unroll
                    for (int e = 0; e < 8; ++e) vals[e] = fmaf(w[3], c4.get(e), fmaf(w[2], c3.get(e), fmaf(w[1], c2.get(e), w[0] * c1.get(e))));
                }
                if (!packed) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.set2(e, vals[2 * e], vals[2 * e + 1]);
                }
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    const i32x4 fa = *(const i32x4*)(wts + (((tap * 2 + sx) * 4 + blk) * 64 + lane) * 16);
                    Fmt16<T>::mfma16(fa, o.raw, acc[blk]);
                }
            }
        }
        };
        if (slow) taps(std::true_type{});
        else taps(std::false_type{});
        // ---- epilogue: bias, folded BN, ReLU; the wave's 16 pixels x 64 channels parked in its own geometry slot, out as whole lines
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = (acc[blk][e] + cb[blk][e]) * cs[blk][e] + ct[blk][e];
                if (p.relu) x = fmaxf(x, 0.f);
                v[e] = x;
            }
            i32x2 o2;
            o2[0] = Fmt16<T>::pack2(v[0], v[1]);
            o2[1] = Fmt16<T>::pack2(v[2], v[3]);
            *(i32x2*)(G0 + n * 128 + (blk * 16 + g * 4) * 2) = o2;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = lane + 64 * r, px = j >> 3, part = j & 7;
            const int y = ty0 + 2 * wave + (px >> 3), x = tx0 + (px & 7);
            const i32x4 val = *(const i32x4*)(G0 + px * 128 + part * 16);
            const uint32_t off = (y < p.Ho && x < p.Wo) ? (uint32_t)((b * p.out_sb + y * p.out_sy + x * p.out_sx) * 2 + part * 16) : kOOBw;
            __builtin_amdgcn_raw_buffer_store_b128(val, out_rsrc, off, 0, 0);      // always issued (the vmcnt count above relies on it)
        }
        if (tn < ntiles) slow_next = write_geometry(tn, buf ^ 1);
        slow = slow_next;
        t = tn;
    }
}

// ---- 64 -> 64 channel DCNv2, second design: the window kernel above with TWO waves per SIMD ---------------------------------------
// dcn_win64_kernel lost to the gather kernel because 151 KB of LDS (74 KB of it weights) left one wave per SIMD.  Here the weights live
// in REGISTERS, split over K: a workgroup is 8 waves = 4 pixel groups (16 pixels of the 8 x 8 tile) x 2 K halves; wave (pg, kh) keeps the
// A fragments of all 64 output channels x its 32 input channels x 9 taps (36 fragments = 144 VGPRs, as conv_resident64), blends ONLY its
// 32 channels (one 16-byte vector per lane and tap: 4 corner ds_read_b128, 32 FMAs) and issues 4 MFMAs per tap.  The two K partial sums
// of a pixel group meet in LDS: the tile's non-owner parks its 16 accumulators, the OWNER (alternating with the tile index, so both waves
// pay the epilogue on every other tile) adds them at the start of the NEXT iteration -- after that iteration's barrier, so a tile still
// costs ONE barrier -- and stores whole 128-byte lines (BOTH waves park their partial sums: keeping the owner's in registers across the
// iteration spilled).  LDS: two 29 KiB windows + geometry + partials = 159 KB, no weights.  Geometry entries carry the four corner
// addresses (swizzle key included) and the four modulated weights, computed once per (pixel, tap) instead of by the 8 lanes sharing it.
// MEASURED (round 3, 16 x 128 x 440 fp16, sigma 0.5 px): 329 us against 350 us for the gather kernel on the same box (361 before the
// precomputed addresses); bf16 537 us (26 spilled registers).  Ablations (runtime flags, removed again): no blend -42 us, no corner
// reads -26, neither -73, no MFMA 0, no window DMA -32, no logit loads -29, no reduction / epilogue -50, ALL of them off: 188 us --
// i.e. 3.4 us per 64-pixel tile of barrier, geometry, tile decode and hand-over that no amount of gather tuning removes.  An 8 x 8 tile
// is too small a unit of work per barrier for a persistent design, and larger tiles do not fit two windows in LDS.  OPT-IN
// (VD3D_DCN_KSPLIT=1); results differ from the gather kernel by the fp32 order of the two K halves (<= 1 ulp, tests/test_dcn_gpu.py).
constexpr int kKsGeoGrp = 9 * 16 * 32;                                     // per pixel group: 9 taps x 16 pixels x 32 B
constexpr int kKsGeo = 4 * kKsGeoGrp, kKsRed = 4 * 2 * 4096;               // per buffer: 4 pixel groups x 2 K halves x 16 accumulators
constexpr int kKsLdsWin = 0, kKsLdsGeo = 2 * kWinBytes, kKsLdsRed = kKsLdsGeo + 2 * kKsGeo, kKsLdsFlag = kKsLdsRed + 2 * kKsRed;
constexpr int kKsLdsTab = kKsLdsFlag + 64, kKsLds = kKsLdsTab + 3 * 64 * 4;

template <typename T>
__global__ void __launch_bounds__(512) dcn_ks64_kernel(const DcnArgs p, int ntiles, int tiles_x, int tiles_y) {
    static_assert(sizeof(T) == 2, "16-bit formats only");
    constexpr uint32_t kOOBw = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* win = smem + kKsLdsWin;
    char* geo = smem + kKsLdsGeo;
    char* red = smem + kKsLdsRed;
    int* flags = (int*)(smem + kKsLdsFlag);            // [2 buffers][4 groups][2 waves]
    float* ctab = (float*)(smem + kKsLdsTab);          // bias | scale | shift
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave >> 1, kh = wave & 1;
    const int n = lane & 15, g = lane >> 4;
    const int vsel = kh * 4 + g;                        // this lane's 16-byte vector (8 channels) of a pixel
    const uint32_t vsel2 = (uint32_t)vsel | ((uint32_t)vsel << 16);
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_xcd = nwg >> 3;
    const int chunk = (ntiles + 7) >> 3;
    auto tile_of = [&](int k) { const int i = jx + k * per_xcd; return i < chunk ? xcd * chunk + i : ntiles; };
    const int tiles_img = tiles_x * tiles_y;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x7fffffff, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    if (tid < 64) {
        ctab[tid] = p.bias ? p.bias[tid] : 0.f;
        ctab[64 + tid] = p.scale ? p.scale[tid] : 1.f;
        ctab[128 + tid] = p.shift ? p.shift[tid] : 0.f;
    }
    // ---- weights -> registers: fragment (tap, blk): row blk*16 + n, k = tap*64 + kh*32 + g*8 .. +7
    i32x4 wf[9][4];
    {
        const char* wbase = (const char*)p.w + ((size_t)n * p.Kpad + kh * 32 + g * 8) * 2;
        static_for<36>([&](auto ic) {
            constexpr int i = decltype(ic)::value, tap = i / 4, blk = i % 4;
            wf[tap][blk] = *(const i32x4*)(wbase + ((size_t)blk * 16 * p.Kpad + tap * 64) * 2);
        });
    }
    constexpr int NIT = (kWinPieces + 7) / 8;
    int d_rel[NIT], d_yx[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = wave + 8 * it, c = q * 64 + lane, pp = c >> 3, sl = c & 7;
        const int v = sl ^ ((pp >> 1) & 7);
        const int wy = pp / kWinD, wx = pp - wy * kWinD;
        d_rel[it] = (int)((wy * p.in_sy + wx * p.in_sx + v * 8) * 2);
        d_yx[it] = (q < kWinPieces && pp < kWinPix) ? (wy | (wx << 8)) : -1;
    }
    auto decode = [&](int t, int& b, int& ty0, int& tx0) {
        b = t / tiles_img;
        const int r = t - b * tiles_img, ty = r / tiles_x;
        ty0 = ty * 8;
        tx0 = (r - ty * tiles_x) * 8;
    };
    auto issue_window = [&](int t, int buf) {
        int b, ty0, tx0;
        decode(t, b, ty0, tx0);
        const int base = (int)((b * p.in_sb + (ty0 - kWinP) * p.in_sy + (tx0 - kWinP) * p.in_sx) * 2);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = wave + 8 * it;
            if (q < kWinPieces) {
                const int wy = d_yx[it] & 255, wx = (d_yx[it] >> 8) & 255;
                const bool ok = d_yx[it] >= 0 && (unsigned)(ty0 - kWinP + wy) < (unsigned)p.H && (unsigned)(tx0 - kWinP + wx) < (unsigned)p.W;
                const uint32_t off = ok ? (uint32_t)(base + d_rel[it]) : kOOBw;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(win + buf * kWinBytes + q * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    // geometry entries of this pixel group: e = r*128 + kh*64 + lane (tap e >> 4, pixel e & 15), split between the pair's two waves
    float l_oh[2], l_ow[2], l_ml[2];
    auto issue_logits = [&](int t) {
        int b, ty0, tx0;
        decode(t, b, ty0, tx0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = r * 128 + kh * 64 + lane, tap = e >> 4, nn = e & 15;
            const int y = ty0 + 2 * pg + (nn >> 3), x = tx0 + (nn & 7);
            l_oh[r] = l_ow[r] = l_ml[r] = 0.f;
            if (e < 144 && y < p.Ho && x < p.Wo) {
                const int64_t ob = b * p.off_sb + y * p.off_sy + x * p.off_sx;
                l_oh[r] = p.offset[ob + (int64_t)(2 * tap) * p.off_sc];
                l_ow[r] = p.offset[ob + (int64_t)(2 * tap + 1) * p.off_sc];
                if (p.mask) l_ml[r] = p.mask[b * p.msk_sb + y * p.msk_sy + x * p.msk_sx + (int64_t)tap * p.msk_sc];
            }
        }
    };
    auto write_geometry = [&](int t, int buf) {
        int b, ty0, tx0;
        decode(t, b, ty0, tx0);
        bool leaves = false;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = r * 128 + kh * 64 + lane, tap = e >> 4, nn = e & 15;
            if (e < 144) {
                const int y = ty0 + 2 * pg + (nn >> 3), x = tx0 + (nn & 7);
                const int ti = tap / 3, tj = tap - ti * 3;
                int hl = ty0, wl = tx0;
                float lh = 0.f, lw = 0.f, m = 0.f;
                if (y < p.Ho && x < p.Wo) {
                    const float h_im = (float)(y - 1 + ti) + l_oh[r];
                    const float w_im = (float)(x - 1 + tj) + l_ow[r];
                    if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                        m = 1.f;
                        if (p.mask) {
                            m = l_ml[r];
                            if (p.mask_sigmoid) m = __frcp_rn(1.0f + __expf(-m));
                        }
                        hl = (int)floorf(h_im);
                        wl = (int)floorf(w_im);
                        lh = h_im - (float)hl;
                        lw = w_im - (float)wl;
                    }
                }
                // entry (32 B): { (h_low + 1) | (w_low + 1) << 16,  a0 | a1 << 16,  a2 | a3 << 16,  0,  w0, w1, w2, w3 } with a_c = LDS
                // address of corner c's pixel in 16-byte units INCLUDING its swizzle key (a lane adds its own vector by XOR) and
                // w_c = bilinear weight x modulation: computed ONCE per (pixel, tap) here instead of by all 8 lanes that share it
                const int wy = hl - (ty0 - kWinP), wx = wl - (tx0 - kWinP);
                const bool inw = (unsigned)wy <= (unsigned)(kWinD - 2) && (unsigned)wx <= (unsigned)(kWinD - 2);
                leaves |= !inw;
                uint32_t ac[4] = {0, 0, 0, 0};
                if (inw) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int q = (wy + (c >> 1)) * kWinD + wx + (c & 1);
                        ac[c] = (uint32_t)(q * 8 + ((q >> 1) & 7));
                    }
                }
                const float hh = 1.f - lh, hw = 1.f - lw;
                float w4[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
                for (int c = 0; c < 4; ++c) w4[c] *= m;
                char* ge = geo + buf * kKsGeo + pg * kKsGeoGrp + e * 32;
                *(i32x4*)ge = i32x4{(hl + 1) | ((wl + 1) << 16), (int)(ac[0] | (ac[1] << 16)), (int)(ac[2] | (ac[3] << 16)), 0};
                *(f32x4*)(ge + 16) = f32x4{w4[0], w4[1], w4[2], w4[3]};
            }
        }
        const int any = __builtin_amdgcn_ballot_w64(leaves) != 0;
        if (lane == 0) flags[(buf * 4 + pg) * 2 + kh] = any;
    };

    int t = tile_of(0);
    if (t < ntiles) {
        issue_window(t, 0);
        issue_logits(t);
        write_geometry(t, 0);
    }
    int t_prev = ntiles;                                // tile whose reduction + epilogue is still due
    // finish tile `tp` (iteration kp): the owner adds the partner's parked partial sums, applies bias / BN / ReLU, parks the 16 pixels x
    // 64 channels in the same LDS region and stores whole lines
    auto finish = [&](int tp, int kp) {
        if (kh != (kp & 1)) return;                    // wave-uniform: not the owner of that tile
        int b, ty0, tx0;
        decode(tp, b, ty0, tx0);
        char* R0 = red + (kp & 1) * kKsRed + pg * 8192;
        f32x4 fin[4];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const f32x4 mine = *(const f32x4*)(R0 + kh * 4096 + (blk * 64 + lane) * 16);
            const f32x4 other = *(const f32x4*)(R0 + (kh ^ 1) * 4096 + (blk * 64 + lane) * 16);
            const f32x4 bz = *(const f32x4*)(ctab + blk * 16 + g * 4), sc = *(const f32x4*)(ctab + 64 + blk * 16 + g * 4),
                        sh = *(const f32x4*)(ctab + 128 + blk * 16 + g * 4);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = ((mine[e] + other[e]) + bz[e]) * sc[e] + sh[e];
                if (p.relu) x = fmaxf(x, 0.f);
                v[e] = x;
            }
            fin[blk] = f32x4{v[0], v[1], v[2], v[3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the partner's partials are in registers: the region becomes the parking tile
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            i32x2 o2;
            o2[0] = Fmt16<T>::pack2(fin[blk][0], fin[blk][1]);
            o2[1] = Fmt16<T>::pack2(fin[blk][2], fin[blk][3]);
            *(i32x2*)(R0 + n * 128 + (blk * 16 + g * 4) * 2) = o2;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = lane + 64 * r, px = j >> 3, part = j & 7;
            const int y = ty0 + 2 * pg + (px >> 3), x = tx0 + (px & 7);
            const i32x4 val = *(const i32x4*)(R0 + px * 128 + part * 16);
            const uint32_t off = (y < p.Ho && x < p.Wo) ? (uint32_t)((b * p.out_sb + y * p.out_sy + x * p.out_sx) * 2 + part * 16) : kOOBw;
            __builtin_amdgcn_raw_buffer_store_b128(val, out_rsrc, off, 0, 0);
        }
    };
    int k = 0;
    for (; t < ntiles; ++k) {
        const int buf = k & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t_prev < ntiles) finish(t_prev, k - 1);
        const int tn = tile_of(k + 1);
        if (tn < ntiles) {
            issue_window(tn, buf ^ 1);
            issue_logits(tn);
        }
        int b, ty0, tx0;
        decode(t, b, ty0, tx0);
        const char* W0 = win + buf * kWinBytes;
        const char* G0 = geo + buf * kKsGeo + pg * kKsGeoGrp;
        const bool slow = (flags[(buf * 4 + pg) * 2] | flags[(buf * 4 + pg) * 2 + 1]) != 0;
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto taps = [&](auto slow_c) {
            constexpr bool SLOW = decltype(slow_c)::value;
            static_for<9>([&](auto tc) {
                constexpr int tap = decltype(tc)::value;
                const i32x4 ge = *(const i32x4*)(G0 + (tap * 16 + n) * 32);
                const f32x4 gw = *(const f32x4*)(G0 + (tap * 16 + n) * 32 + 16);
                const float w[4] = {gw[0], gw[1], gw[2], gw[3]};
                const int hl = (ge[0] & 0xffff) - 1, wl = (int)((uint32_t)ge[0] >> 16) - 1;
                bool inwin = true;
                if constexpr (SLOW) {
                    const int wy = hl - (ty0 - kWinP), wx = wl - (tx0 - kWinP);
                    inwin = (unsigned)wy <= (unsigned)(kWinD - 2) && (unsigned)wx <= (unsigned)(kWinD - 2);
                }
                i32x4 cv[4];
                if (!SLOW || __builtin_amdgcn_ballot_w64(!inwin) == 0) {
                    const uint32_t a01 = (uint32_t)ge[1] ^ vsel2, a23 = (uint32_t)ge[2] ^ vsel2;      // both corners of a pair get the lane's vector
                    cv[0] = *(const i32x4*)(W0 + ((a01 & 0xffffu) << 4));
                    cv[1] = *(const i32x4*)(W0 + ((a01 >> 16) << 4));
                    cv[2] = *(const i32x4*)(W0 + ((a23 & 0xffffu) << 4));
                    cv[3] = *(const i32x4*)(W0 + ((a23 >> 16) << 4));
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int y = hl + (c >> 1), x = wl + (c & 1);
                        const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                        const uint32_t off = ok ? (uint32_t)((b * p.in_sb + y * p.in_sy + x * p.in_sx + vsel * 8) * 2) : kOOBw;
                        cv[c] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0));
                    }
                }
                float vals[8];
                Vec16<T> o;
                bool packed = false;
                if constexpr (std::is_same<T, hf16>::value) {
                  if (p.pk16) {
                    o.raw = dcn_blend8_pk(cv[0], cv[1], cv[2], cv[3], dcn_pk_weight(w[0]), dcn_pk_weight(w[1]), dcn_pk_weight(w[2]), dcn_pk_weight(w[3]));
                    packed = true;
                  } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        vals[2 * d] = mix_fma_lo(cv[3][d], w[3], mix_fma_lo(cv[2][d], w[2], mix_fma_lo(cv[1][d], w[1], mix_mul_lo(cv[0][d], w[0]))));
                        vals[2 * d + 1] = mix_fma_hi(cv[3][d], w[3], mix_fma_hi(cv[2][d], w[2], mix_fma_hi(cv[1][d], w[1], mix_mul_hi(cv[0][d], w[0]))));
                    }
                  }
                } else {
                    Vec16<T> c1, c2, c3, c4;
                    c1.raw = cv[0]; c2.raw = cv[1]; c3.raw = cv[2]; c4.raw = cv[3];
#pragma unroll
                    for (int e = 0; e < 8; ++e) vals[e] = fmaf(w[3], c4.get(e), fmaf(w[2], c3.get(e), fmaf(w[1], c2.get(e), w[0] * c1.get(e))));
                }
                if (!packed) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.set2(e, vals[2 * e], vals[2 * e + 1]);
                }
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) Fmt16<T>::mfma16(wf[tap][blk], o.raw, acc[blk]);
            });
        };
        if (slow) taps(std::true_type{});
        else taps(std::false_type{});
        // hand-over: both waves of the pair park their partial sums; the tile's owner adds them after the next barrier
        {
            char* R0 = red + buf * kKsRed + pg * 8192 + kh * 4096;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) *(f32x4*)(R0 + (blk * 64 + lane) * 16) = acc[blk];
        }
        t_prev = t;
        if (tn < ntiles) write_geometry(tn, buf ^ 1);
        t = tn;
    }
    if (t_prev < ntiles) {                             // the last tile's reduction (uniform over the workgroup: every wave ran k iterations)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        finish(t_prev, k - 1);
    }
}

// ---- sampled columns in HBM (large output-channel counts) -------------------------------------------------------------
// The fused kernel above produces a pixel tile's sampled columns once per BN <= 256 output channels: with O = 2176 (the DCNv2
// head of BASELINE config 3) the gather + blend is repeated 9 times and the launch takes 12 ms.  For such layers the columns
// are written once, in the storage dtype -- the very rounding point of the fused path -- as [B][Ho][Wo][tap][C], and the
// contraction runs as a 1x1 convolution over K = taps x C on the strip tiles of conv_igemm.hip (the reference materialises
// the same matrix, in fp32: deform_conv_cuda.cpp:531-569).  One thread = one (pixel, tap, 16-byte channel vector); identical
// sampling arithmetic to dcn_nhwc_kernel (same fma chain, same modulation order).
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

// the LDS-window kernel: 16-bit, 3x3 / s1 / p1 / d1, C = O = 64, line-aligned NHWC output, tensors addressable with 32-bit offsets
static bool dcn_win64_shape_ok(const DcnArgs& a, int es);
static bool dcn_win64_ok(const DcnArgs& a, int es) { return dcn_win64_shape_ok(a, es) && vd3d_switch(VD3D_SW_DCN_WINDOW); }
static bool dcn_win64_shape_ok(const DcnArgs& a, int es) {
    const int64_t in_span = ((int64_t)(a.B - 1) * a.in_sb + (int64_t)(a.H - 1) * a.in_sy + (int64_t)(a.W - 1) * a.in_sx + a.C) * es;
    return es == 2 && a.kh == 3 && a.kw == 3 && a.sh == 1 && a.sw == 1 && a.ph == 1 && a.pw == 1 && a.dh == 1 && a.dw == 1 && a.Cg == 64 &&
           a.C == 64 && a.O == 64 && a.Kpad >= 576 && a.H < 32768 && a.W < 32768 && in_span < 0x7ffffff0ll &&
           a.out_sx % 8 == 0 && a.out_sy % 8 == 0 && a.out_sb % 8 == 0 && ((uintptr_t)a.out & 15) == 0 &&
           ((int64_t)(a.B - 1) * a.out_sb + (int64_t)(a.H - 1) * a.out_sy + (int64_t)(a.W - 1) * a.out_sx + a.O) * es < 0x7ffffff0ll &&
           true;
}

template <typename T>
int launch_dcn_lw64(const DcnArgs& a, hipStream_t s) {
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)dcn_lw64_kernel<T>, kLwLds, lim, "hipFuncSetAttribute(dcn_lw64)")) return rc;
    const int tiles_x = (a.Wo + 7) / 8, tiles_y = (a.Ho + 7) / 8;
    if ((int64_t)tiles_x * tiles_y > 0x7fffffff || a.B > 65535) return VD3D_ERANGE;
    hipLaunchKernelGGL(dcn_lw64_kernel<T>, dim3((unsigned)(tiles_x * tiles_y), 1, a.B), dim3(256), kLwLds, s, a, tiles_x);
    return vd3d_check_launch("deform_conv(lds window)");
}

template <typename T>
int launch_dcn_bf64(const DcnArgs& a, hipStream_t s) {
    static Vd3dLdsLimit lim, lim_pk;
    const int tiles_x = (a.Wo + kBfTW - 1) / kBfTW, tiles_y = (a.Ho + kBfTH - 1) / kBfTH;
    if ((int64_t)tiles_x * tiles_y > 0x7fffffff || a.B > 65535) return VD3D_ERANGE;
    const dim3 grid((unsigned)(tiles_x * tiles_y), 1, a.B);
    if constexpr (std::is_same<T, hf16>::value) {
        if (a.pk16) {
            if (const int rc = vd3d_raise_lds_limit((const void*)dcn_bf64_kernel<T, true>, kBfLds, lim_pk, "hipFuncSetAttribute(dcn_bf64)")) return rc;
            hipLaunchKernelGGL((dcn_bf64_kernel<T, true>), grid, dim3(256), kBfLds, s, a, tiles_x);
            return vd3d_check_launch("deform_conv(barrier-free)");
        }
    }
    if (const int rc = vd3d_raise_lds_limit((const void*)dcn_bf64_kernel<T, false>, kBfLds, lim, "hipFuncSetAttribute(dcn_bf64)")) return rc;
    hipLaunchKernelGGL((dcn_bf64_kernel<T, false>), grid, dim3(256), kBfLds, s, a, tiles_x);
    return vd3d_check_launch("deform_conv(barrier-free)");
}

template <typename T>
int launch_dcn_win64(const DcnArgs& a, hipStream_t s) {
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)dcn_win64_kernel<T>, kWinLds, lim, "hipFuncSetAttribute(dcn_win64)")) return rc;
    const int tiles_x = (a.Wo + 7) / 8, tiles_y = (a.Ho + 7) / 8;
    const int64_t ntiles = (int64_t)a.B * tiles_x * tiles_y;
    if (ntiles > 0x7fffffff) return VD3D_ERANGE;
    const int cus = vd3d_device_cu_count();
    if (cus < 8) return VD3D_ELAUNCH;
    int64_t grid = (ntiles + 7) / 8 * 8;                       // a multiple of 8: one lane of workgroups per XCD
    if (grid > cus / 8 * 8) grid = cus / 8 * 8;
    hipLaunchKernelGGL(dcn_win64_kernel<T>, dim3((unsigned)grid), dim3(256), kWinLds, s, a, (int)ntiles, tiles_x, tiles_y);
    return vd3d_check_launch("deform_conv(window)");
}

template <typename T>
int launch_dcn_ks64(const DcnArgs& a, hipStream_t s) {
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)dcn_ks64_kernel<T>, kKsLds, lim, "hipFuncSetAttribute(dcn_ks64)")) return rc;
    const int tiles_x = (a.Wo + 7) / 8, tiles_y = (a.Ho + 7) / 8;
    const int64_t ntiles = (int64_t)a.B * tiles_x * tiles_y;
    if (ntiles > 0x7fffffff) return VD3D_ERANGE;
    const int cus = vd3d_device_cu_count();
    if (cus < 8) return VD3D_ELAUNCH;
    int64_t grid = (ntiles + 7) / 8 * 8;
    if (grid > cus / 8 * 8) grid = cus / 8 * 8;
    hipLaunchKernelGGL(dcn_ks64_kernel<T>, dim3((unsigned)grid), dim3(512), kKsLds, s, a, (int)ntiles, tiles_x, tiles_y);
    return vd3d_check_launch("deform_conv(k-split window)");
}

template <typename T>
int dispatch_dcn_nhwc(const DcnArgs& a, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        if (dcn_win64_shape_ok(a, 2) && vd3d_switch(VD3D_SW_DCN_KSPLIT)) return launch_dcn_ks64<T>(a, s);
        if (dcn_win64_ok(a, 2)) return launch_dcn_win64<T>(a, s);
        // C = O = 64, 3x3 / s1 / p1 (KM3D's full-resolution DLA-Up nodes), round-4 experiments, both OPT-IN (neither beats the gather kernel inside
        // the model): VD3D_DCN_LWIN=1 = the gather kernel with its corners from an LDS window, VD3D_DCN_BF=1 = the barrier-free kernel
        if (dcn_win64_shape_ok(a, 2) && a.H * a.in_sy * 2 < 0x7ffffff0ll && vd3d_switch(VD3D_SW_DCN_LWIN)) return launch_dcn_lw64<T>(a, s);   // opt-in
        if (dcn_win64_shape_ok(a, 2) && a.H * a.in_sy * 2 < 0x7ffffff0ll && vd3d_switch(VD3D_SW_DCN_BF)) return launch_dcn_bf64<T>(a, s);       // opt-in
    }
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
    a.pk16 = (q->dtype == VD3D_F16 && vd3d_switch(VD3D_SW_DCN_PK16)) ? 1 : 0;      // opt-in: leaves the 2-ulp bar (see dcn_blend8_pk)
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
#ifdef VD3D_STAMPS
extern "C" int vd3d_debug_read_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 4 * 32); }
#endif

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
