// elementwise.hip -- HBM-bound NHWC helpers: image packing, pools, depth-wise 3x3, slice copies,
// boundary layout converters.  All loads/stores are 16-byte vectors along the channel axis (coalesced:
// consecutive lanes -> consecutive 16 B of one pixel, then the next pixel).
#include "common.h"

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------------------
// NCHW fp32 image -> zero-bordered NHWC4 (stem input).  One thread per output pixel (8 or 16 bytes).
template <typename T>
__global__ void pack_image_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int H, int W,
                                  int pad_y, int pad_l, int Hp, int Wp) {
    const int64_t total = (int64_t)B * Hp * Wp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xp = (int)(i % Wp);
        const int yp = (int)((i / Wp) % Hp);
        const int b = (int)(i / ((int64_t)Wp * Hp));
        const int x = xp - pad_l, y = yp - pad_y;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
            const int64_t base = ((int64_t)b * 3 * H + y) * W + x;
            v0 = in[base];
            v1 = in[base + (int64_t)H * W];
            v2 = in[base + 2 * (int64_t)H * W];
        }
        if constexpr (sizeof(T) == 2) {
            i32x2 o;
            o[0] = Fmt16<T>::pack2(v0, v1);
            o[1] = Fmt16<T>::pack2(v2, 0.f);
            *(i32x2*)(out + i * 4) = o;
        } else {
            f32x4 o = {v0, v1, v2, 0.f};
            *(f32x4*)(out + i * 4) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C,
                                    int Ho, int Wo, int ips, int ops) {
    constexpr int VE = ElemTraits<T>::kVec;
    const int cv = C / VE;
    const int64_t total = (int64_t)B * Ho * Wo * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VE;
        const int64_t pix = i / cv;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((int64_t)Wo * Ho));
        float m[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                Vec16<T> v;
                v.raw = *(const i32x4*)(in + (((int64_t)b * H + iy) * W + ix) * ips + c);
#pragma unroll
                for (int e = 0; e < VE; ++e) m[e] = fmaxf(m[e], v.get(e));
            }
        }
        Vec16<T> o;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, m[2 * e], m[2 * e + 1]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, m[e]);
        }
        *(i32x4*)(out + pix * ops + c) = o.raw;
    }
}

template <typename T>
__global__ void avgpool2x2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C,
                                  int Ho, int Wo, int ips, int ops) {
    constexpr int VE = ElemTraits<T>::kVec;
    const int cv = C / VE;
    const int64_t total = (int64_t)B * Ho * Wo * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VE;
        const int64_t pix = i / cv;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((int64_t)Wo * Ho));
        float s[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) s[e] = 0.f;
        // torch avg_pool2d accumulates the window row-major in fp32: ((a00 + a01) + a10) + a11, then * 0.25
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                Vec16<T> v;
                v.raw = *(const i32x4*)(in + (((int64_t)b * H + oy * 2 + dy) * W + ox * 2 + dx) * ips + c);
#pragma unroll
                for (int e = 0; e < VE; ++e) s[e] += v.get(e);
            }
        Vec16<T> o;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, s[2 * e] * 0.25f, s[2 * e + 1] * 0.25f);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, s[e] * 0.25f);
        }
        *(i32x4*)(out + pix * ops + c) = o.raw;
    }
}

// depth-wise 3x3, pad 1, stride 1; weight [9][C] fp32
template <typename T>
__global__ void dwconv3x3_kernel(const T* __restrict__ in, const float* __restrict__ w, const float* __restrict__ scale,
                                 const float* __restrict__ shift, T* __restrict__ out, int B, int H, int W, int C,
                                 int ips, int ops, int relu) {
    constexpr int VE = ElemTraits<T>::kVec;
    const int cv = C / VE;
    const int64_t total = (int64_t)B * H * W * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VE;
        const int64_t pix = i / cv;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        float s[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) s[e] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = y - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = x - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                Vec16<T> v;
                v.raw = *(const i32x4*)(in + (((int64_t)b * H + iy) * W + ix) * ips + c);
                const float* wt = w + (dy * 3 + dx) * C + c;
#pragma unroll
                for (int e = 0; e < VE; ++e) s[e] = fmaf(v.get(e), wt[e], s[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            float v = s[e] * scale[c + e] + shift[c + e];
            s[e] = relu ? fmaxf(v, 0.f) : v;
        }
        Vec16<T> o;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, s[2 * e], s[2 * e + 1]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, s[e]);
        }
        *(i32x4*)(out + pix * ops + c) = o.raw;
    }
}

// The same arithmetic (taps added row-major in fp32, a tap outside the image adds nothing: bit-identical to dwconv3x3_kernel) with a thread owning a RUN
// of XR output pixels of one row and one 16-byte channel vector: 3 x (XR + 2) vector loads per XR outputs instead of 9 per output, weights and folded BN
// from LDS instead of per-thread global loads.  (Round 6: the step is throughput bound -- tools/side_work_cost.py -- so the ghost modules' depth-wise
// launches, 59 us of a headline step on the old kernel, count even on the side stream.)
template <typename T, int XR>
__global__ void __launch_bounds__(256) dwconv3x3_run_kernel(const T* __restrict__ in, const float* __restrict__ w, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, T* __restrict__ out, int B, int H, int W, int C,
                                                            int ips, int ops, int relu) {
    constexpr int VE = ElemTraits<T>::kVec;
    extern __shared__ __attribute__((aligned(16))) float wl[];            // [9][C] weights | scale[C] | shift[C]
    for (int i = threadIdx.x; i < 11 * C; i += blockDim.x) wl[i] = i < 9 * C ? w[i] : (i < 10 * C ? scale[i - 9 * C] : shift[i - 10 * C]);
    __syncthreads();
    const int cv = C / VE, xr = (W + XR - 1) / XR;
    const int64_t total = (int64_t)B * H * xr * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VE;
        const int64_t r = i / cv;
        const int xb = (int)(r % xr) * XR, y = (int)((r / xr) % H), b = (int)(r / ((int64_t)xr * H));
        float s[XR][VE];
#pragma unroll
        for (int q = 0; q < XR; ++q)
#pragma unroll
            for (int e = 0; e < VE; ++e) s[q][e] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = y - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
            const T* row = in + ((int64_t)b * H + iy) * W * ips + c;
            Vec16<T> v[XR + 2];
#pragma unroll
            for (int j = 0; j < XR + 2; ++j) {
                const int ix = xb - 1 + j;
                if ((unsigned)ix < (unsigned)W) v[j].raw = *(const i32x4*)(row + (int64_t)ix * ips);
                else v[j].raw = i32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float wt[VE];
#pragma unroll
                for (int e4 = 0; e4 < VE; e4 += 4) {
                    const f32x4 t4 = *(const f32x4*)(wl + (dy * 3 + dx) * C + c + e4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) wt[e4 + e] = t4[e];
                }
#pragma unroll
                for (int q = 0; q < XR; ++q) {
                    // (a tap left / right of the image holds zeros: fmaf(0, w, s) == s, the old kernel skipped it)
#pragma unroll
                    for (int e = 0; e < VE; ++e) s[q][e] = fmaf(v[q + dx].get(e), wt[e], s[q][e]);
                }
            }
        }
        float sc[VE], sh[VE];
#pragma unroll
        for (int e4 = 0; e4 < VE; e4 += 4) {
            const f32x4 a4 = *(const f32x4*)(wl + 9 * C + c + e4), b4 = *(const f32x4*)(wl + 10 * C + c + e4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[e4 + e] = a4[e]; sh[e4 + e] = b4[e]; }
        }
        T* orow = out + (((int64_t)b * H + y) * W + xb) * ops + c;
#pragma unroll
        for (int q = 0; q < XR; ++q) {
            if (xb + q >= W) break;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float v1 = s[q][e] * sc[e] + sh[e];
                s[q][e] = relu ? fmaxf(v1, 0.f) : v1;
            }
            Vec16<T> o;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o.set2(e, s[q][2 * e], s[q][2 * e + 1]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o.set(e, s[q][e]);
            }
            *(i32x4*)(orow + (int64_t)q * ops) = o.raw;
        }
    }
}

template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t n_pix, int C, int ips, int ops) {
    constexpr int VE = ElemTraits<T>::kVec;
    const int cv = C / VE;
    const int64_t total = n_pix * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VE;
        const int64_t pix = i / cv;
        *(i32x4*)(out + pix * ops + c) = *(const i32x4*)(in + pix * ips + c);
    }
}

// boundary converters (scalar per element; only used at the module boundary / in tests)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C, int ips) {
    const int64_t total = (int64_t)B * C * H * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((int64_t)W * H)) % C), b = (int)(i / ((int64_t)W * H * C));
        out[i] = ElemTraits<T>::to_f(in[(((int64_t)b * H + y) * W + x) * ips + c]);
    }
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C, int ops) {
    const int64_t total = (int64_t)B * C * H * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        out[pix * ops + c] = ElemTraits<T>::from_f(in[(((int64_t)b * C + c) * H + y) * W + x]);
    }
}

inline int grid_for(int64_t total) {
    int64_t g = (total + kThreads - 1) / kThreads;
    return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}
inline bool vec_ok(int dtype, int C, int s1, int s2, const void* a, const void* b) {
    const int ve = dtype == VD3D_F32 ? 4 : 8;
    return (C % ve == 0) && (s1 % ve == 0) && (s2 % ve == 0) && (((uintptr_t)a & 15) == 0) && (((uintptr_t)b & 15) == 0);
}
#define VD3D_DISPATCH(dtype, ...)                         \
    if ((dtype) == VD3D_BF16) { using T = short; __VA_ARGS__; } \
    else if ((dtype) == VD3D_F16) { using T = hf16; __VA_ARGS__; } \
    else if ((dtype) == VD3D_F32) { using T = float; __VA_ARGS__; } \
    else { vd3d_set_error("bad dtype"); return VD3D_EINVAL; }

}  // namespace

extern "C" int vd3d_pack_image_nhwc4(const float* in, void* out, int B, int H, int W, int pad_y, int pad_l, int pad_r,
                                     int dtype, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0) { vd3d_set_error("pack_image: bad args"); return VD3D_EINVAL; }
    const int Hp = H + 2 * pad_y, Wp = W + pad_l + pad_r;
    const int64_t total = (int64_t)B * Hp * Wp;
    VD3D_DISPATCH(dtype, hipLaunchKernelGGL(pack_image_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                             in, (T*)out, B, H, W, pad_y, pad_l, Hp, Wp));
    return vd3d_check_launch("pack_image");
}

extern "C" int vd3d_maxpool3x3s2(const void* in, void* out, int B, int H, int W, int C, int ips, int ops, int dtype, void* stream) {
    if (!vec_ok(dtype, C, ips, ops, in, out)) { vd3d_set_error("maxpool: channels/strides must be 16-byte multiples"); return VD3D_EINVAL; }
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)B * Ho * Wo * (C / (dtype == VD3D_F32 ? 4 : 8));
    VD3D_DISPATCH(dtype, hipLaunchKernelGGL(maxpool3x3s2_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                             (const T*)in, (T*)out, B, H, W, C, Ho, Wo, ips, ops));
    return vd3d_check_launch("maxpool3x3s2");
}

extern "C" int vd3d_avgpool2x2(const void* in, void* out, int B, int H, int W, int C, int ips, int ops, int dtype, void* stream) {
    if (!vec_ok(dtype, C, ips, ops, in, out)) { vd3d_set_error("avgpool: channels/strides must be 16-byte multiples"); return VD3D_EINVAL; }
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * (C / (dtype == VD3D_F32 ? 4 : 8));
    VD3D_DISPATCH(dtype, hipLaunchKernelGGL(avgpool2x2_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                             (const T*)in, (T*)out, B, H, W, C, Ho, Wo, ips, ops));
    return vd3d_check_launch("avgpool2x2");
}

extern "C" int vd3d_dwconv3x3(const void* in, const float* weight, const float* scale, const float* shift, void* out,
                              int B, int H, int W, int C, int ips, int ops, int relu, int dtype, void* stream) {
    if (!vec_ok(dtype, C, ips, ops, in, out) || !weight || !scale || !shift) { vd3d_set_error("dwconv3x3: bad args"); return VD3D_EINVAL; }
    const int ve = dtype == VD3D_F32 ? 4 : 8;
    if (C <= 1024 && !vd3d_switch(VD3D_SW_DWCONV_PLAIN)) {
        // runs of four pixels per thread (VD3D_DWCONV_PLAIN=1: the one-pixel kernel, bit-identical; A/B)
        constexpr int XR = 4;
        const int64_t total = (int64_t)B * H * ((W + XR - 1) / XR) * (C / ve);
        VD3D_DISPATCH(dtype, hipLaunchKernelGGL((dwconv3x3_run_kernel<T, XR>), dim3(grid_for(total)), dim3(kThreads), 11 * C * sizeof(float), (hipStream_t)stream,
                                                 (const T*)in, weight, scale, shift, (T*)out, B, H, W, C, ips, ops, relu));
        return vd3d_check_launch("dwconv3x3");
    }
    const int64_t total = (int64_t)B * H * W * (C / ve);
    VD3D_DISPATCH(dtype, hipLaunchKernelGGL(dwconv3x3_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                             (const T*)in, weight, scale, shift, (T*)out, B, H, W, C, ips, ops, relu));
    return vd3d_check_launch("dwconv3x3");
}

extern "C" int vd3d_copy_channels(const void* in, void* out, int64_t n_pix, int C, int ips, int ops, int dtype, void* stream) {
    if (!vec_ok(dtype, C, ips, ops, in, out)) { vd3d_set_error("copy_channels: channels/strides must be 16-byte multiples"); return VD3D_EINVAL; }
    const int64_t total = n_pix * (C / (dtype == VD3D_F32 ? 4 : 8));
    VD3D_DISPATCH(dtype, hipLaunchKernelGGL(copy_channels_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                             (const T*)in, (T*)out, n_pix, C, ips, ops));
    return vd3d_check_launch("copy_channels");
}

extern "C" int vd3d_nhwc_to_nchw_f32(const void* in, float* out, int B, int H, int W, int C, int ips, int dtype, void* stream) {
    const int64_t total = (int64_t)B * H * W * C;
    VD3D_DISPATCH(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                             (const T*)in, out, B, H, W, C, ips));
    return vd3d_check_launch("nhwc_to_nchw");
}
extern "C" int vd3d_nchw_f32_to_nhwc(const float* in, void* out, int B, int H, int W, int C, int ops, int dtype, void* stream) {
    const int64_t total = (int64_t)B * H * W * C;
    VD3D_DISPATCH(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                             in, (T*)out, B, H, W, C, ops));
    return vd3d_check_launch("nchw_to_nhwc");
}

// ---------------------------------------------------------------------------------------------------
// LookGround sampling (lib/look_ground.py:24-69): for each output pixel sample [x ; prior disparity] at
// (x, y + y_shift) with grid_sample(bilinear, border, align_corners=True) semantics.  The x coordinate of the flow
// field is the pixel's own column (x_base is the identity grid), so the gather is a 2-row vertical lerp: coalesced
// 16-byte channel vectors from two rows.  Output layout: channels [0, C) = sampled x, channel C = sampled prior
// disparity, channels (C, Cpad) = 0 (the 1x1 `extract` conv weight is packed in the same order).
namespace {
template <typename T>
__global__ void look_ground_kernel(const T* __restrict__ x, const float* __restrict__ disp_raw, const float* __restrict__ P2,
                                   T* __restrict__ out, int B, int H, int W, int C, int Cpad, int ips, int ops,
                                   float baseline, float elevation) {
    constexpr int VE = ElemTraits<T>::kVec;
    const int cv = Cpad / VE;
    const int64_t total = (int64_t)B * H * W * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VE;
        const int64_t pix = i / cv;
        const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        const float fy = P2[b * 12 + 5] / 16.0f, cy = P2[b * 12 + 6] / 16.0f, Ty = P2[b * 12 + 7] / 16.0f;
        // learned offset: 0.1 * (0.05 * d + 0.95 * d), d = tanh(conv)
        const float d = tanhf(disp_raw[pix]);
        const float disp = 0.1f * (0.05f * d + 0.95f * d);
        const float h_mean = 1.535f;
        const float ysb = fmaxf(h_mean * ((float)h - cy) / (2.0f * (elevation - 0.5f * h_mean)), 0.0f) / ((float)H * 0.5f);
        const float y_base = H > 1 ? -1.0f + 2.0f * (float)h / (float)(H - 1) : -1.0f;
        const float gy = y_base + (ysb + disp);
        float iy = (gy + 1.0f) / 2.0f * (float)(H - 1);
        iy = fminf(fmaxf(iy, 0.0f), (float)(H - 1));          // padding_mode='border'
        const int y0 = (int)floorf(iy);
        const int y1 = min(y0 + 1, H - 1);
        const float wy1 = iy - (float)y0, wy0 = 1.0f - wy1;
        float v[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) v[e] = 0.f;
        if (c < C) {
            Vec16<T> a, bb;
            a.raw = *(const i32x4*)(x + (((int64_t)b * H + y0) * W + w) * ips + c);
            bb.raw = *(const i32x4*)(x + (((int64_t)b * H + y1) * W + w) * ips + c);
#pragma unroll
            for (int e = 0; e < VE; ++e) v[e] = a.get(e) * wy0 + bb.get(e) * wy1;
        } else if (c == C) {
            // prior disparity of a ground point seen at row r: relu(fy * baseline * (r - cy) / (|fy * elev + Ty| + 1e-10))
            const float den = fabsf(fy * elevation + Ty) + 1e-10f;
            const float p0 = fmaxf(fy * baseline * ((float)y0 - cy) / den, 0.0f);
            const float p1 = fmaxf(fy * baseline * ((float)y1 - cy) / den, 0.0f);
            v[0] = p0 * wy0 + p1 * wy1;
        }
        Vec16<T> o;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, v[2 * e], v[2 * e + 1]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, v[e]);
        }
        *(i32x4*)(out + pix * ops + c) = o.raw;
    }
}
}  // namespace

extern "C" int vd3d_look_ground_sample(const void* x, const float* disp, const float* P2s, void* out, int B, int H, int W,
                                       int C, int ips, int ops, float baseline, float elevation, int dtype, void* stream) {
    const int ve = dtype == VD3D_F32 ? 4 : 8;
    const int Cpad = (C + 1 + ve - 1) / ve * ve;
    if (!x || !disp || !P2s || !out || C % ve || ips % ve || ops % ve || ops < Cpad || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) {
        vd3d_set_error("look_ground_sample: C, strides must be 16-byte multiples; out needs round_up(C+1) channels");
        return VD3D_EINVAL;
    }
    const int64_t total = (int64_t)B * H * W * (Cpad / ve);
    VD3D_DISPATCH(dtype, hipLaunchKernelGGL(look_ground_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                             (const T*)x, disp, P2s, (T*)out, B, H, W, C, Cpad, ips, ops, baseline, elevation));
    return vd3d_check_launch("look_ground_sample");
}

// ---------------------------------------------------------------------------------------------------
// DLA helpers (backbones/dla.py, dla_utils.py): 2x2/s2 max-pool (Tree.downsample), depth-wise ConvTranspose2d
// (IDAUp.up_i: kernel 2f, stride f, pad f/2, groups = C, no bias) fused with the following `+ layers[i-1]`.
namespace {
template <typename T>
__global__ void maxpool2x2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C, int Ho, int Wo, int ips, int ops) {
    constexpr int VE = ElemTraits<T>::kVec;
    const int cv = C / VE;
    const int64_t total = (int64_t)B * Ho * Wo * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VE;
        const int64_t pix = i / cv;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((int64_t)Wo * Ho));
        float m[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                Vec16<T> v;
                v.raw = *(const i32x4*)(in + (((int64_t)b * H + oy * 2 + dy) * W + ox * 2 + dx) * ips + c);
#pragma unroll
                for (int e = 0; e < VE; ++e) m[e] = fmaxf(m[e], v.get(e));
            }
        Vec16<T> o;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, m[2 * e], m[2 * e + 1]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, m[e]);
        }
        *(i32x4*)(out + pix * ops + c) = o.raw;
    }
}

// out[b,y,x,c] = sum_{ky,kx} in[b,(y+pad-ky)/f,(x+pad-kx)/f,c] * w[ky*K+kx][c]  (terms with exact division, in range)
//                (+ add[b,y,x,c]);  weight layout [K*K][C] fp32
template <typename T>
__global__ void dwconvT_kernel(const T* __restrict__ in, const float* __restrict__ w, const T* __restrict__ add, T* __restrict__ out,
                               int B, int H, int W, int C, int f, int K, int pad, int Ho, int Wo, int ips, int aps, int ops) {
    constexpr int VE = ElemTraits<T>::kVec;
    const int cv = C / VE;
    const int64_t total = (int64_t)B * Ho * Wo * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VE;
        const int64_t pix = i / cv;
        const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho), b = (int)(pix / ((int64_t)Wo * Ho));
        float s[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) s[e] = 0.f;
        // torch's conv_transpose accumulates input-major; the order of the (at most ceil(K/f)^2) terms differs only in
        // fp32 round-off
        for (int ky = (y + pad) % f; ky < K; ky += f) {
            const int iy = (y + pad - ky) / f;
            if (y + pad - ky < 0 || iy >= H) continue;
            for (int kx = (x + pad) % f; kx < K; kx += f) {
                const int ix = (x + pad - kx) / f;
                if (x + pad - kx < 0 || ix >= W) continue;
                Vec16<T> v;
                v.raw = *(const i32x4*)(in + (((int64_t)b * H + iy) * W + ix) * ips + c);
                const float* wt = w + (ky * K + kx) * C + c;
#pragma unroll
                for (int e = 0; e < VE; ++e) s[e] = fmaf(v.get(e), wt[e], s[e]);
            }
        }
        Vec16<T> o;
        if constexpr (sizeof(T) == 2) {
            // the reference rounds the up-sampled map (a tensor) before adding: mimic (bf16 path) so rounding points match
            if (add) {
                Vec16<T> a; a.raw = *(const i32x4*)(add + pix * aps + c);
#pragma unroll
                for (int e = 0; e < VE; ++e) s[e] = ElemTraits<T>::to_f(ElemTraits<T>::from_f(s[e])) + a.get(e);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, s[2 * e], s[2 * e + 1]);
        } else {
            if (add) {
                Vec16<T> a; a.raw = *(const i32x4*)(add + pix * aps + c);
#pragma unroll
                for (int e = 0; e < VE; ++e) s[e] += a.get(e);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set(e, s[e]);
        }
        *(i32x4*)(out + pix * ops + c) = o.raw;
    }
}

// The same op for the IDA-Up shapes of the DLA networks (F = 2 | 4, C a power of two, 16-bit): a workgroup is F*F waves, wave = output
// PHASE (y mod F, x mod F) -- every output of one phase uses the same 2 x 2 kernel taps, so a lane keeps its four weight vectors
// (its 8 channels) in registers for the whole launch; the F*F waves walk the same input row segment at the same time (the 2 x 2 input
// neighbourhoods come from the CU's L1), rows are handed out per workgroup (no per-element 64-bit div / mod: the generic kernel
// spends most of its time there).  Term order and rounding points are the generic kernel's: results are bit-identical.
template <typename T, int F>
__global__ void __launch_bounds__(64 * F * F) dwconvT_phase_kernel(const T* __restrict__ in, const float* __restrict__ w, const T* __restrict__ add,
                                                                   T* __restrict__ out, int B, int H, int W, int C, int lcv, int ips, int aps, int ops) {
    constexpr int K = 2 * F, HALF = F / 2;
    const int lane = threadIdx.x & 63, phase = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int py = phase / F, px = phase % F;
    const int c = (lane & ((1 << lcv) - 1)) * 8, pl = lane >> lcv, ppw = 64 >> lcv;     // this lane's channels; pixels per wave and step
    // output (F j + py, F i + px) <- inputs (ja, ia), (ja, ia - 1), (ja - 1, ia), (ja - 1, ia - 1) with taps (ky0 | ky0 + F) x (kx0 | kx0 + F)
    const int dy = py >= HALF, dx = px >= HALF;
    const int ky0 = dy ? py - HALF : py + HALF, kx0 = dx ? px - HALF : px + HALF;
    float wt[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float* wp = w + ((ky0 + (t >> 1) * F) * K + kx0 + (t & 1) * F) * C + c;
#pragma unroll
        for (int e = 0; e < 8; ++e) wt[t][e] = wp[e];
    }
    const int Ho = F * H, Wo = F * W;
    for (int row = blockIdx.x; row < B * H; row += gridDim.x) {
        const int b = row / H, j = row - b * H;
        const int ja = j + dy;
        const bool ra = ja < H, rb = ja >= 1;                          // wave-uniform: the two input rows exist
        const T* rowa = in + (int64_t)((b * H + ja) * W) * ips + c;
        const T* rowb = rowa - (int64_t)W * ips;
        const int64_t obase = (int64_t)((b * Ho + F * j + py) * Wo + px);
        for (int i = pl + blockIdx.y * ppw; i < W; i += ppw * gridDim.y) {      // blockIdx.y: column split of short launches
            const int ia = i + dx;
            const bool ca = ia < W, cb = ia >= 1;
            const i32x4 zero = {0, 0, 0, 0};
            Vec16<T> v[4];
            v[0].raw = ra && ca ? *(const i32x4*)(rowa + (int64_t)ia * ips) : zero;
            v[1].raw = ra && cb ? *(const i32x4*)(rowa + (int64_t)(ia - 1) * ips) : zero;
            v[2].raw = rb && ca ? *(const i32x4*)(rowb + (int64_t)ia * ips) : zero;
            v[3].raw = rb && cb ? *(const i32x4*)(rowb + (int64_t)(ia - 1) * ips) : zero;
            const int64_t opix = obase + F * i;
            Vec16<T> a;
            if (add) a.raw = *(const i32x4*)(add + opix * aps + c);
            float s[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] = fmaf(v[t].get(e), wt[t][e], s[e]);
            if (add) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] = ElemTraits<T>::to_f(ElemTraits<T>::from_f(s[e])) + a.get(e);
            }
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.set2(e, s[2 * e], s[2 * e + 1]);
            *(i32x4*)(out + opix * ops + c) = o.raw;
        }
    }
}

// NCHW fp32 image -> zero-bordered NHWC with `cpad` channels (3 real + zeros), general borders
template <typename T, int CPAD>
__global__ void pack_image_c_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int H, int W, int pad_y, int pad_l, int Hp, int Wp) {
    const int64_t total = (int64_t)B * Hp * Wp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xp = (int)(i % Wp), yp = (int)((i / Wp) % Hp), b = (int)(i / ((int64_t)Wp * Hp));
        const int x = xp - pad_l, y = yp - pad_y;
        float v[3] = {0.f, 0.f, 0.f};
        if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
            const int64_t base = ((int64_t)b * 3 * H + y) * W + x;
            v[0] = in[base]; v[1] = in[base + (int64_t)H * W]; v[2] = in[base + 2 * (int64_t)H * W];
        }
        T* o = out + i * CPAD;
#pragma unroll
        for (int c = 0; c < CPAD; ++c) o[c] = ElemTraits<T>::from_f(c < 3 ? v[c] : 0.f);
    }
}
}  // namespace

extern "C" int vd3d_maxpool2x2(const void* in, void* out, int B, int H, int W, int C, int ips, int ops, int dtype, void* stream) {
    if (!vec_ok(dtype, C, ips, ops, in, out)) { vd3d_set_error("maxpool2x2: channels/strides must be 16-byte multiples"); return VD3D_EINVAL; }
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * (C / (dtype == VD3D_F32 ? 4 : 8));
    VD3D_DISPATCH(dtype, hipLaunchKernelGGL(maxpool2x2_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                             (const T*)in, (T*)out, B, H, W, C, Ho, Wo, ips, ops));
    return vd3d_check_launch("maxpool2x2");
}

extern "C" int vd3d_dwconv_transpose(const void* in, const float* weight, const void* add, void* out, int B, int H, int W, int C,
                                     int f, int ips, int aps, int ops, int dtype, void* stream) {
    if (!vec_ok(dtype, C, ips, ops, in, out) || !weight || f < 1 || (add && (((uintptr_t)add & 15) || aps % (dtype == VD3D_F32 ? 4 : 8)))) {
        vd3d_set_error("dwconv_transpose: bad args"); return VD3D_EINVAL;
    }
    const int K = 2 * f, pad = f / 2;
    const int Ho = (H - 1) * f - 2 * pad + K, Wo = (W - 1) * f - 2 * pad + K;
    const int64_t total = (int64_t)B * Ho * Wo * (C / (dtype == VD3D_F32 ? 4 : 8));
    // IDA-Up shapes: the phase kernel (weights in registers, no per-element div / mod); a skipped out-of-range term adds exactly zero
    // there, so a non-finite weight is the one input on which the two kernels could differ -- the weights are parameters, finite.
    if (dtype != VD3D_F32 && (f == 2 || f == 4) && C >= 8 && C <= 512 && (C & (C - 1)) == 0 && (int64_t)B * Ho * Wo < (1 << 30) &&
        (int64_t)B * H * W * ips < (1ll << 31) && !vd3d_switch(VD3D_SW_DWCONVT_GENERIC)) {
        int lcv = 0;
        while ((8 << lcv) < C) ++lcv;
        const int rows = B * H, cus = vd3d_device_cu_count();
        if (cus <= 0) return VD3D_ELAUNCH;
        const int grid = rows < 8 * cus ? rows : 8 * cus;
        const int steps = (W + (64 >> lcv) - 1) / (64 >> lcv);                 // wave steps along a row
        int nsplit = 4 * cus / grid;                                           // few rows (deep levels): split the columns as well
        nsplit = nsplit < 1 ? 1 : (nsplit > steps ? steps : nsplit);
#define VD3D_DWT(TT, FF) hipLaunchKernelGGL((dwconvT_phase_kernel<TT, FF>), dim3(grid, nsplit), dim3(64 * FF * FF), 0, (hipStream_t)stream, (const TT*)in, weight, \
                                             (const TT*)add, (TT*)out, B, H, W, C, lcv, ips, aps, ops)
        if (dtype == VD3D_F16) { if (f == 2) VD3D_DWT(hf16, 2); else VD3D_DWT(hf16, 4); }
        else { if (f == 2) VD3D_DWT(short, 2); else VD3D_DWT(short, 4); }
#undef VD3D_DWT
        return vd3d_check_launch("dwconv_transpose(phase)");
    }
    VD3D_DISPATCH(dtype, hipLaunchKernelGGL(dwconvT_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                             (const T*)in, weight, (const T*)add, (T*)out, B, H, W, C, f, K, pad, Ho, Wo, ips, aps, ops));
    return vd3d_check_launch("dwconv_transpose");
}

extern "C" int vd3d_pack_image_nhwc(const float* in, void* out, int B, int H, int W, int pad_y0, int pad_y1, int pad_l, int pad_r,
                                    int cpad, int dtype, void* stream) {
    if (!in || !out || (cpad != 4 && cpad != 8)) { vd3d_set_error("pack_image_nhwc: cpad must be 4 or 8"); return VD3D_EINVAL; }
    const int Hp = H + pad_y0 + pad_y1, Wp = W + pad_l + pad_r;
    const int64_t total = (int64_t)B * Hp * Wp;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VD3D_BF16 && cpad == 8) hipLaunchKernelGGL((pack_image_c_kernel<short, 8>), dim3(grid_for(total)), dim3(kThreads), 0, s, in, (short*)out, B, H, W, pad_y0, pad_l, Hp, Wp);
    else if (dtype == VD3D_BF16) hipLaunchKernelGGL((pack_image_c_kernel<short, 4>), dim3(grid_for(total)), dim3(kThreads), 0, s, in, (short*)out, B, H, W, pad_y0, pad_l, Hp, Wp);
    else if (dtype == VD3D_F16 && cpad == 8) hipLaunchKernelGGL((pack_image_c_kernel<hf16, 8>), dim3(grid_for(total)), dim3(kThreads), 0, s, in, (hf16*)out, B, H, W, pad_y0, pad_l, Hp, Wp);
    else if (dtype == VD3D_F16) hipLaunchKernelGGL((pack_image_c_kernel<hf16, 4>), dim3(grid_for(total)), dim3(kThreads), 0, s, in, (hf16*)out, B, H, W, pad_y0, pad_l, Hp, Wp);
    else if (dtype == VD3D_F32 && cpad == 8) hipLaunchKernelGGL((pack_image_c_kernel<float, 8>), dim3(grid_for(total)), dim3(kThreads), 0, s, in, (float*)out, B, H, W, pad_y0, pad_l, Hp, Wp);
    else if (dtype == VD3D_F32) hipLaunchKernelGGL((pack_image_c_kernel<float, 4>), dim3(grid_for(total)), dim3(kThreads), 0, s, in, (float*)out, B, H, W, pad_y0, pad_l, Hp, Wp);
    else { vd3d_set_error("bad dtype"); return VD3D_EINVAL; }
    return vd3d_check_launch("pack_image_nhwc");
}
