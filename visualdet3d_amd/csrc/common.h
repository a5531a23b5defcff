// common.h -- shared device/host helpers for libvd3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vd3d.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;  // 8 bf16 = 16 B (MFMA A/B fragment)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(2))) short bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

#define VD3D_DEV __device__ __forceinline__

// bf16 <-> f32.  gfx950 has v_cvt_pk_bf16_f32 (RNE); clang emits it for __bf16 conversions.
// NB: __builtin_bit_cast applied directly to an ext-vector ELEMENT lvalue (v[i]) reads lane 0 of the vector with
// this compiler; always go through a by-value helper.
VD3D_DEV float i2f(int v) { return __builtin_bit_cast(float, v); }
VD3D_DEV int f2i(float v) { return __builtin_bit_cast(int, v); }
VD3D_DEV float bf2f(short v) { return __builtin_bit_cast(float, ((uint32_t)(uint16_t)v) << 16); }
VD3D_DEV short f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(short, b);
}

// fp16 storage: `hf16` (unsigned short) is the element-type TAG of half-precision tensors, as `short` is the tag of bf16 ones
typedef unsigned short hf16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
VD3D_DEV float h2f(hf16 v) { return (float)__builtin_bit_cast(_Float16, v); }
VD3D_DEV hf16 f2h(float f) {
    _Float16 h = (_Float16)f;        // v_cvt_f16_f32, round to nearest even
    return __builtin_bit_cast(hf16, h);
}

// The two 16-bit storage formats behind one interface: pack / unpack and the MFMA flavours (same register images and
// instruction rates: v_mfma_f32_32x32x16_{bf16,f16}, v_mfma_f32_16x16x32_{bf16,f16})
template <typename T> struct Fmt16;
template <> struct Fmt16<short> {
    static VD3D_DEV int pack2(float lo, float hi) { return (int)((uint32_t)(uint16_t)f2bf(lo) | ((uint32_t)(uint16_t)f2bf(hi) << 16)); }
    // the same conversion as pack2 pinned to ONE instruction (the compiler turns pack2 followed by integer ops on the pair into two
    // conversions + v_perm_b32)
    static VD3D_DEV int pack2_1(float lo, float hi) { int r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
    static VD3D_DEV float lo(uint32_t u) { return i2f((int)(u << 16)); }
    static VD3D_DEV float hi(uint32_t u) { return i2f((int)(u & 0xffff0000u)); }
    static VD3D_DEV short one(float f) { return f2bf(f); }
    static VD3D_DEV float tof(short v) { return bf2f(v); }
    static VD3D_DEV void mfma32(const i32x4& a, const i32x4& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    static VD3D_DEV void mfma16(const i32x4& a, const i32x4& b, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    static VD3D_DEV void mfma32z(const i32x4& a, const i32x4& b, const f32x16& c, f32x16& acc) {     // acc = a x b + c (c: a constant)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Fmt16<hf16> {
    static VD3D_DEV int pack2(float lo, float hi) {
        f16x2 p = {(_Float16)lo, (_Float16)hi};          // v_cvt_pk_f16_f32 (RNE)
        return __builtin_bit_cast(int, p);
    }
    static VD3D_DEV int pack2_1(float lo, float hi) { int r; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
    static VD3D_DEV float lo(uint32_t u) { return h2f((hf16)(u & 0xffffu)); }
    static VD3D_DEV float hi(uint32_t u) { return h2f((hf16)(u >> 16)); }
    static VD3D_DEV hf16 one(float f) { return f2h(f); }
    static VD3D_DEV float tof(hf16 v) { return h2f(v); }
    static VD3D_DEV void mfma32(const i32x4& a, const i32x4& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    static VD3D_DEV void mfma16(const i32x4& a, const i32x4& b, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    static VD3D_DEV void mfma32z(const i32x4& a, const i32x4& b, const f32x16& c, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// ReLU on a packed pair of 16-bit floats (bf16 or fp16 alike): a negative value has its sign bit set, i.e. is a negative int16 --
// v_pk_max_i16 with 0 clears it (-0 -> +0 as fmaxf does); relu(round(x)) == round(relu(x)) since rounding keeps the sign.  One
// instruction per TWO values where fmaxf on an MFMA result costs two per value (canonicalise + max).  (NaN: a positive NaN stays NaN.)
typedef __attribute__((ext_vector_type(2))) short vd3d_s16x2;
// (floor = 0x00000000: ReLU;  0x80008000 = two INT16_MIN: identity -- a wave-uniform "relu or not" without a branch)
VD3D_DEV int max_pk16(int p, int floor) {
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(vd3d_s16x2, p), __builtin_bit_cast(vd3d_s16x2, floor)));
}
VD3D_DEV int relu_pk16(int p) {
    const vd3d_s16x2 z = {0, 0};
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(vd3d_s16x2, p), z));
}

template <> struct Fmt16<float> {      // never used for arithmetic: lets 16-bit-only epilogue code compile in fp32 instantiations
    static VD3D_DEV int pack2(float lo, float hi) { return Fmt16<short>::pack2(lo, hi); }
    static VD3D_DEV int pack2_1(float lo, float hi) { return Fmt16<short>::pack2_1(lo, hi); }
    static VD3D_DEV float lo(uint32_t u) { return Fmt16<short>::lo(u); }
    static VD3D_DEV float hi(uint32_t u) { return Fmt16<short>::hi(u); }
    static VD3D_DEV float one(float f) { return f; }
    static VD3D_DEV float tof(float v) { return v; }
};

template <typename T> struct ElemTraits;
template <> struct ElemTraits<short> {  // bf16 storage
    static constexpr int kVec = 8;      // elements per 16-byte vector
    static VD3D_DEV float to_f(short v) { return bf2f(v); }
    static VD3D_DEV short from_f(float f) { return f2bf(f); }
};
template <> struct ElemTraits<hf16> {   // fp16 storage
    static constexpr int kVec = 8;
    static VD3D_DEV float to_f(hf16 v) { return h2f(v); }
    static VD3D_DEV hf16 from_f(float f) { return f2h(f); }
};
template <> struct ElemTraits<float> {
    static constexpr int kVec = 4;
    static VD3D_DEV float to_f(float v) { return v; }
    static VD3D_DEV float from_f(float f) { return f; }
};

// 16-byte vector of T <-> floats
template <typename T> struct Vec16;
template <> struct Vec16<short> {
    i32x4 raw;
    VD3D_DEV float get(int i) const {
        const int wi = raw[i >> 1];
        const uint32_t w = (uint32_t)wi;
        return i2f((int)((i & 1) ? (w & 0xffff0000u) : (w << 16)));
    }
    VD3D_DEV void set2(int pair, float lo, float hi) {
        uint32_t l = (uint16_t)f2bf(lo), h = (uint16_t)f2bf(hi);
        raw[pair] = (int)(l | (h << 16));
    }
};
template <> struct Vec16<hf16> {
    i32x4 raw;
    VD3D_DEV float get(int i) const {
        const uint32_t w = (uint32_t)(int)raw[i >> 1];
        return (i & 1) ? Fmt16<hf16>::hi(w) : Fmt16<hf16>::lo(w);
    }
    VD3D_DEV void set2(int pair, float lo, float hi) { raw[pair] = Fmt16<hf16>::pack2(lo, hi); }
};
template <> struct Vec16<float> {
    i32x4 raw;
    VD3D_DEV float get(int i) const { const int w = raw[i]; return i2f(w); }
    VD3D_DEV void set(int i, float v) { raw[i] = f2i(v); }
};

// host-side error plumbing ------------------------------------------------------------------------
void vd3d_set_error(const char* msg);
int vd3d_check_launch(const char* what);

// Exact n / d for 0 <= n < 2^31 by one 32 x 32 -> 64-bit multiply (d fixed per launch): mul = ceil(2^(31 + s) / d), s = ceil(log2 d).
// (A per-lane integer division is ~40 VALU instructions on gfx950; a tile prologue has two per staged pixel row.)
struct FastDiv {
    uint32_t mul = 0x80000000u, shift = 0;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    while ((1u << f.shift) < d) ++f.shift;
    f.mul = (uint32_t)(((1ull << (31 + f.shift)) + d - 1) / d);
    return f;
}
__device__ __forceinline__ int fastdiv(int n, FastDiv f) { return (int)((uint32_t)(((uint64_t)(uint32_t)n * f.mul) >> 31) >> f.shift); }


// A/B switches between two CORRECT implementations.  Each is the environment variable of the same name, read ONCE
// per process (no getenv on the launch path, no race with setenv); tests flip them through vd3d_test_set_switch (test_hooks.h).
enum Vd3dSwitch {
    VD3D_SW_CONV_DEBUG, VD3D_SW_FORCE_GROUP_M, VD3D_SW_NO_GROUP_M, VD3D_SW_NO_LINE_STORE, VD3D_SW_DCN_GENERIC,
    VD3D_SW_DCN_COLUMNS_GENERIC, VD3D_SW_CONV3D_VALU, VD3D_SW_PSM_VALU, VD3D_SW_NO_NARROW, VD3D_SW_DWCONVT_GENERIC, VD3D_SW_HEAD_PARKED, VD3D_SW_STEM_WG4, VD3D_SW_HEAD_NO_STAGGER, VD3D_SW_CONV_NO_STAGGER, VD3D_SW_DCN_NO_LSTAGE, VD3D_SW_NO_STRIP_SPLIT, VD3D_SW_PLAIN_TILE_WALK, VD3D_SW_DWCONV_PLAIN, VD3D_SW_COUNT
};
bool vd3d_switch(Vd3dSwitch s);

static inline int vd3d_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Per-DEVICE launch bookkeeping.  hipFuncSetAttribute acts on the current device's copy of a kernel, so "dynamic-LDS limit
// already raised" must be remembered per device ordinal, not per process (one process may drive several GPUs).
constexpr int kVd3dMaxDevices = 64;
struct Vd3dLdsLimit { int bytes[kVd3dMaxDevices]; };          // zero-initialise (static storage)
int vd3d_current_device();                                      // ordinal of the current device, or -1 (error string set)
int vd3d_device_cu_count();                                     // compute units of the current device (cached per device), or -1
// raise `kern`'s dynamic shared memory limit to >= bytes on the current device (no-op if already done there)
int vd3d_raise_lds_limit(const void* kern, int bytes, Vd3dLdsLimit& state, const char* what);
