// stem_pool.hip -- ResNet stem fused: 7x7/s2 conv + BN + ReLU + 3x3/s2/p1 max pool in one persistent kernel (bf16, gfx950).
//
// Replaces conv1 / bn1 / relu / maxpool of backbones/resnet.py:118-121,187-190 when only the pooled map is consumed.
// The unfused path writes the 64-channel half-resolution map (252 MB at 16 x 384 x 1280) and reads it back for the pool,
// and its implicit-GEMM tiles re-fetch every input pixel ~25 times through L2.  Here a workgroup owns 8 x 16 POOLED
// pixels: it stages the 39 x 72-pixel patch of the packed NHWC4 image once (22 KiB, LDS-DMA, 3-deep ring), computes the
// 17 x 33 conv outputs the pool windows need as 18 blocks of 32 pixels x 64 channels on the MFMA pipe (K = 7 rows x 32:
// 14 v_mfma_f32_32x32x16_bf16 per block and channel half, weights resident in registers), applies BN + ReLU, parks the
// bf16 results in LDS, and the pool phase takes 3x3 maxima (packed 16-bit integer max: post-ReLU bf16 are monotone as
// int16, and -0 loses) and writes 16 bytes per lane.  HBM traffic: packed image once + pooled map once.
#include "common.h"

#include <cstdlib>

namespace {

constexpr int kTPX = 16;                           // pooled tile width
constexpr uint32_t kOOB = 0x80000000u;
// Tile geometry for a pooled tile of TPY x 16 pixels.  TPY = 8, 8 waves: the LDS-DMA (packed-image) variant, one workgroup per CU.
// TPY = 4, 4 waves (round 4 experiment, opt-in VD3D_STEM_WG4=1): 51 KB of LDS, 180 VGPRs -> TWO independent workgroups per CU, so that one
// workgroup's epilogue / pool phase could run under the other's MFMAs.  Measured slower (launch_stem below).
template <int TPY>
struct StemGeo {
    static constexpr int kTPY = TPY;
    static constexpr int kCY = 2 * TPY + 1, kCX = 2 * kTPX + 1;   // conv outputs needed: 17 x 33 (9 x 33)
    static constexpr int kNQ = kCY * kCX;                         // 561 (297)
    static constexpr int kNBLK = (kNQ + 31) / 32;                 // 18 (10) blocks of 32 conv pixels
    static constexpr int kIR = 2 * (kCY - 1) + 7;                 // 39 (23) packed rows
    static constexpr int kCPR = (2 * (kCX - 1) + 8) / 2;          // 36 16-byte chunks (2 NHWC4 pixels) per packed row
    static constexpr int kChunks = kIR * kCPR;                    // 1404 (828)
    static constexpr int kPieces = (kChunks + 63) / 64;           // 22 (13) pieces of 1 KiB
    static constexpr int kInStage = kPieces * 1024;
    static constexpr int kStaging = kNQ * 128;                    // conv outputs: one 128-byte row (64 bf16) per conv pixel
};
template <int TPY, bool F32IN> constexpr int stem_lds_bytes() { return (F32IN ? 1 : 3) * StemGeo<TPY>::kInStage + StemGeo<TPY>::kStaging + 512; }

struct StemArgs {
    const float* img0; const float* img1; int B0;     // F32IN: the fp32 NCHW image(s): batch entries [0, B0) from img0, [B0, B) from img1
    int H, W;
    const char* packed;
    const char* weight;
    const float* scale;
    const float* shift;
    char* out;
    int B, Hp, Wp, Ho, Wo, Hq, Wq, Kpad, out_pix_stride;
    uint32_t in_bytes;
    int ntiles;
    int plain_walk;    // A/B: VD3D_PLAIN_TILE_WALK
};

// F32IN: the patch comes straight from the fp32 NCHW image (the reference's network input) -- converted to the NHWC4 16-bit LDS image
// in registers, one tile ahead -- instead of by LDS-DMA from a packed copy: the two pack_image launches (2 x 17 us, 94 MB read + 64 MB
// written + 64 MB re-read per step) disappear.
template <bool F32IN, int TPY = 8, int NW = 8>
__global__ void __launch_bounds__(NW * 64) stem_pool_kernel(const StemArgs p) {
    using G = StemGeo<TPY>;
    constexpr int kTPY = G::kTPY, kCX = G::kCX, kNQ = G::kNQ, kNBLK = G::kNBLK, kCPR = G::kCPR, kChunks = G::kChunks, kPieces = G::kPieces;
    constexpr int kInStage = G::kInStage, kStaging = G::kStaging;
    constexpr int kInStages = F32IN ? 1 : 3;       // fp32 image: the next patch is written after this tile's conv phase -- one stage
    constexpr int NT = NW * 64, WMI = NW / 2;      // waves = 2 channel halves x WMI block lanes
    constexpr int HP = (kPieces * 64 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* stg = smem + kInStages * kInStage;
    float* ss = (float*)(stg + kStaging);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wmi = wave >> 1;
    const int lr = lane & 31, half = lane >> 5;
    const int tiles_x = p.Wq / kTPX, tiles_y = p.Hq / kTPY, tiles_img = tiles_x * tiles_y;
    // XCD-aware walk (round 5): workgroups b, b + 8, ... share an XCD (round-robin dispatch) and with it an L2.  Each XCD owns a CONTIGUOUS
    // run of the row-major tile list and its workgroups take consecutive tiles of it in every round, so the patch columns neighbouring tiles
    // share (39 x 72 pixels fetched per 32 x 64 of new ones: 1.37 x) and the rows the next round shares are served by THAT L2 -- with the
    // plain walk t = b, b + grid, ... neighbours sat on different XCDs and every overlap was fetched twice from beyond L2 (FETCH_SIZE 222 MB
    // for a 94 MB image).  A grid that is not a multiple of 8 keeps the plain walk.
    int t_first = blockIdx.x, t_end = p.ntiles, nwg = gridDim.x;
    if ((gridDim.x & 7) == 0 && !p.plain_walk) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, q = p.ntiles >> 3, r = p.ntiles & 7;
        const int start = xcd * q + (xcd < r ? xcd : r);
        t_first = start + j;
        t_end = start + q + (xcd < r ? 1 : 0);
        nwg = gridDim.x >> 3;
    }

    if (tid < 64) {
        ss[tid] = p.scale ? p.scale[tid] : 1.f;
        ss[64 + tid] = p.shift ? p.shift[tid] : 0.f;
    }
    // weights -> registers (A operand): row = output channel, fragment (ky, ks) = 16 bytes at k = ky*32 + (2ks+half)*8
    i32x4 wf[14];
    {
        const char* wrow = p.weight + (size_t)(wn * 32 + lr) * p.Kpad * 2;
#pragma unroll
        for (int f = 0; f < 14; ++f) wf[f] = *(const i32x4*)(wrow + ((f >> 1) * 32 + (2 * (f & 1) + half) * 8) * 2);
    }
    // DMA lane state: chunk c = 64*piece + lane -> (row r, chunk j) of the patch; lane-linear LDS image (no swizzle needed:
    // a fragment read is 32 lanes x consecutive 16-byte chunks)
    int d_r[HP], d_j[HP], d_piece[HP];
    bool d_ok[HP];
#pragma unroll
    for (int it = 0; it < HP; ++it) {
        int piece = wave + it * NW;
        if (piece >= kPieces) piece -= NW;            // branch-free partial round: repeat the previous piece
        d_piece[it] = piece;
        const int c = 64 * piece + lane;
        d_r[it] = c / kCPR;
        d_j[it] = c - d_r[it] * kCPR;
        d_ok[it] = c < kChunks;
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.packed, 0, p.in_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto issue_patch = [&](int t, int stage) {
        const bool tv = t < t_end;
        const int tt = tv ? t : 0;
        const int b = tt / tiles_img, trem = tt - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int pr0 = 4 * ty * kTPY - 2, pc0 = 4 * tx * kTPX - 2;      // packed row / column of the patch origin
        char* base = smem + stage * kInStage;
#pragma unroll
        for (int it = 0; it < HP; ++it) {
            const int pr = pr0 + d_r[it], pc = pc0 + 2 * d_j[it];
            const bool v = tv && d_ok[it] && (unsigned)pr < (unsigned)p.Hp && pc >= 0 && pc + 1 < p.Wp;
            const uint32_t off = v ? (uint32_t)(((b * p.Hp + pr) * p.Wp + pc) * 8) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(base + d_piece[it] * 1024), 16, off, 0, 0, 0);
        }
    };
    // F32IN: chunk c = tid + NT it (two NHWC4 pixels) of the patch <- 6 floats of the three image planes, prefetched into registers
    float pre[HP][6];
    typedef f32x2 __attribute__((aligned(4))) f32x2_u;       // a pixel pair starts at an odd x: 4-byte aligned 8-byte loads
    auto load_patch = [&](int t) {
        const bool tv = t < t_end;
        const int tt = tv ? t : 0;
        const int b = tt / tiles_img, trem = tt - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int y0 = 4 * ty * kTPY - 5, x0 = 4 * tx * kTPX - 5;          // image coordinates of the patch origin (packed - 3)
        const float* img = b < p.B0 ? p.img0 + (size_t)b * 3 * p.H * p.W : p.img1 + (size_t)(b - p.B0) * 3 * p.H * p.W;
        const size_t plane = (size_t)p.H * p.W;
#pragma unroll
        for (int it = 0; it < HP; ++it) {
            // UNCONDITIONAL loads (a border pixel pair is fetched from the clamped position and shifted / zeroed afterwards): no
            // exec-masked branches in front of the MFMA phase, three 8-byte loads per chunk, lanes = consecutive pixel pairs
            const int c = tid + NT * it;
            const int cc = c < kChunks ? c : kChunks - 1;
            const int r = cc / kCPR, j = cc - r * kCPR;
            const int y = y0 + r, x = x0 + 2 * j;
            const int yc = y < 0 ? 0 : (y >= p.H ? p.H - 1 : y);
            const int xc = x < 0 ? 0 : (x > p.W - 2 ? p.W - 2 : x);
            const int sh = x - xc;                                          // 0: (lo, hi); -1: (0, lo); +1: (hi, 0); else (0, 0)
            const bool rowv = tv && c < kChunks && y == yc;
            const float* src = img + (size_t)yc * p.W + xc;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const f32x2 v = *(const f32x2_u*)(src + ch * plane);
                const float a0 = sh == 0 ? v[0] : (sh == 1 ? v[1] : 0.f);
                const float a1 = sh == 0 ? v[1] : (sh == -1 ? v[0] : 0.f);
                pre[it][2 * ch] = rowv ? a0 : 0.f;
                pre[it][2 * ch + 1] = rowv ? a1 : 0.f;
            }
        }
    };
    auto store_patch = [&](int stage) {
        char* base = smem + stage * kInStage;
#pragma unroll
        for (int it = 0; it < HP; ++it) {
            const int c = tid + NT * it;
            if (c < kPieces * 64) {       // the tail of the last 1 KiB piece is written too (zeros), like the DMA did
                i32x4 o;
                o[0] = Fmt16<short>::pack2(pre[it][0], pre[it][2]);
                o[1] = Fmt16<short>::pack2(pre[it][4], 0.f);
                o[2] = Fmt16<short>::pack2(pre[it][1], pre[it][3]);
                o[3] = Fmt16<short>::pack2(pre[it][5], 0.f);
                *(i32x4*)(base + c * 16) = o;
            }
        }
    };
    // conv pixel q of block blk for this lane; fragment byte = ((2cy + ky)*36 + cx + 2ks + half)*16
    constexpr int MAXB = (kNBLK + WMI - 1) / WMI;
    // (only the fragment base lives across tiles: the pixel's (cy, cx) is re-derived in the epilogue -- a constant division -- instead of
    // three more arrays held through the MFMA phase: the 4-wave variant must fit 168 registers for three waves per SIMD)
    int fbase[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int q = (wmi + WMI * i) * 32 + lr;
        const int qc = q < kNQ ? q : kNQ - 1;
        const int cy = qc / kCX, cx = qc - cy * kCX;
        fbase[i] = ((2 * cy) * kCPR + cx + half) * 16;
    }

    int t = t_first;
    if constexpr (F32IN) {
        load_patch(t);
        store_patch(0);
    } else {
        issue_patch(t, 0);
        issue_patch(t + nwg, 1);
    }
    int stage = 0;
    for (; t < t_end; t += nwg) {
        if constexpr (F32IN) {
            // patch(k) was written to its stage by every wave at the end of tile k-1 (or in the prologue); pool phase k-1 is done
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            load_patch(t + nwg);                                       // next tile's pixels travel under this tile's MFMAs
        } else {
        // patch(k) landed for every wave; pool phase of tile k-1 finished -> staging and input stage (k-1)%3 are free
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HP) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int st2 = stage >= 1 ? stage - 1 : kInStages - 1;      // (k+2) % 3 == (k-1) % 3
        issue_patch(t + 2 * nwg, st2);
        }

        const int b = t / tiles_img, trem = t - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int cy0 = 2 * ty * kTPY - 1, cx0 = 2 * tx * kTPX - 1;   // conv coordinates of conv pixel (0, 0) of the tile
        const char* in_s = smem + stage * kInStage;
        // ---- conv phase --------------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            if (wmi + WMI * i < kNBLK) {               // wave-uniform
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int f = 0; f < 14; ++f) {
                    const i32x4 frag = *(const i32x4*)(in_s + fbase[i] + ((f >> 1) * kCPR + 2 * (f & 1)) * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[f]), __builtin_bit_cast(bf16x8, frag), acc, 0, 0, 0);
                }
                const int q = (wmi + WMI * i) * 32 + lr;
                const int qcy = q / kCX, qcx = q - qcy * kCX;
                const bool qv = q < kNQ && (unsigned)(cy0 + qcy) < (unsigned)p.Ho && (unsigned)(cx0 + qcx) < (unsigned)p.Wo;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = wn * 32 + 8 * g + 4 * half;
                    const f32x4 sc = *(const f32x4*)(ss + nb);
                    const f32x4 sh = *(const f32x4*)(ss + 64 + nb);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[4 * g + e] * sc[e];
                        v[e] += sh[e];
                        v[e] = qv ? fmaxf(v[e], 0.f) : 0.f;
                    }
                    i32x2 o;
                    o[0] = (int)((uint32_t)(uint16_t)f2bf(v[0]) | ((uint32_t)(uint16_t)f2bf(v[1]) << 16));
                    o[1] = (int)((uint32_t)(uint16_t)f2bf(v[2]) | ((uint32_t)(uint16_t)f2bf(v[3]) << 16));
                    if (q < kNQ) *(i32x2*)(stg + q * 128 + (((wn * 4 + g) ^ (q & 7)) << 4) + half * 8) = o;     // (staging holds exactly kNQ rows)
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- pool phase: item = (pooled pixel, 8-channel slot); TPY * 16 * 8 items / NT threads -----------------------
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        static_assert((kTPY * kTPX * 8) % NT == 0, "pool items per thread");
#pragma unroll
        for (int pass = 0; pass < kTPY * kTPX * 8 / NT; ++pass) {
            const int item = pass * NT + tid;
            const int pp = item >> 3, s = item & 7;
            const int ly = pp >> 4, lx = pp & 15;
            s16x8 m;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int q = (2 * ly + dy) * kCX + 2 * lx + dx;
                    const s16x8 v = *(const s16x8*)(stg + q * 128 + ((s ^ (q & 7)) << 4));
                    m = (dy == 0 && dx == 0) ? v : __builtin_elementwise_max(m, v);
                }
            const int64_t pix = ((int64_t)b * p.Hq + ty * kTPY + ly) * p.Wq + tx * kTPX + lx;
            *(s16x8*)(p.out + (pix * p.out_pix_stride + s * 8) * 2) = m;
        }
        stage = stage + 1 == kInStages ? 0 : stage + 1;
        if constexpr (F32IN) store_patch(stage);     // the (single) stage was last read by this tile's conv phase: every wave is past the barrier behind it
    }
}

}  // namespace

template <bool F32IN, int TPY, int NW>
static int launch_stem_t(StemArgs& a, int wg_per_cu, hipStream_t stream) {
    constexpr int LDS = stem_lds_bytes<TPY, F32IN>();
    a.ntiles = a.B * (a.Hq / TPY) * (a.Wq / kTPX);
    a.plain_walk = vd3d_switch(VD3D_SW_PLAIN_TILE_WALK) ? 1 : 0;      // A/B of the XCD-aware tile walk (same tiles, same arithmetic: bit-identical)
    const int num_cu = vd3d_device_cu_count();
    if (num_cu <= 0) return VD3D_ELAUNCH;
    const int slots = num_cu * wg_per_cu;
    const int grid = a.ntiles < slots ? a.ntiles : slots;
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)stem_pool_kernel<F32IN, TPY, NW>, LDS, lim, "hipFuncSetAttribute(stem_pool)")) return rc;
    hipLaunchKernelGGL((stem_pool_kernel<F32IN, TPY, NW>), dim3(grid), dim3(NW * 64), LDS, stream, a);
    return vd3d_check_launch("stem_conv_pool");
}

static int launch_stem(StemArgs& a, int B, int H, int W, int Kpad, int out_pix_stride, bool f32in, hipStream_t stream) {
    a.B = B; a.H = H; a.W = W; a.Hp = H + 6; a.Wp = W + 8; a.Ho = H / 2; a.Wo = W / 2; a.Hq = H / 4; a.Wq = W / 4; a.Kpad = Kpad;
    a.out_pix_stride = out_pix_stride;
    if (!f32in) return launch_stem_t<false, 8, 8>(a, 1, stream);
    // VD3D_STEM_WG4=1 (round-4 experiment, same arithmetic, bit-identical): 4 x 16 pooled tiles, four waves, TWO independent workgroups per CU
    // (180 VGPRs; a third needs <= 168 and spills 13).  MEASURED SLOWER: 105.7 - 121 us against 99.8 - 109 us for the 8 x 16 / eight-wave /
    // one-per-CU kernel (16 x 3 x 384 x 1280, same box) -- the stem is not held back by its waves running in lock step: per 8 x 16 tile it
    // issues ~4.0k cycles of MFMA per SIMD (K padded 21 -> 32 per kernel row), ~2.9k of epilogue VALU (BN, ReLU, round on 561 x 64 values)
    // and ~3.7k of LDS traffic (504 KB of fragment reads, 72 KB parked, 147 KB of pool reads), and the smaller tile adds 6 % MFMA work and
    // 18 % patch bytes for the same eight waves per CU.
    if (vd3d_switch(VD3D_SW_STEM_WG4)) return launch_stem_t<true, 4, 4>(a, 2, stream);
    return launch_stem_t<true, 8, 8>(a, 1, stream);
}

static bool stem_shape_ok(int B, int H, int W, int Kpad, int out_pix_stride) {
    return B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0 && (H / 4) % 8 == 0 && (W / 4) % kTPX == 0 && Kpad >= 224 && out_pix_stride >= 64 &&
           out_pix_stride % 8 == 0;
}

extern "C" int vd3d_stem_conv_pool(const void* packed, const void* weight, const float* scale, const float* shift, void* out,
                                   int B, int H, int W, int Kpad, int out_pix_stride, void* stream) {
    if (!packed || !weight || !out) { vd3d_set_error("stem_conv_pool: null pointer"); return VD3D_EINVAL; }
    if (!stem_shape_ok(B, H, W, Kpad, out_pix_stride) || ((uintptr_t)packed & 15) || ((uintptr_t)weight & 15) || ((uintptr_t)out & 15)) {
        vd3d_set_error("stem_conv_pool: needs H/4 % 8 == 0, W/4 % 16 == 0, Kpad >= 224, 16-byte aligned pointers / rows");
        return VD3D_EINVAL;
    }
    StemArgs a = {};
    a.packed = (const char*)packed; a.weight = (const char*)weight; a.scale = scale; a.shift = shift; a.out = (char*)out;
    const int64_t in_bytes = (int64_t)B * (H + 6) * (W + 8) * 8;
    if (in_bytes > 0x7ffffff0ll) { vd3d_set_error("stem_conv_pool: packed image exceeds 2 GiB; split the batch"); return VD3D_ERANGE; }
    a.in_bytes = (uint32_t)in_bytes;
    return launch_stem(a, B, H, W, Kpad, out_pix_stride, false, (hipStream_t)stream);
}

extern "C" int vd3d_stem_conv_pool_f32(const float* img0, int B0, const float* img1, int B1, const void* weight, const float* scale,
                                       const float* shift, void* out, int H, int W, int Kpad, int out_pix_stride, void* stream) {
    if (!img0 || (B1 > 0 && !img1) || !weight || !out || B0 <= 0 || B1 < 0) { vd3d_set_error("stem_conv_pool_f32: null pointer / empty batch"); return VD3D_EINVAL; }
    if (!stem_shape_ok(B0 + B1, H, W, Kpad, out_pix_stride) || ((uintptr_t)weight & 15) || ((uintptr_t)out & 15)) {
        vd3d_set_error("stem_conv_pool_f32: needs H/4 % 8 == 0, W/4 % 16 == 0, Kpad >= 224, 16-byte aligned weight / output");
        return VD3D_EINVAL;
    }
    StemArgs a = {};
    a.img0 = img0; a.img1 = img1; a.B0 = B0;
    a.weight = (const char*)weight; a.scale = scale; a.shift = shift; a.out = (char*)out;
    return launch_stem(a, B0 + B1, H, W, Kpad, out_pix_stride, true, (hipStream_t)stream);
}
