// conv_resident.hip -- implicit-GEMM convolutions whose WEIGHTS stay resident in registers for the whole launch (bf16, 3x3 /
// stride 1 / pad 1): persistent workgroups, LDS holds input halos only.  Same math, operand layouts and fused epilogue as the tile
// kernels of conv_igemm.hip (which dispatches here); replaces the same reference layers (backbones/resnet.py:23-52 BasicBlock
// convs of layer1 / layer2, heads/detection_3d_head.py:54-68 cls tower, backbones/dla.py tree blocks).
#include "conv_common.h"

#include <cstdlib>
#include <stdio.h>

using namespace vd3d_conv;

namespace {

// =====================================================================================================
// v5 "resident weights" kernel: 3x3 / stride 1 / pad 1, 64 -> 64 channels, bf16 (ResNet layer1: six launches per forward,
// K = 576 only).  The tile kernels above spend such a layer waiting: nine tiny K slices per tile, each a DMA round trip
// and a barrier for 8 MFMAs per wave, and the 73 KB weight panel is re-fetched by every tile.  Here the whole weight
// panel lives in REGISTERS (a wave owns 32 output channels: 9 taps x 4 sub-steps x 16 B = 144 VGPRs), the workgroups are
// persistent (one per CU, tiles t = wg, wg + nwg, ...), and LDS only holds a 4-deep ring of input halos (8 x 16 pixels
// + border, one 128-byte row per pixel) filled by LDS-DMA two to three tiles ahead.  Per tile: ONE barrier, 36 MFMAs per
// wave each fed by one ds_read_b128 through a 4-fragment ring, epilogue fused as elsewhere.  The layer is then HBM-bound
// (in x1.4 halo + out + residual ~= 215 MB per launch at B = 16 x 96 x 320), not latency-bound.
constexpr int kResTH = 8, kResTW = 16, kResStages = 4;
constexpr int kResHW2 = kResTW + 2, kResHR = (kResTH + 2) * kResHW2;       // 18, 180 halo pixels
constexpr int kResPiecesTot = (kResHR + 7) / 8;                            // 23 pieces of 1 KiB
constexpr int kResStageBytes = kResPiecesTot * 1024;
constexpr int kResLds = kResStages * kResStageBytes + 512;   // + scale[64], shift[64] (fp32): epilogue reads them through
                                                             // lgkmcnt, a global load per tile would drain the DMA queue (vmcnt)

// XCD-aware walk of a row-major tile list by a persistent grid (round 5): workgroups b, b + 8, ... share an XCD and its L2.  Each XCD owns a CONTIGUOUS run of
// the list and its workgroups take consecutive tiles of it in every round, so the halo rows / columns neighbouring tiles share are L2 hits -- with the plain
// walk t = b, b + grid, ... neighbours sit on different XCDs and every overlap is fetched from beyond L2 twice.  -> first tile, end of the run, stride.
// A grid that is not a multiple of 8 keeps the plain walk.  Same tiles, same arithmetic: bit-identical (VD3D_PLAIN_TILE_WALK=1 for A/B).
struct TileWalk { int first, end, stride; };
VD3D_DEV TileWalk xcd_tile_walk(int ntiles, int plain_walk) {
    TileWalk w = {(int)blockIdx.x, ntiles, (int)gridDim.x};
    if ((gridDim.x & 7) == 0 && !plain_walk) {
        const int xcd = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        const int start = xcd * q + (xcd < r ? xcd : r);
        w.first = start + (int)(blockIdx.x >> 3);
        w.end = start + q + (xcd < r ? 1 : 0);
        w.stride = (int)(gridDim.x >> 3);
    }
    return w;
}
inline int plain_tile_walk() { return vd3d_switch(VD3D_SW_PLAIN_TILE_WALK) ? 1 : 0; }

template <typename T, bool RES>
__global__ void __launch_bounds__(512) conv_resident64_kernel(const ConvArgs p, int ntiles, int plain_walk) {
    constexpr int NW = 8, HP = 3;                     // waves; halo pieces per wave (23 = 7 waves x 3 + 1 wave x 2)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;          // 4 x 2 waves: 32 pixels x 32 channels each
    const int lr = lane & 31, half = lane >> 5;
    const int tiles_x = (p.W + kResTW - 1) / kResTW, tiles_y = (p.H + kResTH - 1) / kResTH, tiles_img = tiles_x * tiles_y;   // ragged
    // edges: halo rows beyond the image are zero-filled by the bounds check, pixels beyond it are computed and dropped

    // ---- weights -> registers, MFMA A-operand layout: row = output channel, 16 bytes = 8 k values -----------
    i32x4 wf[36];
    {
        if (p.wfrag) {      // register image: 36 coalesced 1 KiB reads per wave
            const char* wimg = p.wfrag + ((size_t)wn * 36 * 64 + lane) * 16;
#pragma unroll
            for (int f = 0; f < 36; ++f) wf[f] = *(const i32x4*)(wimg + f * 1024);
        } else {
            const char* wrow = p.weight + (size_t)(wn * 32 + lr) * p.Kpad * 2;
#pragma unroll
            for (int f = 0; f < 36; ++f) wf[f] = *(const i32x4*)(wrow + ((f >> 2) * 64 + (2 * (f & 3) + half) * 8) * 2);
        }
    }
    // ---- halo DMA lane state (same lane-linear image + source-side swizzle as the halo kernel) -------------------
    const int prow = lane >> 3;
    // swizzle key of a halo row = (pixel index hy*16 + hx) / 2 mod 8: the 32 rows a wave reads for any tap shift then carry
    // consecutive keys, so every ds_read_b128 lane group covers all 16 bank slots (the plain (row/2)%8 key conflicts 2-way
    // across the 18-row pitch: PMC showed 47 % of this kernel's LDS cycles as bank conflicts)
    auto hkey = [](int row) { const int hy = row / kResHW2; return ((hy * kResTW + row - hy * kResHW2) >> 1) & 7; };
    int h_slot[HP];
    int h_y[HP], h_x[HP];
    bool h_ok[HP];
    int h_piece[HP];
#pragma unroll
    for (int it = 0; it < HP; ++it) {
        int piece = wave + it * NW;
        if (piece >= kResPiecesTot) piece -= NW;      // branch-free partial round: repeat the previous piece
        h_piece[it] = piece;
        const int hr = 8 * piece + prow;
        h_y[it] = hr / kResHW2;
        h_x[it] = hr - h_y[it] * kResHW2;
        h_ok[it] = hr < kResHR;
        h_slot[it] = (lane & 7) ^ hkey(hr);
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    // XCD-aware tile walk (xcd_tile_walk above): 10 x 18 pixels are staged per 8 x 16 computed (1.41 x) -- with the plain walk FETCH_SIZE read 1.38 x the
    // input of a kernel that runs at the bandwidth of a mixed read / write stream
    const TileWalk walk = xcd_tile_walk(ntiles, plain_walk);
    int t = walk.first;
    const int t_end = walk.end, nwg = walk.stride;
    auto issue_halo = [&](int t, int stage) {
        const bool tv = t < t_end;
        const int tt = tv ? t : 0;
        const int b = tt / tiles_img, trem = tt - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        char* base = smem + stage * kResStageBytes;
#pragma unroll
        for (int it = 0; it < HP; ++it) {
            const int iy = ty * kResTH - 1 + h_y[it], ix = tx * kResTW - 1 + h_x[it];
            const bool v = tv && h_ok[it] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t off = v ? (uint32_t)((int)(b * p.in_batch_stride) + iy * p.in_row_stride + ix * p.in_pix_stride + h_slot[it] * 8) * 2 : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(base + h_piece[it] * 1024), 16, off, 0, 0, 0);
        }
    };
    // ---- fragment addressing: pixel pp of the tile, tap (dy, dx) -> halo row; byte = row*128 + ((2ks+half) ^ sw(row))*16
    //      = A_tap ^ (ks << 5) with A_tap = row*128 + ((half ^ sw(row)) << 4), sw = hkey
    const int pp = wm * 32 + lr;
    const int hrow0 = (pp / kResTW) * kResHW2 + (pp & (kResTW - 1));
    int a_tap[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int row = hrow0 + (tap / 3) * kResHW2 + (tap % 3);
        a_tap[tap] = row * 128 + ((half ^ hkey(row)) << 4);
    }
    auto ld_frag = [&](int stage, int f) {
        int base = a_tap[f >> 2];
        asm volatile("" : "+v"(base));            // keeps the 36 (tap, ks) addresses from being hoisted out of the tile loop (they spilled)
        return *(const i32x4*)(smem + stage * kResStageBytes + (base ^ ((f & 3) << 5)));
    };

    float* ss = (float*)(smem + kResStages * kResStageBytes);
    if (tid < 64) {
        ss[tid] = p.scale ? p.scale[tid] : 1.f;
        ss[64 + tid] = p.shift ? p.shift[tid] : 0.f;
    }
#pragma unroll
    for (int s0 = 0; s0 < kResStages; ++s0) issue_halo(t + s0 * nwg, s0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kResStages - 1) * HP) : "memory");   // the first halo has landed (per wave) ...
    __builtin_amdgcn_s_barrier();                                        // ... for every wave
    asm volatile("" ::: "memory");
    i32x4 ring[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) ring[f] = ld_frag(0, f);

    // Per tile k a wave issues, in order: [R(k): 4 residual loads, at the top] D(k+4): HP DMA pieces (after the barrier),
    // S(k): 2 stores.  At tile k's barrier "at most 2*HP outstanding" leaves only R(k) / S(k-1) (and, without a residual,
    // D(k+3)) in flight: D(k+2) and everything older -- in particular this barrier's D(k+1) -- has landed, and each halo gets
    // at least one full tile time to arrive.  Also right for k = 0 (prologue D(0..3): D(0), D(1) landed).  gfx9 vmcnt
    // retires loads and stores in issue order.
    constexpr int kYounger = 2 * HP + (RES ? 4 : 0);     // RES: the 4 residual loads of this tile are younger too
    int stage = 0;
    for (; t < t_end; t += nwg) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int b = t / tiles_img, trem = t - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int y = ty * kResTH + pp / kResTW, x = tx * kResTW + (pp & (kResTW - 1));
        const bool pin = y < p.H && x < p.W;
        const int64_t m = pin ? ((int64_t)b * p.H + y) * p.W + x : 0;
        const int nb0 = wn * 32 + 4 * half;
        i32x2 rr[4];
        if constexpr (RES) {   // issued a whole tile ahead of their use in the epilogue (HBM latency hidden by the 36 MFMAs)
#pragma unroll
            for (int g = 0; g < 4; ++g) rr[g] = *(const i32x2*)(p.residual + (m * p.res_pix_stride + nb0 + 8 * g) * 2);   // (pixel 0 if outside)
        }
        const int nstage = stage + 1 == kResStages ? 0 : stage + 1;
#pragma unroll
        for (int f = 0; f < 36; ++f) {
            if (f == 32) {
                // all reads of this tile's halo are issued (and, with lgkmcnt(0), done); the next halo must have landed
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kYounger) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue_halo(t + kResStages * nwg, stage);                 // tile k+4 into the stage just released
            }
            Fmt16<T>::mfma32(wf[f], ring[f & 3], acc);
            ring[f & 3] = f + 4 < 36 ? ld_frag(stage, f + 4) : ld_frag(nstage, f + 4 - 36);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        // ---- epilogue: scale/shift (+residual) (+ReLU), 2 x 16-byte NHWC stores per lane (half-wave pairing) ----
        const int relu_floor = p.relu ? 0 : (int)0x80008000u;
        int pk[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nb = nb0 + 8 * g;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[4 * g + e];
            if (p.scale) {
                const f32x4 sc = *(const f32x4*)(ss + nb);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= sc[e];
            }
            if (p.shift) {
                const f32x4 sh = *(const f32x4*)(ss + 64 + nb);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += sh[e];
            }
            if constexpr (RES) {
                const uint32_t r0 = (uint32_t)(int)rr[g][0], r1 = (uint32_t)(int)rr[g][1];
                v[0] += Fmt16<T>::lo(r0);
                v[1] += Fmt16<T>::hi(r0);
                v[2] += Fmt16<T>::lo(r1);
                v[3] += Fmt16<T>::hi(r1);
            }
            // round, then ReLU on the packed pairs (bit-identical to fmaxf before the rounding; 4 instead of ~12 instructions per four values)
            pk[g][0] = max_pk16(Fmt16<T>::pack2_1(v[0], v[1]), relu_floor);
            pk[g][1] = max_pk16(Fmt16<T>::pack2_1(v[2], v[3]), relu_floor);
        }
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            auto r0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
            i32x4 o = {(int)r0[0], (int)r1[0], (int)r0[1], (int)r1[1]};
            if (pin) *(i32x4*)(p.out + (m * p.out_pix_stride + wn * 32 + 8 * (g + half)) * 2) = o;
        }
        stage = nstage;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead DMAs must not land in a successor workgroup's LDS
}

template <int V> struct IntC { static constexpr int value = V; };
// compile-time loop: the body is instantiated once per index (a `#pragma unroll` loop this large may be left rolled, and a rolled
// loop would index the weight registers dynamically, i.e. put them in scratch memory)
template <int... Is, class F> VD3D_DEV void static_for_impl(F&& f, IntC<0>, std::integer_sequence<int, Is...>) { (f(IntC<Is>{}), ...); }
template <int N, class F> VD3D_DEV void static_for(F&& f) { static_for_impl(f, IntC<0>{}, std::make_integer_sequence<int, N>{}); }

// (Round 5, built, measured, removed -- source and numbers: profiles/r05_resident64_two_workgroups_experiment.txt: this kernel as FOUR-wave
// workgroups, two per CU, so that one workgroup's epilogue runs under the other's MFMAs.  Bit-identical; 589 / 633 against 626 / 629 TF/s on the
// layer-1 shape, headline 3.799 / 3.833 against 3.799 / 3.833 ms: NOTHING.  The eight-wave kernel is not held back by its waves' lock step: at
// 3.0 - 4.5 TB/s of halo + output + residual traffic it sits at the bandwidth a mixed read / write stream gets on this part.)

// =====================================================================================================
// v6 "register-resident weights" kernel for the mid-size 3x3 / stride 1 / pad 1 layers with 128 or 256 input channels (ResNet
// layer2 / layer3, the 256 -> 256 cls conv, DLA levels): bf16.  The tile kernels above stream the weight panel through LDS
// once per pixel tile; with only 128 - 256 pixels per panel pass these layers are bound by that L2 -> LDS fill (layer3:
// 1.2 MB of weights per 128-pixel tile, ~15 B/clk/CU of LDS-DMA is all a CU gets) and by one barrier per 64-deep K slice.
// Here the weights never touch LDS:
//   * a workgroup is 4 waves, ONE per SIMD, 512 registers each.  A wave owns 32 output channels x 128 input channels
//     (two 64-channel chunks x 9 taps x 4 sub-steps = 72 MFMA A-fragments = 288 registers), loaded once per launch;
//     CIN = 128: 4 channel groups = 128 output channels per workgroup; CIN = 256: 2 channel groups x 2 K halves
//     (input channels 0-127 / 128-255) = 64 output channels per workgroup, the K halves are added through LDS per tile;
//   * workgroups are persistent (one per CU); the grid is (channel slice, pixel-tile lane): a CU keeps its slice's
//     weights and walks 8 x 16-pixel tiles.  The slices of one pixel tile sit on the same XCD (shared L2 for the input);
//   * LDS only holds input halos: one 23 KiB image (180 halo pixels x 64 channels, the resident64 layout + swizzle) per
//     64-channel chunk, filled by LDS-DMA one to three phases ahead;  a phase = one chunk per K half = 36 fragments x 4
//     pixel blocks = 144 MFMAs per wave between barriers, each MFMA fed by ONE ds_read_b128 through a fragment ring;
//   * every VMEM instruction is issued unconditionally (out-of-image lanes use an out-of-range buffer offset), so the
//     counted s_waitcnt vmcnt(N) below always sees the same queue.
#ifdef VD3D_TUNING
__device__ unsigned long long g_regw_dbg[64];
#endif
constexpr int kRwImg = kResStageBytes;            // 23 KiB: one 64-channel halo image of an 8 x 16 tile

// RWRING = pixel-fragment ring depth (ds_read_b128 issued that many MFMAs ahead); ABL != 0: timing ablations of the tuning
// build (WRONG results): 1 = no fragment reads, 2 = no halo DMA, 3 = neither
template <typename T, int CIN, bool RES, int RWRING = 4, int ABL = 0>
__global__ void __launch_bounds__(256) conv_regw_kernel(const ConvArgs p, int ntiles, int nslices) {
    constexpr int kRwRing = RWRING;
#ifdef VD3D_TUNING
    int dbg_n = 0;
#define RW_STAMP() do { if (ABL == 9 && blockIdx.x == 0 && threadIdx.x == 0 && dbg_n < 64) g_regw_dbg[dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define RW_STAMP() do { } while (0)
#endif
    RW_STAMP();
    static_assert(CIN == 128 || CIN == 256, "register-resident weights: 128 or 256 input channels");
    constexpr int KG = CIN / 128;                 // K halves (waves that share an output block)
    constexpr int CG = 4 / KG;                    // channel groups of 32 per workgroup
    constexpr int CSL = CG * 32;                  // output channels per workgroup (slice)
    constexpr int NSLOT = KG == 1 ? 4 : 2;        // LDS slots of one phase unit (KG images) each
    constexpr int UNIT = KG * kRwImg;
    constexpr int PIECES = KG * kResPiecesTot;    // 1 KiB DMA pieces per unit
    constexpr int P = (PIECES + 3) / 4;           // per wave (overshoot repeats the wave's previous piece)
    constexpr int RED_OFF = NSLOT * UNIT;         // K-half reduction buffer: CG x 16 KiB (KG == 2)
    constexpr int RB_OFF = RED_OFF + (KG == 2 ? CG * 16384 : 0);   // residual tiles: CG x 8 KiB ([128 px][32 ch] per wave)
    constexpr int SS_OFF = RB_OFF + (RES ? CG * 8192 : 0);
    constexpr int NR = RES ? 8 : 0, NS = 8;       // residual DMA pieces / output stores per (epilogue) wave and tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % CG, kg = wave / CG;
    const int lr = lane & 31, half = lane >> 5;
    const int tiles_x = (p.W + kResTW - 1) / kResTW, tiles_y = (p.H + kResTH - 1) / kResTH, tiles_img = tiles_x * tiles_y;
    // blockIdx -> (slice, tile lane): blocks b, b + 8, b + 16, ... share an XCD; consecutive ones take the slices of one tile lane
    const int bx = blockIdx.x, xcd = bx & 7, j8 = bx >> 3;
    const int slice = j8 % nslices;
    const int lanes_per_xcd = (int)(gridDim.x >> 3) / nslices;
    const int tlane = xcd * lanes_per_xcd + j8 / nslices, tstride = 8 * lanes_per_xcd;
    const int n_base = slice * CSL + cg * 32;      // first output channel of this wave

    // ---- weights -> registers (MFMA A operand: row = output channel, 16 bytes = 8 k values; k = tap * CIN + c) ----------
    i32x4 wf[72];
    {
        // register image [Cout/32][CIN/64][36][64][16 B]: 72 coalesced 1 KiB reads (chunks 2 * kg, 2 * kg + 1 of this wave's block)
        const char* wimg = p.wfrag + ((((size_t)(n_base >> 5) * (CIN / 64) + 2 * kg) * 36) * 64 + lane) * 16;
        static_for<72>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            wf[i] = *(const i32x4*)(wimg + i * 1024);
        });
    }
    auto hkey = [](int row) { const int hy = row / kResHW2; return ((hy * kResTW + row - hy * kResHW2) >> 1) & 7; };
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? p.residual : p.out), 0, 0x80000000u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto tile_origin = [&](int t, int& b, int& ty, int& tx) {
        b = t / tiles_img;
        const int trem = t - b * tiles_img;
        ty = trem / tiles_x;
        tx = trem - ty * tiles_x;
    };
    // unit u = 2 * k + ph of this workgroup's k-th tile: the images of chunk 2 * g + ph for every K half g, P pieces per wave.
    // The pieces of unit u + NSLOT - 1 are issued ONE AT A TIME inside phase u's MFMA stream (a piece costs 60-180 issue
    // cycles plus ~30 VALU of address arithmetic; in one burst they stall the only wave of the SIMD), into the slot the
    // barrier at the end of phase u - 1 released.  Everything but the lane id is recomputed per piece: keeping per-piece
    // lane state alive would cost 2 registers per piece.
    struct UnitOrigin { int b, ty, tx, ph, slot; bool tv; };
    auto unit_origin = [&](int u) {
        UnitOrigin o;
        const int k = u >> 1;
        const int t = tlane + k * tstride;
        o.ph = u & 1;
        o.tv = t < ntiles;
        o.slot = u % NSLOT;
        tile_origin(o.tv ? t : 0, o.b, o.ty, o.tx);
        return o;
    };
    auto issue_piece = [&](const UnitOrigin& o, int it) {
        int ln = lane;
        asm volatile("" : "+v"(ln));               // keeps the per-piece address arithmetic out of the loop-invariant set
        int q = wave + it * 4;
        if (q >= PIECES) q -= 4;                   // branch-free partial round: repeat the previous piece (same data)
        const int g = q / kResPiecesTot, pi = q - g * kResPiecesTot;
        const int hr = 8 * pi + (ln >> 3);
        const int hy = hr / kResHW2, hx = hr - hy * kResHW2;
        const int iy = o.ty * kResTH - 1 + hy, ix = o.tx * kResTW - 1 + hx;
        const bool v = o.tv && hr < kResHR && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int col = (2 * g + o.ph) * 64 + ((ln & 7) ^ hkey(hr)) * 8;
        // always computed; an out-of-image lane gets the top bit = out of range (a select the compiler cannot turn into an
        // exec-masked branch inside the MFMA stream; in_bytes < 2^31)
        const uint32_t off = ((uint32_t)(o.b * (int)p.in_batch_stride + iy * p.in_row_stride + ix * p.in_pix_stride + col) * 2u) | (v ? 0u : kOOB);
        if constexpr (ABL != 2 && ABL != 3)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(smem + o.slot * UNIT + g * kRwImg + pi * 1024), 16, off, 0, 0, 0);
    };
    auto issue_unit = [&](int u) {
        const UnitOrigin o = unit_origin(u);
#pragma unroll
        for (int it = 0; it < P; ++it) issue_piece(o, it);
    };
    // Residual of a wave's own 128 px x 32 ch output block -> its private 8 KiB LDS tile [128 px][64 B], by DMA (no registers
    // held across the MFMA phases).  16-byte slot s of pixel px sits at physical slot s ^ (px & 3) ^ ((px >> 2) & 3): the
    // epilogue's ds_read_b64 (32 pixels x one 8-byte half slot) then conflicts 2-way at most.
    char* rb = smem + RB_OFF + cg * 8192;
    auto issue_residual = [&](int b, int ty, int tx) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {
            const int px = 16 * pc + (ln >> 2);
            const int y = ty * kResTH + px / kResTW, x = tx * kResTW + (px & (kResTW - 1));
            const int sl = (ln & 3) ^ (px & 3) ^ ((px >> 2) & 3);
            const bool v = y < p.H && x < p.W;
            const uint32_t off = v ? (uint32_t)(((((int64_t)b * p.H + y) * p.W + x) * p.res_pix_stride + n_base + sl * 8) * 2) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(res_rsrc, (lds_ptr_t)(rb + pc * 1024), 16, off, 0, 0, 0);
        }
    };
    // ---- fragment addressing (as resident64): byte = a_tap[tap] ^ (ks << 5), + 4608 per 32-pixel block, + image base ----
    int a_tap[9];
    {
        const int hrow0 = (lr / kResTW) * kResHW2 + (lr & (kResTW - 1));
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int row = hrow0 + (tap / 3) * kResHW2 + (tap % 3);
            a_tap[tap] = row * 128 + ((half ^ hkey(row)) << 4) + kg * kRwImg;
        }
    }
    auto ld_frag = [&](int u, int m) {            // m = f * 4 + j within the unit
        const int f = m >> 2, j = m & 3;
        return *(const i32x4*)(smem + (u % NSLOT) * UNIT + j * 4608 + (a_tap[f >> 2] ^ ((f & 3) << 5)));
    };
    float* ss = (float*)(smem + SS_OFF);
    if (tid < CSL) {
        ss[tid] = p.scale ? p.scale[slice * CSL + tid] : 1.f;
        ss[CSL + tid] = p.shift ? p.shift[slice * CSL + tid] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NSLOT - 1; ++u) issue_unit(u);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSLOT - 2) * P) : "memory");   // unit 0 has landed (this wave's pieces) ...
    __builtin_amdgcn_s_barrier();                                               // ... for every wave (and scale / shift are in LDS)
    asm volatile("" ::: "memory");
    i32x4 ring[kRwRing];
#pragma unroll
    for (int m = 0; m < kRwRing; ++m) ring[m] = ld_frag(0, m);
    RW_STAMP();

    const bool epi = kg == 0;                     // the waves that own the epilogue (KG == 2: the other half only contributes)
    // VMEM queue of an epilogue wave, tile k:  R(k) [NR residual pieces, top]  D(2k + NSLOT - 1) [P, spread over phase 0]
    // D(2k + NSLOT) [P, spread over phase 1]  S(k) [NS stores].  The other K half issues the D's only.
    constexpr int STEP = KG == 1 ? 144 / P : 72 / P;      // MFMAs between two pieces (CIN 256: all in the first half of the phase,
    static_assert(STEP >= 2, "piece spacing");             // its unit is needed at the end of the SAME phase)
    int k = 0;
    for (int t = tlane; t < ntiles; t += tstride, ++k) {
        f32x16 acc[4];             // never zeroed: the first MFMA of a block takes C = 0
        int b, ty, tx;
        tile_origin(t, b, ty, tx);
        if constexpr (RES) {
            if (epi) issue_residual(b, ty, tx);
        }
        // one phase = one 64-channel chunk per K half; `ph` must be a compile-time constant (it selects weight REGISTERS), so the
        // two phases are two instantiations of this lambda rather than a loop the unroller may decline to unroll
        auto phase = [&](auto ph_c) {
            constexpr int ph = decltype(ph_c)::value;
            const int u = 2 * k + ph;
            const UnitOrigin nxt = unit_origin(u + NSLOT - 1);
            static_for<144>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                if constexpr (m % STEP == 1 && m / STEP < P) issue_piece(nxt, m / STEP);
                if constexpr (m == 144 - kRwRing) {
                    // every read of this unit has been issued (lgkmcnt(0): and is done); the next unit, D(u + 1), must have
                    // landed: at most the VMEM ops issued AFTER it may still be in flight (steady state; the first two
                    // tiles simply drain the queue).
                    //   NSLOT 4:  D(u + 1) was issued during phase u - 2; younger: D(u + 2), D(u + 3), and one S / R pair
                    //   NSLOT 2:  D(u + 1) was issued in the first half of THIS phase; nothing is younger
                    if (k < 2 || NSLOT == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * P + NS + NR) : "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                constexpr int f = m >> 2, j = m & 3;
                if constexpr (ph == 0 && f == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    Fmt16<T>::mfma32z(wf[0], ring[m % kRwRing], zero, acc[j]);
                } else
                    Fmt16<T>::mfma32(wf[ph * 36 + f], ring[m % kRwRing], acc[j]);
                if constexpr (ABL != 1 && ABL != 3)
                    ring[m % kRwRing] = m + kRwRing < 144 ? ld_frag(u, m + kRwRing) : ld_frag(u + 1, m + kRwRing - 144);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            });
        };
        RW_STAMP();
        phase(IntC<0>{});
        RW_STAMP();
        phase(IntC<1>{});
        RW_STAMP();
        if constexpr (KG == 2) {
            // K halves: the upper half parks its partial sums in LDS (16 ds_write_b128, lane-linear), the lower half adds them
            float* red = (float*)(smem + RED_OFF + cg * 16384);
            if (!epi) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                        *(f32x4*)(red + ((j * 4 + g) * 64 + lane) * 4) = v;
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (epi) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = *(const f32x4*)(red + ((j * 4 + g) * 64 + lane) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j][4 * g + e] += v[e];
                    }
            }
        }
        if (epi) {
            // ---- epilogue: scale/shift (+residual) (+ReLU), 2 x 16-byte NHWC stores per lane and block (half-wave pairing) ----
            if constexpr (RES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");    // R(k) landed (own pieces only)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int px = 32 * j + lr;
                const int y = ty * kResTH + px / kResTW, x = tx * kResTW + (px & (kResTW - 1));
                const bool pin = y < p.H && x < p.W;
                const uint32_t o_off = pin ? (uint32_t)(((((int64_t)b * p.H + y) * p.W + x) * p.out_pix_stride + n_base) * 2) : kOOB;
                int pk[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = cg * 32 + 4 * half + 8 * g;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[j][4 * g + e];
                    const f32x4 sc = *(const f32x4*)(ss + nl), sh = *(const f32x4*)(ss + CSL + nl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
                    if constexpr (RES) {
                        const i32x2 rr = *(const i32x2*)(rb + px * 64 + ((g ^ (px & 3) ^ ((px >> 2) & 3)) << 4) + 8 * half);
                        const uint32_t r0 = (uint32_t)(int)rr[0], r1 = (uint32_t)(int)rr[1];
                        v[0] += Fmt16<T>::lo(r0);
                        v[1] += Fmt16<T>::hi(r0);
                        v[2] += Fmt16<T>::lo(r1);
                        v[3] += Fmt16<T>::hi(r1);
                    }
                    pk[g][0] = max_pk16(Fmt16<T>::pack2_1(v[0], v[1]), p.relu ? 0 : (int)0x80008000u);      // round, then ReLU on the packed pairs (bit-identical)
                    pk[g][1] = max_pk16(Fmt16<T>::pack2_1(v[2], v[3]), p.relu ? 0 : (int)0x80008000u);
                }
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    auto r0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                    i32x4 o = {(int)r0[0], (int)r1[0], (int)r0[1], (int)r1[1]};
                    const uint32_t off = pin ? o_off + 16 * (g + half) : kOOB;
                    __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, off, 0, 0);
                }
            }
            if constexpr (RES) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the residual tile is read before R(k + 1) overwrites it
        }
        RW_STAMP();
    }
    RW_STAMP();
#undef RW_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead DMAs must not land in a successor workgroup's LDS
}


// =====================================================================================================
// "small-channel streaming" kernel: 3x3 / pad 1 / stride 1 | 2, Cin = 16 | 32 | 64, Cout <= 32 (DLA level 0 / 1 at full and half
// resolution -- 16 -> 16 at 16 x 512 x 1760 is 0.9 GB of activations for 66 GF -- and the 64 -> 27 offset convs of the DCN blocks).
// These layers are HBM-bound by a wide margin, but the tile kernels run them at 1.1 - 1.5 TB/s: every input pixel goes through
// LDS-DMA nine times (once per tap) in 32-byte crumbs.  Here: persistent workgroups of 8 waves own 8 x 32 OUTPUT pixels (one
// 32-pixel row per wave), the (8S+2) x (32S+2) input halo is staged ONCE by LDS-DMA (ring of halo stages filled one to three
// tiles ahead), the weights (<= 36 fragments) live in registers, a tile costs 9 * Cin/16 MFMAs per wave (tap x 16-channel
// k-step; the MFMA's 32 output-channel rows are padded with zero filters) and one barrier.  Input read once (+ halo overlap),
// output written once.  Epilogue: scale / shift (+ReLU), 16-bit (Cout % 16 == 0) or fp32 (Cout % 4 == 0) 16-byte stores.
template <typename T, int CIN, int S, bool OUTF32>
__global__ void __launch_bounds__(512) conv_small_kernel(const ConvArgs p, int ntiles, int plain_walk) {
    constexpr int TH = 8, TW = 32;
    constexpr int HH = (TH - 1) * S + 3, HWD = (TW - 1) * S + 3;       // input halo of one output tile
    constexpr int PITCH = CIN * 2;                                    // bytes per halo pixel in LDS
    constexpr int SLOTS = PITCH / 16;                                 // 16-byte slots per pixel
    constexpr int PXPP = 1024 / PITCH;                                // halo pixels per 1 KiB DMA piece
    constexpr int HR = HH * HWD;
    constexpr int PIECES = (HR + PXPP - 1) / PXPP;
    constexpr int P = (PIECES + 7) / 8;                               // per wave (overshoot repeats the wave's previous piece)
    constexpr int STAGE = PIECES * 1024;
    constexpr int NST = (150 * 1024 / STAGE) >= 4 ? 4 : ((150 * 1024 / STAGE) >= 3 ? 3 : 2);
    constexpr int KS = CIN / 16, NF = 9 * KS;                          // MFMAs (= A fragments) per tile and wave
    constexpr int NS = OUTF32 ? 4 : 2;                                // stores per tile and wave
    // Fragment ring: fragment j of EVERY tile lives in slot j % RING, and the slot freed by fragment f is refilled with fragment f + RING
    // -- across the tile boundary that is next tile's fragment f + RING - NF, whose slot is (f + RING - NF) % RING: the two agree only
    // when RING divides NF.  (Round 2 shipped RING = 4 with NF = 9 | 18 (Cin 16 | 32): from a workgroup's SECOND tile on the MFMAs read
    // rotated fragments -- wrong results whenever a launch had more tiles than workgroup slots, i.e. at BASELINE config 5's size;
    // found by the fp16-at-size test of round 3.  Cin 64 (NF = 36) was unaffected.)
    constexpr int RING = NF % 4 == 0 ? 4 : 3;
    static_assert(NF % RING == 0, "the fragment ring must divide the fragments of a tile");
    static_assert(CIN == 16 || CIN == 32 || CIN == 64, "small-channel kernel: Cin 16 | 32 | 64");
    static_assert(NST * STAGE + 512 <= 160 * 1024, "halo ring does not fit the LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, half = lane >> 5;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH, tiles_img = tiles_x * tiles_y;

    // ---- weights -> registers (A operand: row = output channel (zero rows beyond Cout), 16 bytes = 8 k values) ------------
    i32x4 wf[NF];
    {
        const char* wrow = p.weight + (size_t)lr * p.Kpad * 2;
        static_for<NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value, tap = f / KS, ks = f % KS;
            wf[f] = *(const i32x4*)(wrow + (tap * CIN + ks * 16 + half * 8) * 2);
        });
    }
    // swizzle key of a halo row (128-byte rows only: the tile kernels' XOR on the 16-byte slot)
    auto key = [](int row) { return PITCH == 128 ? ((row >> 1) & 7) : 0; };
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x80000000u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int in_bs = (int)p.in_batch_stride;
    const TileWalk walk = xcd_tile_walk(ntiles, plain_walk);       // (the halo overlaps of neighbouring tiles become L2 hits)
    const int t_end = walk.end, nwg = walk.stride;
    auto issue_halo = [&](int t, int stage) {
        const bool tv = t < t_end;
        const int tt = tv ? t : 0;
        const int b = tt / tiles_img, trem = tt - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        char* base = smem + stage * STAGE;
        int ln = lane;
        asm volatile("" : "+v"(ln));               // per-piece lane arithmetic is recomputed here, not kept in registers across the tile
#pragma unroll
        for (int it = 0; it < P; ++it) {
            int q = wave + it * 8;
            if (q >= PIECES) q -= 8;                  // branch-free partial round: repeat the previous piece (same data)
            const int hr = q * PXPP + ln / SLOTS;
            const int hy = hr / HWD, hx = hr - hy * HWD;
            const int iy = ty * TH * S - 1 + hy, ix = tx * TW * S - 1 + hx;
            const bool v = tv && hr < HR && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int sl = (ln % SLOTS) ^ key(hr);
            const uint32_t off = ((uint32_t)(b * in_bs + iy * p.in_row_stride + ix * p.in_pix_stride + sl * 8) * 2u) | (v ? 0u : kOOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(base + q * 1024), 16, off, 0, 0, 0);
        }
    };
    // fragment f = (tap, ks): lane (lr = pixel of this wave's row, half) reads 16 bytes = channels ks*16 + half*8 .. +7 of the halo
    // pixel at (wave * S + dy, lr * S + dx)
    int a_tap[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int row = (wave * S + tap / 3) * HWD + lr * S + tap % 3;
        a_tap[tap] = row * PITCH + ((half ^ key(row)) << 4);
    }
    auto ld_frag = [&](int stage, int f) {
        return *(const i32x4*)(smem + stage * STAGE + (a_tap[f / KS] ^ ((f % KS) << 5)));
    };
    float* ss = (float*)(smem + NST * STAGE);
    if (tid < 32) {
        ss[tid] = (p.scale && tid < p.Cout) ? p.scale[tid] : 1.f;
        ss[32 + tid] = (p.shift && tid < p.Cout) ? p.shift[tid] : 0.f;
    }
    int t = walk.first;
#pragma unroll
    for (int s0 = 0; s0 < NST; ++s0) issue_halo(t + s0 * nwg, s0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NST - 1) * P) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    i32x4 ring[RING];
#pragma unroll
    for (int f = 0; f < RING; ++f) ring[f] = ld_frag(0, f);

    // VMEM queue per tile k and wave: [D(k + NST): P pieces, after the barrier] [S(k): NS stores].  At tile k's barrier D(k + 1)
    // (issued at tile k + 1 - NST's barrier) must have landed: younger than it are (NST - 2) D's and (NST - 1) S's.
    constexpr int kYounger = (NST - 2) * P + (NST - 1) * NS;
    int stage = 0, k = 0;
    for (; t < t_end; t += nwg, ++k) {
        f32x16 acc;
        const int b = t / tiles_img, trem = t - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int y = ty * TH + wave, x = tx * TW + lr;
        const bool pin = y < p.Ho && x < p.Wo;
        const int nstage = stage + 1 == NST ? 0 : stage + 1;
        static_for<NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            if constexpr (f == NF - RING) {
                if (k < NST) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kYounger) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue_halo(t + NST * nwg, stage);                 // tile k + NST into the stage just released
            }
            if constexpr (f == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                Fmt16<T>::mfma32z(wf[0], ring[0], zero, acc);
            } else
                Fmt16<T>::mfma32(wf[f], ring[f % RING], acc);
            ring[f % RING] = f + RING < NF ? ld_frag(stage, f + RING) : ld_frag(nstage, f + RING - NF);
        });
        // ---- epilogue: lane (lr = pixel, half) holds channels 8g + 4 half + e of its pixel --------------------------------------
        const uint32_t obase = (uint32_t)(((b * p.Ho + y) * p.Wo + x) * p.out_pix_stride);
        const float relu_lo = p.relu ? 0.f : -3.0e38f;
        auto chan4 = [&](int g, float (&v)[4]) {       // channels 8g + 4 half .. +3 of this lane's pixel: scale, shift, ReLU
            const f32x4 sc = *(const f32x4*)(ss + 8 * g + 4 * half), sh = *(const f32x4*)(ss + 32 + 8 * g + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[4 * g + e] * sc[e] + sh[e], relu_lo);
        };
        if constexpr (OUTF32) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
                chan4(g, v);
                const int n = 8 * g + 4 * half;
                const uint32_t off = ((obase + n) * 4u) | (pin && n < p.Cout ? 0u : kOOB);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f32x4{v[0], v[1], v[2], v[3]}), out_rsrc, off, 0, 0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                float va[4], vb[4];
                chan4(g, va);
                chan4(g + 1, vb);
                const int x0 = Fmt16<T>::pack2(va[0], va[1]), x1 = Fmt16<T>::pack2(va[2], va[3]);
                const int y0 = Fmt16<T>::pack2(vb[0], vb[1]), y1 = Fmt16<T>::pack2(vb[2], vb[3]);
                auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                i32x4 o = {(int)r0[0], (int)r1[0], (int)r0[1], (int)r1[1]};
                const int n = 8 * (g + half);
                const uint32_t off = ((obase + n) * 2u) | (pin && n < p.Cout ? 0u : kOOB);
                __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, off, 0, 0);
            }
        }
        stage = nstage;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead DMAs must not land in a successor workgroup's LDS
}


// =====================================================================================================
// v7 "K-split resident weights" kernel for Cin = 256 (ResNet layer3, DLA 256 -> 256; 3x3 / stride 1 / pad 1, 16-bit).
// The 1.18 MB weight panel of a 256 -> 256 conv does not fit one CU; the 4-wave register-resident kernel above handles it as 2
// channel groups x 2 K halves with ONE wave per SIMD, where nothing hides a wave's DMA issue, reduction and epilogue.  Here a
// workgroup is 8 waves, two per SIMD: wave (cg, kq) owns 32 output channels x ONE 64-channel input chunk (36 A-fragments = 144
// registers, as in conv_resident64), 2 channel groups x 4 chunks = 64 output channels per workgroup.  A tile is 8 x 8 output
// pixels (two 32-pixel MFMA blocks per wave, 72 MFMAs); its four 13 KiB chunk halos (10 x 10 pixels x 128 B) are staged by
// LDS-DMA one tile ahead (two stages).  The four K partial sums of a channel group meet in LDS: three waves park their
// accumulators (8 KiB each), the fourth -- the tile's OWNER, rotating with the tile index so that every wave pays the reduction
// and the epilogue on one tile in four -- adds them and runs the fused epilogue while the others already work on the next tile.
constexpr int kKsTH = 8, kKsTW = 8, kKsHW = kKsTW + 2, kKsHR = (kKsTH + 2) * kKsHW;       // 10 x 10 halo pixels
constexpr int kKsImgPieces = (kKsHR + 7) / 8, kKsImg = kKsImgPieces * 1024;                 // 13 pieces = 13 KiB per chunk image
constexpr int kKsStage = 4 * kKsImg, kKsNst = 2;
constexpr int kKsRedOff = kKsNst * kKsStage;                                                // 2 channel groups x 3 partials x 8 KiB
constexpr int kKsSsOff = kKsRedOff + 2 * 3 * 8192;
constexpr int kKsLds = kKsSsOff + 512;

// ABL != 0: timing ablations of the tuning build (WRONG results by construction; VD3D_X_KSPLIT_ABL): 1 = every second pixel-fragment read dropped (the LDS
// traffic of a wave that owned 64 channels x half a chunk), 2 = no pixel-fragment reads in the tile loop, 3 = no K reduction through LDS, 4 = no MFMAs,
// 5 = no owner epilogue (no residual loads, no partial reads, no stores), 6 = no halo DMA in the tile loop, 7 = no barriers in the tile loop, 8 = 5 + 6 + 7.
template <typename T, bool RES, int ABL = 0>
__global__ void __launch_bounds__(512) conv_ksplit256_kernel(const ConvArgs p, int ntiles, int nslices) {
    constexpr int PIECES = 4 * kKsImgPieces;       // 52 per stage
    constexpr int P = (PIECES + 7) / 8;            // 7 per wave (overshoot repeats the wave's previous piece)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 1, kq = wave >> 1;
    const int lr = lane & 31, half = lane >> 5;
    const int tiles_x = (p.W + kKsTW - 1) / kKsTW, tiles_y = (p.H + kKsTH - 1) / kKsTH, tiles_img = tiles_x * tiles_y;
    const int bx = blockIdx.x, xcd = bx & 7, j8 = bx >> 3;
    const int slice = j8 % nslices;
    const int lanes_per_xcd = (int)(gridDim.x >> 3) / nslices;
    const int tlane = xcd * lanes_per_xcd + j8 / nslices, tstride = 8 * lanes_per_xcd;
    const int n_base = slice * 64 + cg * 32;

    i32x4 wf[36];
    {
        const char* wimg = p.wfrag + ((((size_t)(n_base >> 5) * 4 + kq) * 36) * 64 + lane) * 16;
        static_for<36>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            wf[i] = *(const i32x4*)(wimg + i * 1024);
        });
    }
    // swizzle key of a halo row: pixel index (hy * 8 + hx) / 2 mod 8 -- the rows a wave reads for any tap shift then carry
    // consecutive keys (cf. conv_resident64)
    auto hkey = [](int row) { const int hy = row / kKsHW; return ((hy * kKsTW + row - hy * kKsHW) >> 1) & 7; };
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? p.residual : p.out), 0, 0x80000000u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int in_bs = (int)p.in_batch_stride;
    auto tile_origin = [&](int t, int& b, int& ty, int& tx) {
        b = t / tiles_img;
        const int trem = t - b * tiles_img;
        ty = trem / tiles_x;
        tx = trem - ty * tiles_x;
    };
    auto issue_halo = [&](int t, int stage) {
        const bool tv = t < ntiles;
        int b, ty, tx;
        tile_origin(tv ? t : 0, b, ty, tx);
        char* base = smem + stage * kKsStage;
        int ln = lane;
        asm volatile("" : "+v"(ln));               // per-piece lane arithmetic is recomputed here (72 MFMAs apart), not kept in registers
#pragma unroll
        for (int it = 0; it < P; ++it) {
            int q = wave + it * 8;
            if (q >= PIECES) q -= 8;
            const int g = q / kKsImgPieces, pi = q - g * kKsImgPieces;
            const int hr = 8 * pi + (ln >> 3);
            const int hy = hr / kKsHW, hx = hr - hy * kKsHW;
            const int iy = ty * kKsTH - 1 + hy, ix = tx * kKsTW - 1 + hx;
            const bool v = tv && hr < kKsHR && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int col = g * 64 + ((ln & 7) ^ hkey(hr)) * 8;
            const uint32_t off = ((uint32_t)(b * in_bs + iy * p.in_row_stride + ix * p.in_pix_stride + col) * 2u) | (v ? 0u : kOOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(base + g * kKsImg + pi * 1024), 16, off, 0, 0, 0);
        }
    };
    // fragment (tap, ks) of block j: lane (lr, half) reads halo row (4j + lr / 8 + dy, lr % 8 + dx) of THIS wave's chunk image
    int a_tap[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int row = (lr / kKsTW + tap / 3) * kKsHW + (lr & (kKsTW - 1)) + tap % 3;
        a_tap[tap] = kq * kKsImg + row * 128 + ((half ^ hkey(row)) << 4);
    }
    auto ld_frag = [&](int stage, int m) {        // m = f * 2 + j
        const int f = m >> 1, j = m & 1;
        int base = a_tap[f >> 2];
        asm volatile("" : "+v"(base));            // keeps the 36 (tap, ks) addresses from being hoisted out of the tile loop (they spill)
        return *(const i32x4*)(smem + stage * kKsStage + j * (4 * kKsHW * 128) + (base ^ ((f & 3) << 5)));
    };
    float* ss = (float*)(smem + kKsSsOff);
    if (tid < 64) {
        ss[tid] = p.scale ? p.scale[slice * 64 + tid] : 1.f;
        ss[64 + tid] = p.shift ? p.shift[slice * 64 + tid] : 0.f;
    }
    int t = tlane;
    issue_halo(t, 0);
    issue_halo(t + tstride, 1);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(P) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    constexpr int RING = 2;                        // (registers: 144 weights + 32 accumulators leave room for a 2-deep ring; 4 spills and measured slower)
    i32x4 ring[RING];
#pragma unroll
    for (int m = 0; m < RING; ++m) ring[m] = ld_frag(0, m);
    const float relu_lo = p.relu ? 0.f : -3.0e38f;

    int k = 0;
    for (; t < ntiles; t += tstride, ++k) {
        const int stage = k & 1;
        const int owner = k & 3;                   // the K quarter whose waves finish this tile
        const bool own = kq == owner;
        f32x16 acc[2];
        int b, ty, tx;
        tile_origin(t, b, ty, tx);
        static_for<72>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            if constexpr (m == 72 - RING) {
                // all reads of this stage are issued (and, with lgkmcnt(0), done); the next tile's halos (issued one tile ago) must
                // have landed.  The queue differs per wave (residual loads / stores of the owners): drain it.
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if constexpr (ABL != 7 && ABL != 8) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (ABL != 6 && ABL != 8) issue_halo(t + 2 * tstride, stage);
            }
            constexpr int f = m >> 1, j = m & 1;
            if constexpr (ABL == 4) {
                if constexpr (f == 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][e] = __builtin_bit_cast(float, ring[m % RING][e & 3]);
                }
            } else if constexpr (f == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                Fmt16<T>::mfma32z(wf[0], ring[m % RING], zero, acc[j]);
            } else
                Fmt16<T>::mfma32(wf[f], ring[m % RING], acc[j]);
            if constexpr (ABL == 0 || ABL >= 3 || (ABL == 1 && (m & 2) == 0))
                ring[m % RING] = m + RING < 72 ? ld_frag(stage, m + RING) : ld_frag(stage ^ 1, m + RING - 72);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
        // ---- the four K partial sums of a channel group meet in LDS: 3 writers, the owner adds --------------------------------
        char* red = smem + kKsRedOff + cg * (3 * 8192);
        // the owner's output offsets and residual (8-byte loads, in flight across the reduction barrier; loading them at the top
        // of the tile would hold 16 registers through the MFMA loop and spill)
        uint32_t o_off[2];
        i32x2 rr[2][4];
        if (own && ABL != 5 && ABL != 8) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = ty * kKsTH + 4 * j + lr / kKsTW, x = tx * kKsTW + (lr & (kKsTW - 1));
                const bool pin = y < p.H && x < p.W;
                const int mpix = (b * p.H + y) * p.W + x;
                o_off[j] = ((uint32_t)(mpix * p.out_pix_stride + n_base) * 2u) | (pin ? 0u : kOOB);
                if constexpr (RES) {
                    const uint32_t r_off = ((uint32_t)(mpix * p.res_pix_stride + n_base + 4 * half) * 2u) | (pin ? 0u : kOOB);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        rr[j][g] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(res_rsrc, r_off + 16 * g, 0, 0));
                }
            }
        }
        if (ABL != 3 && !own) {
            char* mine = red + ((kq - owner - 1) & 3) * 8192;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                    *(f32x4*)(mine + ((j * 4 + g) * 64 + lane) * 16) = v;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (ABL != 7 && ABL != 8) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (own && ABL != 5 && ABL != 8) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int pk[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4] = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
#pragma unroll
                    for (int w = 0; w < (ABL == 3 ? 0 : 3); ++w) {
                        const f32x4 q = *(const f32x4*)(red + w * 8192 + ((j * 4 + g) * 64 + lane) * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += q[e];
                    }
                    const int nl = cg * 32 + 4 * half + 8 * g;
                    const f32x4 sc = *(const f32x4*)(ss + nl), sh = *(const f32x4*)(ss + 64 + nl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
                    if constexpr (RES) {
                        const uint32_t r0 = (uint32_t)(int)rr[j][g][0], r1 = (uint32_t)(int)rr[j][g][1];
                        v[0] += Fmt16<T>::lo(r0);
                        v[1] += Fmt16<T>::hi(r0);
                        v[2] += Fmt16<T>::lo(r1);
                        v[3] += Fmt16<T>::hi(r1);
                    }
                    pk[g][0] = max_pk16(Fmt16<T>::pack2_1(v[0], v[1]), p.relu ? 0 : (int)0x80008000u);      // round, then ReLU on the packed pairs (bit-identical)
                    pk[g][1] = max_pk16(Fmt16<T>::pack2_1(v[2], v[3]), p.relu ? 0 : (int)0x80008000u);
                }
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    auto r0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                    i32x4 o = {(int)r0[0], (int)r1[0], (int)r0[1], (int)r1[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, o_off[j] + 16 * (g + half), 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// =====================================================================================================
// Point-wise (1x1 / stride 1) expansion convolutions with few input channels (ResNet-50 bottleneck conv3 and the first
// stage's down-sample branch: 64 -> 256 at 64 x 72 x 320, 128 -> 512): K is one or two 64-deep slices, the layer moves
// 0.9 - 1.7 GB (output + residual dominate) and the tile kernels -- one workgroup per CU, load -> MFMA -> epilogue in
// sequence, nothing in flight across tiles -- reach 2.1 - 2.8 TB/s of the ~5 TB/s a mixed read / write stream gets here.
//   * no LDS, no barriers: a wave keeps its 64 output channels x CIN weights in registers (MFMA A operand, 32 | 64 VGPRs)
//     and walks 32-pixel blocks; the pixel operand is loaded straight into the MFMA B layout (lane = pixel, 16 bytes = 8
//     consecutive input channels: the four sub-step loads of a 64-channel chunk cover the pixel's 128-byte line);
//   * the four waves of a workgroup take the same pixel block and four different 64-channel groups (256 channels per
//     workgroup; the slices of one block sequence sit on the same XCD), so the block's input is fetched once per CU;
//   * the next block's pixel fragments and residual quads are requested before the current block's MFMAs: plain loads, the
//     compiler's own vmcnt bookkeeping; 3 - 4 workgroups per CU keep > 200 KB in flight per CU.
constexpr int kPwRow = 144;      // bytes per pixel row of the output staging tile (128 + pad: 16-byte aligned rows, 2-way bank conflicts)
template <typename T, int CIN, bool RES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CIN == 64 ? 3 : (CIN == 128 ? 2 : 1), 8))) conv_pw_kernel(const ConvArgs p, int nblk, int nslices) {
    constexpr int NCH = CIN / 64;
    __shared__ __attribute__((aligned(16))) float ss[512];     // this slice's folded BN scale | shift
    __shared__ __attribute__((aligned(16))) char otile[4][32 * kPwRow];   // per wave: one block's outputs, [pixel][64 channels] (+ pad)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, half = lane >> 5;
    // blockIdx -> (slice, block lane): blocks b, b + 8, ... share an XCD; consecutive ones take the slices of one block lane
    const int bx = blockIdx.x, xcd = bx & 7, j8 = bx >> 3;
    const int slice = j8 % nslices;
    const int lanes_per_xcd = (int)(gridDim.x >> 3) / nslices;
    const int blane = xcd * lanes_per_xcd + j8 / nslices, bstride = 8 * lanes_per_xcd;
    const int n_base = slice * 256 + wave * 64;                // first output channel of this wave

    i32x4 wf[2][NCH][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                wf[cb][c][ks] = *(const i32x4*)(p.wfrag + ((((size_t)((n_base >> 5) + cb) * NCH + c) * 4 + ks) * 64 + lane) * 16);
    ss[tid] = p.scale ? p.scale[slice * 256 + tid] : 1.f;
    ss[256 + tid] = p.shift ? p.shift[slice * 256 + tid] : 0.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? p.residual : p.out), 0, 0x80000000u, 0x00020000);

    struct Blk { i32x4 b[NCH][4]; };                  // pixel fragments of one 32-pixel block
    auto request = [&](int blk, Blk& q) {
        const int m = blk * 32 + lr;
        const bool ok = blk < nblk && m < p.M;
        const uint32_t boff = ok ? (uint32_t)(m * p.in_pix_stride * 2 + half * 16) : kOOB;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                q.b[c][ks] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, boff, c * 128 + ks * 32, 0));
    };
    // residual in the STORE layout (16 bytes = channels 16 j + 8 half .. + 7).  Requested BEFORE the next block's pixel fragments:
    // loads return in order, so the epilogue's wait for these four leaves the sixteen younger fragment loads in flight
    struct Res { i32x4 v[2][2]; };
    auto request_res = [&](int blk, Res& rs) {
        if constexpr (RES) {
            const int m = blk * 32 + lr;
            const uint32_t roff = (blk < nblk && m < p.M) ? (uint32_t)((m * p.res_pix_stride + n_base + 8 * half) * 2) : kOOB;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    rs.v[cb][j] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, roff, (cb * 32 + 16 * j) * 2, 0));
        }
    };
    const int relu_floor = p.relu ? 0 : (int)0x80008000u;      // packed-pair floor: ReLU | identity
    auto finish = [&](int blk, const Blk& q, const Res& rs) {
        f32x16 acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) Fmt16<T>::mfma32(wf[cb][c][ks], q.b[c][ks], acc[cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            int pk[4][2];
            int rq[4][2];                              // residual back in the accumulator layout (the store pairing is an involution)
            if constexpr (RES) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    auto a0 = __builtin_amdgcn_permlane32_swap(rs.v[cb][j][0], rs.v[cb][j][2], false, false);
                    auto a1 = __builtin_amdgcn_permlane32_swap(rs.v[cb][j][1], rs.v[cb][j][3], false, false);
                    rq[2 * j][0] = (int)a0[0]; rq[2 * j + 1][0] = (int)a0[1];
                    rq[2 * j][1] = (int)a1[0]; rq[2 * j + 1][1] = (int)a1[1];
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int nl = wave * 64 + cb * 32 + 8 * g + 4 * half;                // channel inside the slice
                asm volatile("" : "+v"(nl));           // (opaque: else the 16 constant vectors are hoisted out of the block loop -- 64 registers)
                const f32x4 sc = *(const f32x4*)(ss + nl), sh = *(const f32x4*)(ss + 256 + nl);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[cb][4 * g + e] * sc[e] + sh[e];
                if constexpr (RES) {
                    const uint32_t r0 = (uint32_t)rq[g][0], r1 = (uint32_t)rq[g][1];
                    v[0] += Fmt16<T>::lo(r0);
                    v[1] += Fmt16<T>::hi(r0);
                    v[2] += Fmt16<T>::lo(r1);
                    v[3] += Fmt16<T>::hi(r1);
                }
                // round (one packed conversion per pair), then ReLU on the packed pairs (relu(round(x)) == round(relu(x)): bit-identical to fmaxf first;
                // fmaxf on an accumulator costs a canonicalise + a max per value, v_pk_max_i16 one instruction per two)
                pk[g][0] = max_pk16(Fmt16<T>::pack2_1(v[0], v[1]), relu_floor);
                pk[g][1] = max_pk16(Fmt16<T>::pack2_1(v[2], v[3]), relu_floor);
            }
            // through the wave's LDS tile: the MFMA layout gives a lane 8 bytes of one pixel; stored from there an instruction
            // writes 32 bytes into each of 32 lines.  Re-read row-major, a store instruction covers 8 pixels x 128 bytes: whole lines
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(i32x2*)(otile[wave] + lr * kPwRow + (cb * 32 + 8 * g + 4 * half) * 2) = i32x2{pk[g][0], pk[g][1]};
        }
        {
            const int prow = lane >> 3, pslot = lane & 7;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int mm = blk * 32 + it * 8 + prow;
                const i32x4 o = *(const i32x4*)(otile[wave] + (it * 8 + prow) * kPwRow + pslot * 16);
                const uint32_t off = mm < p.M ? (uint32_t)((mm * p.out_pix_stride + n_base) * 2 + pslot * 16) : kOOB;
                __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, off, 0, 0);
            }
        }
    };
    Blk q0, q1;
    Res rs;
    int blk = blane;
    request(blk, q0);
    for (; blk < nblk; blk += 2 * bstride) {
        request_res(blk, rs);
        request(blk + bstride, q1);
        finish(blk, q0, rs);
        if (blk + bstride >= nblk) break;
        request_res(blk + bstride, rs);
        request(blk + 2 * bstride, q0);
        finish(blk + bstride, q1, rs);
    }
}

}  // namespace

namespace vd3d_conv {

constexpr int regw_lds(int cin, bool res) {
    const int cg = cin == 256 ? 2 : 4;
    return 4 * kRwImg + (cin == 256 ? cg * 16384 : 0) + (res ? cg * 8192 : 0) + 2 * cg * 32 * 4;
}

template <typename T, int CIN, int RWRING = 4, int ABL = 0>
static int launch_regw_t(ConvArgs& a, hipStream_t stream) {
    constexpr int CSL = CIN == 128 ? 128 : 64;
    constexpr int LDS_R = regw_lds(CIN, true), LDS_N = regw_lds(CIN, false);
    static Vd3dLdsLimit lim_res, lim_nores;
    int rc = vd3d_raise_lds_limit((const void*)conv_regw_kernel<T, CIN, true, RWRING, ABL>, LDS_R, lim_res, "hipFuncSetAttribute(conv_regw)");
    if (!rc) rc = vd3d_raise_lds_limit((const void*)conv_regw_kernel<T, CIN, false, RWRING, ABL>, LDS_N, lim_nores, "hipFuncSetAttribute(conv_regw)");
    if (rc) return rc;
    const int num_cu = vd3d_device_cu_count();
    if (num_cu <= 0) return VD3D_ELAUNCH;
    const int nslices = a.Cout / CSL;
    const int ntiles = a.B * ((a.H + kResTH - 1) / kResTH) * ((a.W + kResTW - 1) / kResTW);
    // grid = 8 XCDs x lanes x slices, at most one workgroup per CU, no more tile lanes than tiles
    int lanes = num_cu / (8 * nslices);
    const int need = (ntiles + 7) / 8;
    if (lanes > need) lanes = need;
    if (lanes < 1) lanes = 1;
    const int grid = 8 * lanes * nslices;
    if (a.residual) hipLaunchKernelGGL((conv_regw_kernel<T, CIN, true, RWRING, ABL>), dim3(grid), dim3(256), LDS_R, stream, a, ntiles, nslices);
    else hipLaunchKernelGGL((conv_regw_kernel<T, CIN, false, RWRING, ABL>), dim3(grid), dim3(256), LDS_N, stream, a, ntiles, nslices);
    return vd3d_check_launch("conv_regw");
}

bool regw_shape_ok(const ConvArgs& a) {
    const int csl = a.Cin == 128 ? 128 : 64;
    return (a.Cin == 128 || a.Cin == 256) && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.Ho == a.H &&
           a.Wo == a.W && a.Cout % csl == 0 && a.Cout / csl <= 32 && a.wide_store && !a.out_f32 && a.wfrag &&
           (int64_t)a.M * a.out_pix_stride * 2 < 0x7ffffff0ll && (!a.residual || (int64_t)a.M * a.res_pix_stride * 2 < 0x7ffffff0ll);
}

template <typename T>
static int launch_resident64_t(ConvArgs& a, hipStream_t stream) {
    static Vd3dLdsLimit lim_res, lim_nores;
    int rc = vd3d_raise_lds_limit((const void*)conv_resident64_kernel<T, true>, kResLds, lim_res, "hipFuncSetAttribute(conv_resident64)");
    if (!rc) rc = vd3d_raise_lds_limit((const void*)conv_resident64_kernel<T, false>, kResLds, lim_nores, "hipFuncSetAttribute(conv_resident64)");
    if (rc) return rc;
    const int num_cu = vd3d_device_cu_count();
    if (num_cu <= 0) return VD3D_ELAUNCH;
    const int ntiles = a.B * ((a.H + kResTH - 1) / kResTH) * ((a.W + kResTW - 1) / kResTW);
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    const int plain_walk = plain_tile_walk();
    if (a.residual) hipLaunchKernelGGL((conv_resident64_kernel<T, true>), dim3(grid), dim3(512), kResLds, stream, a, ntiles, plain_walk);
    else hipLaunchKernelGGL((conv_resident64_kernel<T, false>), dim3(grid), dim3(512), kResLds, stream, a, ntiles, plain_walk);
    return vd3d_check_launch("conv_resident64");
}

int launch_resident64(ConvArgs& a, hipStream_t stream, int fmt) {
    return fmt == VD3D_F16 ? launch_resident64_t<hf16>(a, stream) : launch_resident64_t<short>(a, stream);
}


bool small_shape_ok(const ConvArgs& a) {
    const bool out16 = !a.out_f32;
    return (a.Cin == 16 || a.Cin == 32 || a.Cin == 64) && a.kh == 3 && a.kw == 3 && (a.stride == 1 || a.stride == 2) && a.pad == 1 && a.dil == 1 &&
           a.Cout <= 32 && a.Cout % (out16 ? 16 : 4) == 0 && !a.residual && a.in_pix_stride % 8 == 0 && a.out_pix_stride % (out16 ? 8 : 4) == 0 &&
           (!a.scale || ((uintptr_t)a.scale & 15) == 0) && (!a.shift || ((uintptr_t)a.shift & 15) == 0) &&
           ((uintptr_t)a.out & 15) == 0 && (int64_t)a.M * a.out_pix_stride * (out16 ? 2 : 4) < 0x7ffffff0ll &&
           !(a.stride == 2 && a.Cin == 64);
}

template <typename T, int CIN, int S, bool OUTF32>
static int launch_small_t(ConvArgs& a, hipStream_t stream) {
    constexpr int HR = ((8 - 1) * S + 3) * ((32 - 1) * S + 3), PXPP = 1024 / (CIN * 2), STAGE = ((HR + PXPP - 1) / PXPP) * 1024;
    constexpr int NST = (150 * 1024 / STAGE) >= 4 ? 4 : ((150 * 1024 / STAGE) >= 3 ? 3 : 2);
    constexpr int LDS = NST * STAGE + 512;
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)conv_small_kernel<T, CIN, S, OUTF32>, LDS, lim, "hipFuncSetAttribute(conv_small)")) return rc;
    const int num_cu = vd3d_device_cu_count();
    if (num_cu <= 0) return VD3D_ELAUNCH;
    const int ntiles = a.B * ((a.Ho + 7) / 8) * ((a.Wo + 31) / 32);
    // small halo stages leave room for two workgroups per CU (16 waves hide the DMA latency of these short tiles better)
    const int per_cu = LDS <= 72 * 1024 ? 2 : 1;
    const int grid = ntiles < num_cu * per_cu ? ntiles : num_cu * per_cu;
    hipLaunchKernelGGL((conv_small_kernel<T, CIN, S, OUTF32>), dim3(grid), dim3(512), LDS, stream, a, ntiles, plain_tile_walk());
    return vd3d_check_launch("conv_small");
}

template <typename T>
static int launch_small_f(ConvArgs& a, hipStream_t stream) {
#define VD3D_SMALL(CIN, S) \
    if (a.Cin == CIN && a.stride == S) return a.out_f32 ? launch_small_t<T, CIN, S, true>(a, stream) : launch_small_t<T, CIN, S, false>(a, stream);
    VD3D_SMALL(16, 1) VD3D_SMALL(32, 1) VD3D_SMALL(64, 1) VD3D_SMALL(16, 2) VD3D_SMALL(32, 2)
#undef VD3D_SMALL
    vd3d_set_error("conv_small: unsupported (Cin, stride)");
    return VD3D_EINVAL;
}

int launch_small(ConvArgs& a, hipStream_t stream, int fmt) {
    return fmt == VD3D_F16 ? launch_small_f<hf16>(a, stream) : launch_small_f<short>(a, stream);
}


// =====================================================================================================
// "narrow-output streaming" kernel (tile id 70): 3x3 / s1 / p1, Cin a multiple of 64 (>= 128), Cout <= 32 -- the DCN offset convs of the
// deeper DLA-Up levels (128 | 256 | 512 -> 27) and of the stereo base head (2176 -> 27).  The generic 256 x 32 tile kernel re-stages
// every input pixel nine times (once per tap) and ran these at 3-19x their HBM bound (512 -> 32 at 16 x 16 x 55: 60 us for 14 MB).
// Here: the small-channel kernel's tile (8 waves x one 32-pixel row, 8 x 32 output pixels, halo staged ONCE) walked over 64-channel
// CHUNKS: an item = (tile, chunk); per item the 10 x 34-pixel halo of that chunk (43 KiB) and the chunk's 36 weight fragments (36 KiB,
// straight from the register image `weight_frag` -- lane-linear, conflict-free ds_read_b128) arrive by LDS-DMA into one of two stages
// while the previous item computes: 36 MFMAs per wave and item, accumulators carried over the chunks of a tile, epilogue on the last.
constexpr int kNwTH = 8, kNwTW = 32, kNwHW = kNwTW + 2, kNwHR = (kNwTH + 2) * kNwHW;      // 10 x 34 halo pixels
constexpr int kNwHP = (kNwHR + 7) / 8, kNwWP = 36, kNwPieces = kNwHP + kNwWP;             // 43 + 36 DMA pieces of 1 KiB
constexpr int kNwStage = kNwPieces * 1024, kNwLds = 2 * kNwStage + 512;

template <typename T, bool OUTF32>
__global__ void __launch_bounds__(512) conv_narrow_kernel(const ConvArgs p, int ntiles, int nchunks) {
    constexpr int P = (kNwPieces + 7) / 8;             // per wave (overshoot repeats the wave's previous piece)
    constexpr int NS = OUTF32 ? 4 : 2;
    (void)NS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, half = lane >> 5;
    const int tiles_x = (p.Wo + kNwTW - 1) / kNwTW, tiles_y = (p.Ho + kNwTH - 1) / kNwTH, tiles_img = tiles_x * tiles_y;
    auto key = [](int row) { return (row >> 1) & 7; };
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wfrag, 0, (uint32_t)nchunks * kNwWP * 1024u, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x80000000u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int in_bs = (int)p.in_batch_stride;
    // item u of this workgroup = (its k-th tile, chunk c): u = k * nchunks + c
    const int nwg = gridDim.x;
    auto issue_item = [&](int t, int c, int stage) {
        const bool tv = t < ntiles;
        const int tt = tv ? t : 0;
        const int b = tt / tiles_img, trem = tt - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        char* base = smem + stage * kNwStage;
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int it = 0; it < P; ++it) {
            int q = wave + it * 8;
            if (q >= kNwPieces) q -= 8;               // branch-free partial round: repeat the previous piece (same data)
            uint32_t off;
            if (q < kNwHP) {                          // wave-uniform: a halo piece (8 pixels x 128 B of chunk c)
                const int hr = q * 8 + (ln >> 3);
                const int hy = hr / kNwHW, hx = hr - hy * kNwHW;
                const int iy = ty * kNwTH - 1 + hy, ix = tx * kNwTW - 1 + hx;
                const bool v = tv && hr < kNwHR && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const int sl = (ln & 7) ^ key(hr);
                off = ((uint32_t)(b * in_bs + iy * p.in_row_stride + ix * p.in_pix_stride + c * 64 + sl * 8) * 2u) | (v ? 0u : kOOB);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(base + q * 1024), 16, off, 0, 0, 0);
            } else {                                  // a weight fragment of chunk c: 1 KiB, lane-linear
                off = tv ? (uint32_t)((c * kNwWP + (q - kNwHP)) * 1024 + ln * 16) : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(base + q * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    int a_tap[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int row = (wave + tap / 3) * kNwHW + lr + tap % 3;
        a_tap[tap] = row * 128 + ((half ^ key(row)) << 4);
    }
    float* ss = (float*)(smem + 2 * kNwStage);
    if (tid < 32) {
        ss[tid] = (p.scale && tid < p.Cout) ? p.scale[tid] : 1.f;
        ss[32 + tid] = (p.shift && tid < p.Cout) ? p.shift[tid] : 0.f;
    }
    int t = blockIdx.x, c = 0, stage = 0;
    issue_item(t, 0, 0);
    f32x16 acc;
    for (; t < ntiles;) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this item's DMA was issued a whole item ago
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int cn = c + 1 == nchunks ? 0 : c + 1, tn = cn == 0 ? t + nwg : t;
        issue_item(tn, cn, stage ^ 1);
        const char* H0 = smem + stage * kNwStage;
        const char* W0 = H0 + kNwHP * 1024;
        static_for<36>([&](auto fc) {
            constexpr int f = decltype(fc)::value, tap = f / 4, ks = f % 4;
            const i32x4 fa = *(const i32x4*)(W0 + (f * 64 + lane) * 16);
            const i32x4 fb = *(const i32x4*)(H0 + (a_tap[tap] ^ (ks << 5)));
            if (f == 0 && c == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                Fmt16<T>::mfma32z(fa, fb, zero, acc);
            } else
                Fmt16<T>::mfma32(fa, fb, acc);
        });
        if (cn == 0) {
            // ---- epilogue (as conv_small): lane (lr = pixel, half) holds channels 8g + 4 half + e of its pixel
            const int b = t / tiles_img, trem = t - b * tiles_img;
            const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
            const int y = ty * kNwTH + wave, x = tx * kNwTW + lr;
            const bool pin = y < p.Ho && x < p.Wo;
            const uint32_t obase = (uint32_t)(((b * p.Ho + y) * p.Wo + x) * p.out_pix_stride);
            const float relu_lo = p.relu ? 0.f : -3.0e38f;
            auto chan4 = [&](int g, float (&v)[4]) {
                const f32x4 sc = *(const f32x4*)(ss + 8 * g + 4 * half), sh = *(const f32x4*)(ss + 32 + 8 * g + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[4 * g + e] * sc[e] + sh[e], relu_lo);
            };
            if constexpr (OUTF32) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
                    chan4(g, v);
                    const int n = 8 * g + 4 * half;
                    const uint32_t off = ((obase + n) * 4u) | (pin && n < p.Cout ? 0u : kOOB);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f32x4{v[0], v[1], v[2], v[3]}), out_rsrc, off, 0, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    float va[4], vb[4];
                    chan4(g, va);
                    chan4(g + 1, vb);
                    const int x0 = Fmt16<T>::pack2(va[0], va[1]), x1 = Fmt16<T>::pack2(va[2], va[3]);
                    const int y0 = Fmt16<T>::pack2(vb[0], vb[1]), y1 = Fmt16<T>::pack2(vb[2], vb[3]);
                    auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                    i32x4 o = {(int)r0[0], (int)r1[0], (int)r0[1], (int)r1[1]};
                    const int n = 8 * (g + half);
                    const uint32_t off = ((obase + n) * 2u) | (pin && n < p.Cout ? 0u : kOOB);
                    __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, off, 0, 0);
                }
            }
        }
        t = tn;
        c = cn;
        stage ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead DMA must not land in a successor workgroup's LDS
}

// =====================================================================================================
// "level pair" kernel: conv A (3x3 / s1 / p1, 16 -> 16) + BN + ReLU followed by conv B (3x3 / s2 / p1, 16 -> <= 32) + BN + ReLU in ONE
// launch -- DLA level0 -> level1 (backbones/dla.py:118-121, :142-148) at full / half resolution: 16 x 512 x 1760 x 16 channels are
// 461 MB that the two small-channel launches wrote and read back (197 + 132 us, both HBM bound).  A workgroup owns 8 x 32 outputs of
// conv B = 17 x 65 outputs of conv A (1.08x the pixels a non-overlapping split would compute) = a 19 x 67-pixel halo of the input,
// staged by LDS-DMA ONE tile ahead into a single 40 KiB halo stage (the next tile's halo is requested after phase 1 has consumed the
// current one; two workgroups of 4 waves share a CU, so the partner's phases cover the latency).  Phase 1: conv A on the matrix cores from the halo (32-pixel blocks dealt to
// the waves, weights in registers), BN + ReLU, rounded to the 16-bit format (the rounding point of the unfused tensor), positions
// outside the image zeroed (conv B's zero padding), into a [17 x 65][16] LDS image.  Phase 2: the small-channel kernel's stride-2
// step from that image.  Same MFMA shapes and tap order as the two launches: results are bit-identical.
constexpr int kPrTH = 8, kPrTW = 32;
constexpr int kPrYH = 2 * kPrTH + 1, kPrYW = 2 * kPrTW + 1, kPrYN = kPrYH * kPrYW;            // 17 x 65 conv-A outputs
constexpr int kPrXH = kPrYH + 2, kPrXW = kPrYW + 2, kPrXN = kPrXH * kPrXW;                    // 19 x 67 input halo
constexpr int kPrXP = 2 * ((kPrXN + 63) / 64), kPrXStage = kPrXP * 1024;                     // 2 x 20 pieces of 64 pixels x 16 B
constexpr int kPrNW = 4;                                                                     // waves per workgroup; TWO workgroups per CU
// LDS images are PLANAR in the 8-channel half (the MFMA k-group of a lane): lanes 0-31 of a fragment read then walk consecutive 16-byte
// slots -- the pixel-major [pixel][32 B] image of the first version put the lanes 32 bytes apart, a 2-way bank conflict on every
// ds_read_b128 (4-way for the stride-2 reads of conv B): 57 % of the LDS cycles were conflict cycles (SQ_LDS_BANK_CONFLICT), LDS 64 % busy.
// The conv-A image is additionally split by column parity, so that conv B's stride-2 fragment reads are contiguous as well.
constexpr int kPrXPlane = kPrXStage / 2;                                                     // 20 pieces of 64 pixels x 16 B per half
constexpr int kPrYC = kPrTW + 1;                                                             // 33 columns per parity
constexpr int kPrYQ = ((kPrYH * kPrYC * 16 + 255) / 256) * 256 + 128;                        // parity plane (odd multiple of 128 B)
constexpr int kPrYP = 2 * kPrYQ;                                                             // half plane
constexpr int kPrYB = (kPrYN + 31) / 32, kPrYOff = kPrXStage, kPrYBytes = 2 * kPrYP;         // 35 blocks
constexpr int kPrSS = kPrYOff + kPrYBytes, kPrLds = kPrSS + 512;                              // 76.5 KiB

template <typename T>
__global__ void __launch_bounds__(kPrNW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_pair_kernel(const ConvArgs pa, const ConvArgs pb, int ntiles) {
    constexpr int P = kPrXP / kPrNW;                  // DMA pieces per wave and tile
    constexpr int NB = (kPrYB + kPrNW - 1) / kPrNW;   // conv-A blocks per wave (the last one only for the first waves)
    constexpr int RW = kPrTH / kPrNW;                 // conv-B output rows per wave
    static_assert(kPrXP % kPrNW == 0 && kPrTH % kPrNW == 0, "halo pieces / output rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, half = lane >> 5;
    const int H = pa.H, W = pa.W;                      // conv A: H x W -> H x W;  conv B: -> Ho x Wo
    const int tiles_x = (pb.Wo + kPrTW - 1) / kPrTW, tiles_y = (pb.Ho + kPrTH - 1) / kPrTH, tiles_img = tiles_x * tiles_y;
    // ---- weights -> registers (A operand: row = output channel, zero rows beyond Cout; one 16-channel k-step per tap) ----------------
    i32x4 wa[9], wb[9];
    {
        const char* ra = pa.weight + (size_t)lr * pa.Kpad * 2;
        const char* rb = pb.weight + (size_t)lr * pb.Kpad * 2;
        static_for<9>([&](auto tc) {
            constexpr int tap = decltype(tc)::value;
            wa[tap] = *(const i32x4*)(ra + (tap * 16 + half * 8) * 2);
            wb[tap] = *(const i32x4*)(rb + (tap * 16 + half * 8) * 2);
        });
    }
    float* ss = (float*)(smem + kPrSS);                // A: scale[32] shift[32] | B: scale[32] shift[32]
    if (tid < 32) {
        ss[tid] = (pa.scale && tid < pa.Cout) ? pa.scale[tid] : 1.f;
        ss[32 + tid] = (pa.shift && tid < pa.Cout) ? pa.shift[tid] : 0.f;
        ss[64 + tid] = (pb.scale && tid < pb.Cout) ? pb.scale[tid] : 1.f;
        ss[96 + tid] = (pb.shift && tid < pb.Cout) ? pb.shift[tid] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)pa.in, 0, pa.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)pb.out, 0, 0x80000000u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int in_bs = (int)pa.in_batch_stride;
    // a lane's halo pixels are the same in every tile (piece q = wave + NW it = (half plane, 64-pixel run), pixel = run * 64 + lane): position and source offset
    // relative to the tile origin are computed once (per tile they were constant divisions and ~30 VALU instructions per piece)
    int h_yx[P], h_off[P];
#pragma unroll
    for (int it = 0; it < P; ++it) {
        const int q2 = wave + it * kPrNW, plane = q2 / (kPrXP / 2);
        const int hr = (q2 - plane * (kPrXP / 2)) * 64 + lane;
        const int hy = hr / kPrXW, hx = hr - hy * kPrXW;
        h_yx[it] = hr < kPrXN ? (hy << 16) | hx : -1;
        h_off[it] = hy * pa.in_row_stride + hx * pa.in_pix_stride + plane * 8;
    }
    auto issue_halo = [&](int t) {
        const bool tv = t < ntiles;
        const int tt = tv ? t : 0;
        const int b = tt / tiles_img, trem = tt - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int y0 = 2 * ty * kPrTH - 2, x0 = 2 * tx * kPrTW - 2;
        const int org = b * in_bs + y0 * pa.in_row_stride + x0 * pa.in_pix_stride;
#pragma unroll
        for (int it = 0; it < P; ++it) {
            const int q = wave + it * kPrNW;
            const int iy = y0 + (h_yx[it] >> 16), ix = x0 + (h_yx[it] & 0xffff);
            const bool v = tv && h_yx[it] >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const uint32_t off = ((uint32_t)(org + h_off[it]) * 2u) | (v ? 0u : kOOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(smem + q * 1024), 16, off, 0, 0, 0);
        }
    };
    // phase-2 fragment addresses (as the small-channel kernel with S = 2): wave = RW output rows, lane = output pixel
    int b_tap[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        b_tap[tap] = half * kPrYP + ((tap % 3) & 1) * kPrYQ + ((wave * RW * 2 + tap / 3) * kPrYC + lr + ((tap % 3) >> 1)) * 16;
    const int floor_a = pa.relu ? 0 : (int)0x80008000u, floor_b = pb.relu ? 0 : (int)0x80008000u;     // packed-pair floors: ReLU | identity
    const int nwg = gridDim.x;
    int t = blockIdx.x;
    issue_halo(t);
    // VMEM queue of a wave per tile: D(k + 1) [P pieces, after phase 1 of tile k] S(k) [2 RW stores].  At the top of tile k the halo D(k)
    // must have landed: younger are only the stores of tile k - 1.  The FIRST tile has no such stores behind its halo, so the counted
    // wait below would let up to 2 RW of its P pieces still be in flight: drain the prologue's DMA explicitly (the compiler happened to
    // put a vmcnt(0) in the loop preheader for the weight loads; nothing guaranteed it).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; t < ntiles; t += nwg) {
        const int b = t / tiles_img, trem = t - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * RW) : "memory");
        __builtin_amdgcn_s_barrier();                                     // ... for everybody; the conv-A image is free again
        asm volatile("" ::: "memory");
        // ---- phase 1: conv A on 32-pixel blocks of the 17 x 65 region -------------------------------------------------------------
        // Software pipeline over the wave's blocks (wave, wave + NW, ...): the nine MFMAs of block j + 1 are ISSUED before the VALU
        // epilogue of block j, so the matrix pipe works through them while the wave converts and stores.
        const char* X = smem;
        const f32x4 s0 = *(const f32x4*)(ss + 4 * half), h0 = *(const f32x4*)(ss + 32 + 4 * half);
        const f32x4 s1 = *(const f32x4*)(ss + 8 + 4 * half), h1 = *(const f32x4*)(ss + 40 + 4 * half);
        auto chain = [&](int blk, f32x16& acc) {
            const int q = blk * 32 + lr;
            const int qc = q < kPrYN ? q : kPrYN - 1;
            const int ry = qc / kPrYW, rx = qc - ry * kPrYW;
            const char* xq = X + half * kPrXPlane + (ry * kPrXW + rx) * 16;
            static_for<9>([&](auto tc) {
                constexpr int tap = decltype(tc)::value;
                const i32x4 fb = *(const i32x4*)(xq + ((tap / 3) * kPrXW + tap % 3) * 16);
                if constexpr (tap == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    Fmt16<T>::mfma32z(wa[0], fb, zero, acc);
                } else
                    Fmt16<T>::mfma32(wa[tap], fb, acc);
            });
        };
        auto finish = [&](int blk, const f32x16& acc) {
            // lane (pixel, half) holds channels 8 g + 4 half + e (g = 0, 1 real): BN + ReLU, 16-bit, zero outside the image
            const int q = blk * 32 + lr;
            const int qc = q < kPrYN ? q : kPrYN - 1;
            const int ry = qc / kPrYW, rx = qc - ry * kPrYW;
            const int ay = 2 * ty * kPrTH - 1 + ry, ax = 2 * tx * kPrTW - 1 + rx;
            const bool inside = (unsigned)ay < (unsigned)H && (unsigned)ax < (unsigned)W;
            float va[4], vb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                va[e] = acc[e] * s0[e] + h0[e];
                vb[e] = acc[4 + e] * s1[e] + h1[e];
            }
            // round (one packed conversion per pair), ReLU on the packed pairs, zero outside the image: 12 instructions for 8 values
            // (fmaxf + select per value + two conversions and a pack per pair were 36: the kernel is VALU-issue bound)
            const int x0 = inside ? max_pk16(Fmt16<T>::pack2_1(va[0], va[1]), floor_a) : 0, x1 = inside ? max_pk16(Fmt16<T>::pack2_1(va[2], va[3]), floor_a) : 0;
            const int y0 = inside ? max_pk16(Fmt16<T>::pack2_1(vb[0], vb[1]), floor_a) : 0, y1 = inside ? max_pk16(Fmt16<T>::pack2_1(vb[2], vb[3]), floor_a) : 0;
            auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
            *(i32x4*)(smem + kPrYOff + half * kPrYP + (rx & 1) * kPrYQ + (ry * kPrYC + (rx >> 1)) * 16) =
                i32x4{(int)r0[0], (int)r1[0], (int)r0[1], (int)r1[1]};                                                   // channels 8 half .. + 7
        };
        {
            f32x16 c[2];
            chain(wave, c[0]);
            static_for<NB - 1>([&](auto jc) {
                constexpr int j = decltype(jc)::value + 1;               // block j's MFMAs, then block j - 1's epilogue
                if (j < NB - 1 || wave + j * kPrNW < kPrYB) chain(wave + j * kPrNW, c[j & 1]);      // (wave-uniform)
                finish(wave + (j - 1) * kPrNW, c[(j - 1) & 1]);
            });
            if (wave + (NB - 1) * kPrNW < kPrYB) finish(wave + (NB - 1) * kPrNW, c[(NB - 1) & 1]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                     // the conv-A image is complete, the halo stage is free
        asm volatile("" ::: "memory");
        issue_halo(t + nwg);                                              // lands under phase 2 and the partner workgroup's phases
        // ---- phase 2: conv B (stride 2) from the conv-A image ------------------------------------------------------------------------
        const char* Y = smem + kPrYOff;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            f32x16 acc;
            static_for<9>([&](auto tc) {
                constexpr int tap = decltype(tc)::value;
                const i32x4 fb = *(const i32x4*)(Y + b_tap[tap] + r * 2 * kPrYC * 16);
                if constexpr (tap == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    Fmt16<T>::mfma32z(wb[0], fb, zero, acc);
                } else
                    Fmt16<T>::mfma32(wb[tap], fb, acc);
            });
            const int y = ty * kPrTH + wave * RW + r, x = tx * kPrTW + lr;
            const bool pin = y < pb.Ho && x < pb.Wo;
            const uint32_t obase = (uint32_t)(((b * pb.Ho + y) * pb.Wo + x) * pb.out_pix_stride);
            auto chan4 = [&](int g, float (&v)[4]) {
                const f32x4 sc = *(const f32x4*)(ss + 64 + 8 * g + 4 * half), sh = *(const f32x4*)(ss + 96 + 8 * g + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[4 * g + e] * sc[e] + sh[e];
            };
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                float va[4], vb[4];
                chan4(g, va);
                chan4(g + 1, vb);
                const int x0 = max_pk16(Fmt16<T>::pack2_1(va[0], va[1]), floor_b), x1 = max_pk16(Fmt16<T>::pack2_1(va[2], va[3]), floor_b);
                const int y0 = max_pk16(Fmt16<T>::pack2_1(vb[0], vb[1]), floor_b), y1 = max_pk16(Fmt16<T>::pack2_1(vb[2], vb[3]), floor_b);
                auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                i32x4 o = {(int)r0[0], (int)r1[0], (int)r0[1], (int)r1[1]};
                const int n = 8 * (g + half);
                const uint32_t off = ((obase + n) * 2u) | (pin && n < pb.Cout ? 0u : kOOB);
                __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, off, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead DMA must not land in a successor workgroup's LDS
}

bool pair_shape_ok(const ConvArgs& a, const ConvArgs& b) {
    return a.Cin == 16 && a.Cout == 16 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && !a.residual && !a.out_f32 &&
           b.Cin == 16 && b.Cout <= 32 && b.Cout % 16 == 0 && b.kh == 3 && b.kw == 3 && b.stride == 2 && b.pad == 1 && b.dil == 1 && !b.residual &&
           !b.out_f32 && b.B == a.B && b.H == a.Ho && b.W == a.Wo && a.Ho == a.H && a.Wo == a.W && a.in_pix_stride % 8 == 0 &&
           b.out_pix_stride % 8 == 0 && ((uintptr_t)b.out & 15) == 0 && (int64_t)b.M * b.out_pix_stride * 2 < 0x7ffffff0ll &&
           (!a.scale || ((uintptr_t)a.scale & 15) == 0) && (!a.shift || ((uintptr_t)a.shift & 15) == 0) &&
           (!b.scale || ((uintptr_t)b.scale & 15) == 0) && (!b.shift || ((uintptr_t)b.shift & 15) == 0);
}

int launch_pair(ConvArgs& a, ConvArgs& b, hipStream_t stream, int fmt) {
    const int num_cu = vd3d_device_cu_count();
    if (num_cu <= 0) return VD3D_ELAUNCH;
    const int ntiles = b.B * ((b.Ho + kPrTH - 1) / kPrTH) * ((b.Wo + kPrTW - 1) / kPrTW);
    const int grid = ntiles < 2 * num_cu ? ntiles : 2 * num_cu;          // two persistent workgroups per CU (76.5 KiB of LDS each)
    static Vd3dLdsLimit limh, limb;
    if (fmt == VD3D_F16) {
        if (const int rc = vd3d_raise_lds_limit((const void*)conv_pair_kernel<hf16>, kPrLds, limh, "hipFuncSetAttribute(conv_pair)")) return rc;
        hipLaunchKernelGGL(conv_pair_kernel<hf16>, dim3(grid), dim3(kPrNW * 64), kPrLds, stream, a, b, ntiles);
    } else {
        if (const int rc = vd3d_raise_lds_limit((const void*)conv_pair_kernel<short>, kPrLds, limb, "hipFuncSetAttribute(conv_pair)")) return rc;
        hipLaunchKernelGGL(conv_pair_kernel<short>, dim3(grid), dim3(kPrNW * 64), kPrLds, stream, a, b, ntiles);
    }
    return vd3d_check_launch("conv_pair");
}


bool narrow_shape_ok(const ConvArgs& a) {
    const bool out16 = !a.out_f32;
    return a.Cin % 64 == 0 && a.Cin >= 128 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.Cout <= 32 &&
           a.Cout % (out16 ? 16 : 4) == 0 && !a.residual && a.wfrag && a.in_pix_stride % 8 == 0 && a.out_pix_stride % (out16 ? 8 : 4) == 0 &&
           (!a.scale || ((uintptr_t)a.scale & 15) == 0) && (!a.shift || ((uintptr_t)a.shift & 15) == 0) && ((uintptr_t)a.out & 15) == 0 &&
           ((uintptr_t)a.wfrag & 15) == 0 && (int64_t)a.M * a.out_pix_stride * (out16 ? 2 : 4) < 0x7ffffff0ll && a.Ho == a.H && a.Wo == a.W;
}

template <typename T, bool OUTF32>
static int launch_narrow_t(ConvArgs& a, hipStream_t stream) {
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)conv_narrow_kernel<T, OUTF32>, kNwLds, lim, "hipFuncSetAttribute(conv_narrow)")) return rc;
    const int num_cu = vd3d_device_cu_count();
    if (num_cu <= 0) return VD3D_ELAUNCH;
    const int ntiles = a.B * ((a.Ho + kNwTH - 1) / kNwTH) * ((a.Wo + kNwTW - 1) / kNwTW);
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    hipLaunchKernelGGL((conv_narrow_kernel<T, OUTF32>), dim3(grid), dim3(512), kNwLds, stream, a, ntiles, a.Cin / 64);
    return vd3d_check_launch("conv_narrow");
}

int launch_narrow(ConvArgs& a, hipStream_t stream, int fmt) {
    if (fmt == VD3D_F16) return a.out_f32 ? launch_narrow_t<hf16, true>(a, stream) : launch_narrow_t<hf16, false>(a, stream);
    return a.out_f32 ? launch_narrow_t<short, true>(a, stream) : launch_narrow_t<short, false>(a, stream);
}


bool pw_shape_ok(const ConvArgs& a) {
    return (a.Cin == 64 || a.Cin == 128 || a.Cin == 256) && a.kh == 1 && a.kw == 1 && a.stride == 1 && a.pad == 0 && a.Ho == a.H && a.Wo == a.W &&
           a.Cout % 256 == 0 && a.Cout / 256 <= 16 && a.wide_store && !a.out_f32 && a.wfrag && a.in_pix_stride % 8 == 0 &&
           a.in_row_stride == a.W * a.in_pix_stride && a.in_batch_stride == (int64_t)a.H * a.in_row_stride && a.in_bytes < 0x7ffffff0u &&
           (int64_t)a.M * a.in_pix_stride * 2 < 0x7ffffff0ll && (int64_t)a.M * a.out_pix_stride * 2 < 0x7ffffff0ll &&
           (!a.residual || ((int64_t)a.M * a.res_pix_stride * 2 < 0x7ffffff0ll && a.res_pix_stride % 4 == 0 && ((uintptr_t)a.residual & 7) == 0));
}

template <typename T, int CIN>
static int launch_pw_t(ConvArgs& a, hipStream_t stream) {
    const int num_cu = vd3d_device_cu_count();
    if (num_cu <= 0) return VD3D_ELAUNCH;
    const int nslices = a.Cout / 256;
    const int nblk = (a.M + 31) / 32;
    // grid = 8 XCDs x block lanes x slices; CIN 64: <= 168 VGPRs (no spills) -> 3 workgroups per CU (4 at 128 VGPRs spill 13 and run the same 4.76 TB/s), CIN 128: 2, CIN 256: 1 (128 weight + 2 x 64 pixel registers)
    const int per_cu = CIN == 64 ? 3 : (CIN == 128 ? 2 : 1);
    int lanes = num_cu * per_cu / (8 * nslices);
    const int need = (nblk + 7) / 8;
    if (lanes > need) lanes = need;
    if (lanes < 1) lanes = 1;
    const int grid = 8 * lanes * nslices;
    if (a.residual) hipLaunchKernelGGL((conv_pw_kernel<T, CIN, true>), dim3(grid), dim3(256), 0, stream, a, nblk, nslices);
    else hipLaunchKernelGGL((conv_pw_kernel<T, CIN, false>), dim3(grid), dim3(256), 0, stream, a, nblk, nslices);
    return vd3d_check_launch("conv_pw");
}

int launch_pw(ConvArgs& a, hipStream_t stream, int fmt) {
    if (a.Cin == 64) return fmt == VD3D_F16 ? launch_pw_t<hf16, 64>(a, stream) : launch_pw_t<short, 64>(a, stream);
    if (a.Cin == 128) return fmt == VD3D_F16 ? launch_pw_t<hf16, 128>(a, stream) : launch_pw_t<short, 128>(a, stream);
    return fmt == VD3D_F16 ? launch_pw_t<hf16, 256>(a, stream) : launch_pw_t<short, 256>(a, stream);
}


bool ksplit_shape_ok(const ConvArgs& a) {
    return a.Cin == 256 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.Ho == a.H && a.Wo == a.W &&
           a.Cout % 64 == 0 && a.Cout / 64 <= 32 && a.wide_store && !a.out_f32 && a.wfrag &&
           (int64_t)a.M * a.out_pix_stride * 2 < 0x7ffffff0ll && (!a.residual || (int64_t)a.M * a.res_pix_stride * 2 < 0x7ffffff0ll);
}

#ifdef VD3D_TUNING
static int ksplit_abl() { static const int v = getenv("VD3D_X_KSPLIT_ABL") ? atoi(getenv("VD3D_X_KSPLIT_ABL")) : 0; return v; }
template <typename T, int ABL>
static int launch_ksplit_abl(ConvArgs& a, hipStream_t stream, int grid, int ntiles, int nslices) {
    static Vd3dLdsLimit lim;
    if (const int rc = vd3d_raise_lds_limit((const void*)conv_ksplit256_kernel<T, true, ABL>, kKsLds, lim, "hipFuncSetAttribute(conv_ksplit256 abl)")) return rc;
    hipLaunchKernelGGL((conv_ksplit256_kernel<T, true, ABL>), dim3(grid), dim3(512), kKsLds, stream, a, ntiles, nslices);
    return vd3d_check_launch("conv_ksplit256 abl");
}
#endif

template <typename T>
static int launch_ksplit_t(ConvArgs& a, hipStream_t stream) {
    static Vd3dLdsLimit lim_res, lim_nores;
    int rc = vd3d_raise_lds_limit((const void*)conv_ksplit256_kernel<T, true>, kKsLds, lim_res, "hipFuncSetAttribute(conv_ksplit256)");
    if (!rc) rc = vd3d_raise_lds_limit((const void*)conv_ksplit256_kernel<T, false>, kKsLds, lim_nores, "hipFuncSetAttribute(conv_ksplit256)");
    if (rc) return rc;
    const int num_cu = vd3d_device_cu_count();
    if (num_cu <= 0) return VD3D_ELAUNCH;
    const int nslices = a.Cout / 64;
    const int ntiles = a.B * ((a.H + kKsTH - 1) / kKsTH) * ((a.W + kKsTW - 1) / kKsTW);
    int lanes = num_cu / (8 * nslices);
    const int need = (ntiles + 7) / 8;
    if (lanes > need) lanes = need;
    if (lanes < 1) lanes = 1;
    const int grid = 8 * lanes * nslices;
#ifdef VD3D_TUNING
    if (a.residual && ksplit_abl() > 0) {       // timing ablations (the residual variant only)
        switch (ksplit_abl()) {
            case 1: return launch_ksplit_abl<T, 1>(a, stream, grid, ntiles, nslices);
            case 2: return launch_ksplit_abl<T, 2>(a, stream, grid, ntiles, nslices);
            case 3: return launch_ksplit_abl<T, 3>(a, stream, grid, ntiles, nslices);
            case 5: return launch_ksplit_abl<T, 5>(a, stream, grid, ntiles, nslices);
            case 6: return launch_ksplit_abl<T, 6>(a, stream, grid, ntiles, nslices);
            case 7: return launch_ksplit_abl<T, 7>(a, stream, grid, ntiles, nslices);
            case 8: return launch_ksplit_abl<T, 8>(a, stream, grid, ntiles, nslices);
            default: return launch_ksplit_abl<T, 4>(a, stream, grid, ntiles, nslices);
        }
    }
#endif
    if (a.residual) hipLaunchKernelGGL((conv_ksplit256_kernel<T, true>), dim3(grid), dim3(512), kKsLds, stream, a, ntiles, nslices);
    else hipLaunchKernelGGL((conv_ksplit256_kernel<T, false>), dim3(grid), dim3(512), kKsLds, stream, a, ntiles, nslices);
    return vd3d_check_launch("conv_ksplit256");
}

int launch_ksplit(ConvArgs& a, hipStream_t stream, int fmt) {
    return fmt == VD3D_F16 ? launch_ksplit_t<hf16>(a, stream) : launch_ksplit_t<short>(a, stream);
}

int launch_regw(ConvArgs& a, hipStream_t stream, int fmt, int ring, int abl) {
    if (ring == 4 && abl == 0) {
        if (fmt == VD3D_F16) return a.Cin == 128 ? launch_regw_t<hf16, 128>(a, stream) : launch_regw_t<hf16, 256>(a, stream);
        return a.Cin == 128 ? launch_regw_t<short, 128>(a, stream) : launch_regw_t<short, 256>(a, stream);
    }
#ifdef VD3D_TUNING
    if (a.Cin == 128 && fmt == VD3D_BF16) {
        if (ring == 8 && abl == 0) return launch_regw_t<short, 128, 8>(a, stream);
        if (ring == 2 && abl == 0) return launch_regw_t<short, 128, 2>(a, stream);
        if (ring == 4 && abl == 1) return launch_regw_t<short, 128, 4, 1>(a, stream);
        if (ring == 4 && abl == 2) return launch_regw_t<short, 128, 4, 2>(a, stream);
        if (ring == 4 && abl == 3) return launch_regw_t<short, 128, 4, 3>(a, stream);
        if (ring == 4 && abl == 9) return launch_regw_t<short, 128, 4, 9>(a, stream);
    }
#endif
    vd3d_set_error("conv_regw: this ring depth / ablation exists in the tuning build only (bf16, Cin 128)");
    return VD3D_EINVAL;
}

}  // namespace vd3d_conv

#ifdef VD3D_TUNING
extern "C" int vd3d_tuning_regw_stamps(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_regw_dbg), sizeof(unsigned long long) * (n < 64 ? n : 64)) == hipSuccess ? 0 : 1;
}
#endif
