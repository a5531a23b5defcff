// conv_bottleneck.hip -- a whole ResNet Bottleneck (backbones/resnet.py:55-91: conv1 1x1 + bn1 + relu -> conv2 3x3 + bn2 + relu -> conv3 1x1 + bn3,
// + identity | downsample(x), relu) in ONE launch for the 64-wide stage (ResNet-50/101/152 layer1: 256 | 64 -> 64 -> 64 -> 256 at 1/4 resolution), 16-bit formats.
//
// Launched separately the block moves ~3 GB at BASELINE config 3's size (64 x 72 x 320): it writes and re-reads both 64-channel intermediates and reads the
// 256-channel input twice (conv1 operand, conv3 residual) -- every one of its launches runs at the bandwidth of a mixed read / write stream (4.6 - 5.2 TB/s,
// profiles/r05_c3_conv_layers.txt).  Here x is read once (+ the 3x3 halo), the output written once, and both intermediates live in LDS only:
//   * a workgroup = 4 waves (one per SIMD), two workgroups per CU, persistent over 8 x 16-pixel tiles (XCD-aware walk);
//   * ALL weights live in registers as MFMA A operands (rows = output channels).  A wave owns an N slice of every conv: 16 of the 64 mid channels
//     of conv1 and conv2, 64 of the 256 output channels of conv3 (and of the downsample conv): 32 + 72 + 32 (+ 32) registers;
//   * conv1 runs on the tile's 10 x 18 halo (recomputed where tiles overlap: 1.41 x of the cheapest conv), streaming x through a two-stage ring of
//     64-channel halo images filled by LDS-DMA (the resident64 image: one 128-byte row per pixel, 16-byte slots XOR-swizzled by row & 7 -- every
//     ds_read_b128 of a 16-pixel MFMA B operand is conflict free for the aligned blocks of conv1 / conv3 AND for all nine tap shifts of conv2);
//   * conv1's result (bn1, relu, rounded to the storage type, ZERO outside the image: conv2's padding) is written to LDS in the same image layout,
//     conv2 reads its nine taps from there, its result overwrites it, conv3 reads that; the residual (identity blocks) is re-read from global memory
//     (the rows were staged moments ago: an L2 / MALL hit), the downsample conv of a stage's first block reads the staged x image itself;
//   * rounding points = the unfused launches' (each intermediate rounded once to the storage type, the downsample branch rounded before the add).
#include "conv_common.h"

using namespace vd3d_conv;

namespace {

constexpr int kBnTH = 8, kBnTW = 16, kBnHW = kBnTW + 2, kBnHR = (kBnTH + 2) * kBnHW;   // 18, 180 halo pixels
constexpr int kBnRows = 192;                       // halo rows padded to 12 pixel blocks of 16
constexpr int kBnImg = kBnRows * 128;              // 24 KiB: one 64-channel image
constexpr int kRing = 4;                          // B-fragment ring: ds_read_b128 issued this many MFMAs ahead
constexpr int kBnPW = 6;                           // 1 KiB DMA pieces per wave and image (24 pieces; rows >= 180 read out of range: zeros)
constexpr int kBnT1 = 2 * kBnImg;                  // conv1 / conv2 results
constexpr int kBnSS = 3 * kBnImg;                  // fp32: scale1[64] shift1[64] scale2[64] shift2[64] | scale3[256] shift3[256] scaleD[256] shiftD[256]
constexpr int kBnLds = kBnSS + (256 + 4 * 256) * 4;    // 78 848 B: two workgroups per CU

struct BnWalk { int first, end, stride; };
VD3D_DEV BnWalk bn_tile_walk(int ntiles, int plain_walk) {      // (conv_resident.hip xcd_tile_walk)
    BnWalk w = {(int)blockIdx.x, ntiles, (int)gridDim.x};
    if ((gridDim.x & 7) == 0 && !plain_walk) {
        const int xcd = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        const int start = xcd * q + (xcd < r ? xcd : r);
        w.first = start + (int)(blockIdx.x >> 3);
        w.end = start + q + (xcd < r ? 1 : 0);
        w.stride = (int)(gridDim.x >> 3);
    }
    return w;
}

template <int V> struct BnC { static constexpr int value = V; };
template <int... Is, class F> VD3D_DEV void bn_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(BnC<Is>{}), ...); }
template <int N, class F> VD3D_DEV void bn_for(F&& f) { bn_for_impl(f, std::make_integer_sequence<int, N>{}); }

// c1 / c2 / c3 / cd: the parameter blocks of conv1, conv2, conv3 and (DS) the downsample conv.  CIN = 256: identity block, residual = c1.in;
// CIN = 64 + DS: the stage's first block.
template <typename T, int CIN, bool DS>
__global__ void __launch_bounds__(256, 2) bottleneck64_kernel(const ConvArgs c1, const ConvArgs c2, const ConvArgs c3, const ConvArgs cd, int ntiles, int plain_walk) {
    static_assert((CIN == 256 && !DS) || (CIN == 64 && DS), "identity block (256 in) or first block (64 in + downsample)");
    constexpr int NCH = CIN / 64;                  // 64-channel chunks of x
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    const int H = c1.H, W = c1.W;
    const int tiles_x = (W + kBnTW - 1) / kBnTW, tiles_y = (H + kBnTH - 1) / kBnTH, tiles_img = tiles_x * tiles_y;

    // ---- weights -> registers (MFMA A operand of the 16x16x32 form: lane (m = lane & 15, kg = lane >> 4) holds 8 k values of output row m) -------------
    // conv1 / conv2: row m of the wave's block = mid channel 16 wave + m.  conv3 / downsample: row m = 4 g' + r' of block cb = output channel
    // 64 wave + 32 (cb >> 1) + 8 g' + 4 (cb & 1) + r': lane (px, g) then holds channels 64 wave + 8 g + [0, 8) (blocks 0, 1) and + 32 (blocks 2, 3) of its pixel,
    // i.e. two 16-byte runs whose store instructions each cover 64 contiguous bytes per pixel
    i32x4 w1[NCH * 2], w2[18], w3[8], wd[DS ? 8 : 1];
    {
        const char* r1 = c1.weight + ((size_t)(16 * wave + px) * c1.Kpad + g * 8) * 2;
        bn_for<NCH * 2>([&](auto ic) { constexpr int i = decltype(ic)::value; w1[i] = *(const i32x4*)(r1 + i * 64); });
        const char* r2 = c2.weight + ((size_t)(16 * wave + px) * c2.Kpad + g * 8) * 2;
        bn_for<18>([&](auto ic) { constexpr int i = decltype(ic)::value; w2[i] = *(const i32x4*)(r2 + i * 64); });
        bn_for<8>([&](auto ic) {
            constexpr int i = decltype(ic)::value, cb = i >> 1, ks = i & 1;
            const int ch = 64 * wave + 32 * (cb >> 1) + 8 * (px >> 2) + 4 * (cb & 1) + (px & 3);
            w3[i] = *(const i32x4*)(c3.weight + ((size_t)ch * c3.Kpad + ks * 32 + g * 8) * 2);
            if constexpr (DS) wd[i] = *(const i32x4*)(cd.weight + ((size_t)ch * cd.Kpad + ks * 32 + g * 8) * 2);
        });
    }
    // folded BN of every conv in LDS (fp32): a lane reads the four values of its channels in the epilogues (registers are the scarce resource here)
    float* ss12 = (float*)(smem + kBnSS);
    float* ss = ss12 + 256;
    {
        const int q = tid >> 6, ch = tid & 63;
        const float* src = q == 0 ? c1.scale : (q == 1 ? c1.shift : (q == 2 ? c2.scale : c2.shift));
        ss12[tid] = src ? src[ch] : ((q & 1) ? 0.f : 1.f);
    }
    ss[tid] = c3.scale ? c3.scale[tid] : 1.f;
    ss[256 + tid] = c3.shift ? c3.shift[tid] : 0.f;
    if constexpr (DS) {
        ss[512 + tid] = cd.scale ? cd.scale[tid] : 1.f;
        ss[768 + tid] = cd.shift ? cd.shift[tid] : 0.f;
    }

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)c1.in, 0, c1.in_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const BnWalk walk = bn_tile_walk(ntiles, plain_walk);
    const int t_end = walk.end, nwg = walk.stride;

    // one 64-channel halo image (chunk `c` of tile `t`) -> stage `s`: 6 unconditional DMA instructions per wave
    auto issue_image = [&](int t, int c, int s) {
        const bool tv = t < t_end;
        const int tt = tv ? t : 0;
        const int b = tt / tiles_img, trem = tt - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
#pragma unroll
        for (int it = 0; it < kBnPW; ++it) {
            const int piece = wave + 4 * it;
            const int hr = 8 * piece + (lane >> 3);
            const int hy = (hr * 3641) >> 16, hx = hr - hy * kBnHW;           // hr / 18 for hr < 192
            const int iy = ty * kBnTH - 1 + hy, ix = tx * kBnTW - 1 + hx;
            const bool v = tv && hr < kBnHR && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int slot = (lane & 7) ^ (hr & 7);                         // source-side swizzle: LDS slot position q holds channel slot q ^ (row & 7)
            const uint32_t off = v ? (uint32_t)((int)(b * c1.in_batch_stride) + iy * c1.in_row_stride + ix * c1.in_pix_stride + c * 64 + slot * 8) * 2 : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr_t)(smem + s * kBnImg + piece * 1024), 16, off, 0, 0, 0);
        }
    };
    // `pxv` / `gv`: the lane's pixel / k group re-defined opaquely at the top of every phase -- the ~60 distinct LDS addresses of a tile are loop
    // invariants, which the compiler otherwise hoists out of the tile loop and keeps in registers for the whole kernel (it then spills ~200 of them)
    int pxv = px, gv = g;
#define BN_OPAQUE() asm volatile("" : "+v"(pxv), "+v"(gv))
    // B operand (16 pixels x 32 k) of image row `row`, k step ks: 16 bytes at slot (4 ks + g) ^ (row & 7)
    auto frag = [&](int base, int row, int ks) { return *(const i32x4*)(smem + base + row * 128 + ((((ks << 2) | gv) ^ (row & 7)) << 4)); };
    // where lane (px, g) of the wave writes its four mid channels of image row `row`: slot 2 wave + (g >> 1), half g & 1
    auto t_addr = [&](int row) { return kBnT1 + row * 128 + ((((wave << 1) | (gv >> 1)) ^ (row & 7)) << 4) + ((gv & 1) << 3); };

    int t = walk.first;
    // prologue: the first tile's first images
    if constexpr (NCH == 4) { issue_image(t, 0, 0); issue_image(t, 1, 1); }
    else issue_image(t, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int par = 0;                                    // (CIN = 64) the stage holding this tile's image
    for (; t < t_end; t += nwg) {
        const int b = t / tiles_img, trem = t - b * tiles_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        __builtin_amdgcn_s_barrier();               // this tile's landed images are visible to every wave; the previous tile's LDS reads are done
        asm volatile("" ::: "memory");
        // ---- conv1 on the halo: 12 pixel blocks x (NCH x 2) k steps ------------------------------------------------------------------------
        BN_OPAQUE();
        f32x4 a1[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) a1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NCH == 4) {
            bn_for<4>([&](auto ic) {
                constexpr int c = decltype(ic)::value, s = c & 1;
                if constexpr (c >= 2) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kBnPW) : "memory");     // chunk c has landed (the younger kBnPW are chunk c + 1 / the next tile's chunk 0)
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                // 24 MFMAs, each fed by ONE ds_read_b128 issued kRing MFMAs ahead (pinned 1 : 1: left alone the scheduler issues every read of the
                // phase up front and spills the fragments)
                i32x4 ring[kRing];
                bn_for<kRing>([&](auto jf) { constexpr int f = decltype(jf)::value; ring[f] = frag(s * kBnImg, 16 * (f >> 1) + pxv, f & 1); });
                bn_for<24>([&](auto jf) {
                    constexpr int f = decltype(jf)::value;
                    Fmt16<T>::mfma16(w1[2 * c + (f & 1)], ring[f % kRing], a1[f >> 1]);
                    if constexpr (f + kRing < 24) ring[f % kRing] = frag(s * kBnImg, 16 * ((f + kRing) >> 1) + pxv, (f + kRing) & 1);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();        // every wave is done with stage s
                asm volatile("" ::: "memory");
                if constexpr (c < 2) issue_image(t, c + 2, s);
                else issue_image(t + nwg, c - 2, s);                                  // the next tile's chunks 0 / 1
            });
        } else {
            issue_image(t + nwg, 0, par ^ 1);        // the next tile's image into the other stage (last read by the previous tile: behind the top barrier)
            i32x4 ring[kRing];
            bn_for<kRing>([&](auto jf) { constexpr int f = decltype(jf)::value; ring[f] = frag(par * kBnImg, 16 * (f >> 1) + pxv, f & 1); });
            bn_for<24>([&](auto jf) {
                constexpr int f = decltype(jf)::value;
                Fmt16<T>::mfma16(w1[f & 1], ring[f % kRing], a1[f >> 1]);
                if constexpr (f + kRing < 24) ring[f % kRing] = frag(par * kBnImg, 16 * ((f + kRing) >> 1) + pxv, (f + kRing) & 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            });
        }
        // bn1 + relu -> storage type -> t1 (zero outside the image: conv2 pads its INPUT with zeros)
        BN_OPAQUE();
        const f32x4 s1 = *(const f32x4*)(ss12 + 16 * wave + 4 * gv), h1 = *(const f32x4*)(ss12 + 64 + 16 * wave + 4 * gv);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int row = 16 * j + pxv;
            const int hy = (row * 3641) >> 16, hx = row - hy * kBnHW;
            const int iy = ty * kBnTH - 1 + hy, ix = tx * kBnTW - 1 + hx;
            const bool v = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = a1[j][r] * s1[r] + h1[r];
            // round (one packed conversion per pair), ReLU on the packed pairs, zero outside the image: 8 instructions for 4 values
            const int q0 = max_pk16(Fmt16<T>::pack2_1(o[0], o[1]), 0), q1 = max_pk16(Fmt16<T>::pack2_1(o[2], o[3]), 0);
            *(i32x2*)(smem + t_addr(row)) = i32x2{v ? q0 : 0, v ? q1 : 0};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- conv2: 8 pixel blocks (tile rows) x 9 taps x 2 k steps from t1 ---------------------------------------------------------------
        BN_OPAQUE();
        f32x4 a2[8];
#pragma unroll
        for (int y = 0; y < 8; ++y) a2[y] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            // fragment f = (tap * 8 + y) * 2 + ks: 144 MFMAs, the same 1 : 1 pipeline
            auto ld2 = [&](auto jf) {
                constexpr int f = decltype(jf)::value, tap = f >> 4, y = (f >> 1) & 7, ks = f & 1, dy = tap / 3, dx = tap % 3;
                return frag(kBnT1, (y + dy) * kBnHW + dx + pxv, ks);
            };
            i32x4 ring[kRing];
            bn_for<kRing>([&](auto jf) { ring[decltype(jf)::value] = ld2(jf); });
            bn_for<144>([&](auto jf) {
                constexpr int f = decltype(jf)::value;
                Fmt16<T>::mfma16(w2[2 * (f >> 4) + (f & 1)], ring[f % kRing], a2[(f >> 1) & 7]);
                if constexpr (f + kRing < 144) ring[f % kRing] = ld2(BnC<f + kRing>{});
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // every wave is done reading t1
        asm volatile("" ::: "memory");
        BN_OPAQUE();
        const f32x4 s2 = *(const f32x4*)(ss12 + 128 + 16 * wave + 4 * gv), h2 = *(const f32x4*)(ss12 + 192 + 16 * wave + 4 * gv);
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = a2[y][r] * s2[r] + h2[r];
            *(i32x2*)(smem + t_addr(16 * y + pxv)) = i32x2{max_pk16(Fmt16<T>::pack2_1(o[0], o[1]), 0), max_pk16(Fmt16<T>::pack2_1(o[2], o[3]), 0)};
        }
        // the next tile's images (issued before conv2) have had conv2's time to land: waited for HERE, while no store is in flight yet
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- conv3 (+ downsample conv) per pixel block, epilogue: bn3 + residual + relu -> 2 x 16-byte stores per lane ----------------------
        // operands of row y + 1 (t2 fragments, residual / x fragments) are requested before row y's epilogue
        BN_OPAQUE();
        const int64_t m00 = ((int64_t)b * H + ty * kBnTH) * W + tx * kBnTW + pxv;       // the lane's pixel of tile row 0
        const bool xin = tx * kBnTW + pxv < W;
        i32x4 bq[2][2], xq[2][2];
        auto ld3 = [&](int y, i32x4* b2, i32x4* x2) {
            b2[0] = frag(kBnT1, 16 * y + pxv, 0);
            b2[1] = frag(kBnT1, 16 * y + pxv, 1);
            if constexpr (DS) {
                const int row = (y + 1) * kBnHW + 1 + pxv;
                x2[0] = frag(par * kBnImg, row, 0);
                x2[1] = frag(par * kBnImg, row, 1);
            } else {
                const bool pin = xin && ty * kBnTH + y < H;
                const char* rp = c3.residual + ((pin ? m00 + (int64_t)y * W : 0) * c3.res_pix_stride + 64 * wave + 8 * gv) * 2;
                x2[0] = *(const i32x4*)rp;
                x2[1] = *(const i32x4*)(rp + 64);
            }
        };
        auto row3 = [&](int y, const i32x4* b2, const i32x4* x2) {
            const bool pin = xin && ty * kBnTH + y < H;
            f32x4 a3[4], ad[4];
            auto mm = [&](int cb) {
                a3[cb] = ad[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                Fmt16<T>::mfma16(w3[2 * cb], b2[0], a3[cb]);
                Fmt16<T>::mfma16(w3[2 * cb + 1], b2[1], a3[cb]);
                if constexpr (DS) {
                    Fmt16<T>::mfma16(wd[2 * cb], x2[0], ad[cb]);
                    Fmt16<T>::mfma16(wd[2 * cb + 1], x2[1], ad[cb]);
                }
            };
            if constexpr (!DS) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) mm(cb);
            }
            i32x4 o2[2];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                if constexpr (DS) mm(cb);          // (two accumulator sets: one channel block at a time keeps the first-block variant inside its 256 registers)
                const int ch = 64 * wave + 32 * (cb >> 1) + 8 * gv + 4 * (cb & 1);
                const f32x4 sc = *(const f32x4*)(ss + ch), sh = *(const f32x4*)(ss + 256 + ch);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = a3[cb][r] * sc[r] + sh[r];
                if constexpr (DS) {
                    const f32x4 dsc = *(const f32x4*)(ss + 512 + ch), dsh = *(const f32x4*)(ss + 768 + ch);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += Fmt16<T>::tof(Fmt16<T>::one(ad[cb][r] * dsc[r] + dsh[r]));     // the branch is a stored tensor in the unfused path: rounded once
                } else {
                    const uint32_t r0 = (uint32_t)(int)x2[cb >> 1][2 * (cb & 1)], r1 = (uint32_t)(int)x2[cb >> 1][2 * (cb & 1) + 1];
                    v[0] += Fmt16<T>::lo(r0);
                    v[1] += Fmt16<T>::hi(r0);
                    v[2] += Fmt16<T>::lo(r1);
                    v[3] += Fmt16<T>::hi(r1);
                }
                // ReLU on the packed pair (one v_pk_max_i16 per two values; relu(round(x)) == round(relu(x)))
                const int floor16 = c3.relu ? 0 : (int)0x80008000u;
                o2[cb >> 1][2 * (cb & 1)] = max_pk16(Fmt16<T>::pack2_1(v[0], v[1]), floor16);
                o2[cb >> 1][2 * (cb & 1) + 1] = max_pk16(Fmt16<T>::pack2_1(v[2], v[3]), floor16);
            }
            if (pin) {
                char* op = c3.out + ((m00 + (int64_t)y * W) * c3.out_pix_stride + 64 * wave + 8 * gv) * 2;
                *(i32x4*)op = o2[0];
                *(i32x4*)(op + 64) = o2[1];
            }
        };
        if constexpr (!DS) {
            ld3(0, bq[0], xq[0]);
#pragma unroll 1
            for (int y = 0; y < 8; y += 2) {
                ld3(y + 1, bq[1], xq[1]);
                row3(y, bq[0], xq[0]);
                if (y + 2 < 8) ld3(y + 2, bq[0], xq[0]);
                row3(y + 1, bq[1], xq[1]);
            }
        } else {
            // (the first-block variant holds 144 weight registers: no look-ahead, one row at a time)
#pragma unroll 1
            for (int y = 0; y < 8; ++y) {
                ld3(y, bq[0], xq[0]);
                row3(y, bq[0], xq[0]);
            }
        }
        par ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead DMAs must not land in a successor workgroup's LDS
}

bool conv_is(const ConvArgs& a, int k, int cin, int cout) {
    return a.kh == k && a.kw == k && a.stride == 1 && a.pad == k / 2 && a.dil == 1 && a.Cin == cin && a.Cout == cout && !a.out_f32 && a.Ho == a.H && a.Wo == a.W;
}

}  // namespace

namespace vd3d_conv {

bool bottleneck_shape_ok(const ConvArgs& c1, const ConvArgs& c2, const ConvArgs& c3, const ConvArgs* cd) {
    const int cin = cd ? 64 : 256;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!conv_is(c1, 1, cin, 64) || !conv_is(c2, 3, 64, 64) || !conv_is(c3, 1, 64, 256)) return false;
    if (c1.residual || c2.residual || !c1.relu || !c2.relu) return false;
    if (c2.B != c1.B || c2.H != c1.H || c2.W != c1.W || c3.B != c1.B || c3.H != c1.H || c3.W != c1.W) return false;
    if (c1.in_pix_stride % 8 || c1.in_row_stride % 8 || c1.in_batch_stride % 8 || !al16(c1.in) || !al16(c3.out) || c3.out_pix_stride % 8) return false;
    if ((int64_t)c3.M * c3.out_pix_stride * 2 >= 0x7ffffff0ll) return false;
    if (cd) {
        if (!conv_is(*cd, 1, 64, 256) || cd->residual || cd->relu || c3.residual || cd->in != c1.in) return false;
    } else {
        if (c3.residual != c1.in || c3.res_pix_stride != c1.in_pix_stride || c1.in_row_stride != c1.W * c1.in_pix_stride ||
            c1.in_batch_stride != (int64_t)c1.H * c1.W * c1.in_pix_stride) return false;      // the residual IS the block's input, densely laid out
    }
    return true;
}

int launch_bottleneck(ConvArgs& c1, ConvArgs& c2, ConvArgs& c3, ConvArgs* cd, hipStream_t stream, int fmt) {
    const int num_cu = vd3d_device_cu_count();
    if (num_cu <= 0) return VD3D_ELAUNCH;
    const int ntiles = c1.B * ((c1.H + kBnTH - 1) / kBnTH) * ((c1.W + kBnTW - 1) / kBnTW);
    const int grid = ntiles < 2 * num_cu ? ntiles : 2 * num_cu;          // two persistent workgroups per CU
    const int plain = vd3d_switch(VD3D_SW_PLAIN_TILE_WALK) ? 1 : 0;
    static Vd3dLdsLimit lim[4];
#define VD3D_BN_LAUNCH(T, CIN, DS, slot, CD)                                                                                                     \
    do {                                                                                                                                         \
        if (const int rc = vd3d_raise_lds_limit((const void*)bottleneck64_kernel<T, CIN, DS>, kBnLds, lim[slot], "hipFuncSetAttribute(bottleneck64)")) return rc; \
        hipLaunchKernelGGL((bottleneck64_kernel<T, CIN, DS>), dim3(grid), dim3(256), kBnLds, stream, c1, c2, c3, CD, ntiles, plain);             \
    } while (0)
    if (cd) {
        if (fmt == VD3D_F16) VD3D_BN_LAUNCH(hf16, 64, true, 0, *cd);
        else VD3D_BN_LAUNCH(short, 64, true, 1, *cd);
    } else {
        if (fmt == VD3D_F16) VD3D_BN_LAUNCH(hf16, 256, false, 2, c3);
        else VD3D_BN_LAUNCH(short, 256, false, 3, c3);
    }
#undef VD3D_BN_LAUNCH
    return vd3d_check_launch("conv_bottleneck");
}

}  // namespace vd3d_conv
