"""GPU: EVERY production tile of the implicit-GEMM conv family against the CPU oracle (torch fp32 convolution on operands
rounded exactly like the HIP path rounds them), not against another tile of the same family.

  * `vd3d_conv2d_production_tiles` lists the tile ids the heuristic in csrc/conv_igemm.hip can select; each one is FORCED
    (`vd3d_test_force_conv_tile`) on small shapes that are deliberately awkward for it: M not a tile multiple, Cout not a
    tile multiple, with / without residual, channel-slice input and output, fp32 output from bf16 compute.
  * the tiles the bench (BASELINE config 2, batch 8) actually runs are then checked under NATURAL dispatch at the
    bench's own layer shapes (M = 8 x 24 x 80 = 15360 pixels: 1408->1408, 1152->1152, 1408->576, 1408->256; layer1/2/3 of
    the stacked L/R batch; the R50 head 2176->2176 at 16 x 18 x 80), with `VD3D_CONV_DEBUG`-free proof of which kernel ran
    left to the rocprof traces in profiles/.

Bars: bf16 -- every element within ONE bf16 ulp of the oracle (|d| <= 2^-7 |ref|) plus the fp32 summation-order noise of a
K-long dot product (3e-5 of the output scale); fp32 -- 2e-5 of the output scale (v_mfma_f32_32x32x2_f32 chains)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16_TILES = {50, 54, 52, 53, 76, 79, 73, 61, 68, 69, 58, 70}    # (16-bit-only tiles: bf16 and fp16, no fp32 variant)       # 16x16x32 bf16 MFMA tiles; 61 = register-resident weights (bf16, Cin 128 | 256)
HALO_TILES = {21, 23, 27}                # 3x3 / s1 / p1, Cin % K-slice == 0
NARROW = {87: 32, 30: 64, 130: 64}       # tiles whose N extent bounds Cout in production (130: the split-K form of the 128 x 64 tile)


# = vd3d_conv2d_production_tiles() (tests/test_abi.py checks the two lists agree, on CPU)
PRODUCTION_TILES = [44, 42, 40, 41, 50, 54, 52, 53, 12, 11, 76, 43, 79, 73, 87, 30, 21, 23, 27, 61, 68, 69, 58, 70, 144, 130]   # 144 / 130: split-K (two launches)


class forced_tile:
    def __init__(self, cfg):
        self.cfg = cfg

    def __enter__(self):
        from visualdet3d_amd import _lib
        _lib.lib().vd3d_test_force_conv_tile(self.cfg)

    def __exit__(self, *exc):
        from visualdet3d_amd import _lib
        _lib.lib().vd3d_test_force_conv_tile(0)


def run_case(B, H, W, Cin, Cout, k=3, stride=1, pad=1, dil=1, residual=False, relu=True, dtype=torch.bfloat16, bn=True,
             in_extra=0, out_extra=0, out_f32=False, seed=0, cfg=0):
    """-> (max ulp-normalised error, max error relative to the output scale)."""
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    rnd = (lambda t: t.to(dtype).float()) if dtype in (torch.bfloat16, torch.float16) else (lambda t: t)
    y = F.conv2d(rnd(x), rnd(w), None, stride, pad, dil)
    bnp = None
    scale, shift = torch.ones(Cout), b.clone()
    if bn:
        bnp = (torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1, torch.randn(Cout, generator=g) * 0.1,
               torch.rand(Cout, generator=g) + 0.5, 1e-5)
        s = bnp[0] / torch.sqrt(bnp[3] + bnp[4])
        shift = shift * s + (bnp[1] - bnp[2] * s)
        scale = s
    y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    Ho, Wo = y.shape[2:]
    res = None
    if residual:
        res = rnd(torch.randn(B, Cout, Ho, Wo, generator=g))
        y = y + res
    if relu:
        y = F.relu(y)
    dev = 'cuda'
    xin = torch.zeros(B, H, W, Cin + in_extra, dtype=dtype, device=dev)
    xin[..., in_extra:] = x.permute(0, 2, 3, 1).to(dev).to(dtype)
    pc = ops.pack_conv(w.to(dev), b.to(dev), tuple(t.to(dev) if torch.is_tensor(t) else t for t in bnp) if bn else None, dtype, stride, pad, dil)
    odt = torch.float32 if out_f32 else dtype
    obuf = torch.full((B, Ho, Wo, Cout + out_extra), 7.0, dtype=odt, device=dev)
    rv = res.permute(0, 2, 3, 1).contiguous().to(dev).to(dtype) if residual else None
    with forced_tile(cfg):
        out = ops.conv2d(xin[..., in_extra:], pc, out=obuf[..., :Cout], residual=rv, relu=relu, out_f32=out_f32)
        torch.cuda.synchronize()
    if out_extra:
        assert bool((obuf[..., Cout:] == 7.0).all()), 'kernel wrote outside its channel slice'
    got = out.float().cpu().permute(0, 3, 1, 2)
    sc = y.abs().max().item()
    d = (got - y).abs()
    rel = d.max().item() / sc
    if dtype == torch.bfloat16 and not out_f32:
        ulp = (d / (y.abs() * 2.0 ** -7 + 3e-5 * sc)).max().item()      # <= 1: within one bf16 ulp (+ summation noise)
    elif dtype == torch.float16 and not out_f32:
        ulp = (d / (y.abs() * 2.0 ** -10 + 3e-5 * sc)).max().item()     # <= 1: within one fp16 ulp (+ summation noise)
    elif dtype in (torch.bfloat16, torch.float16):
        ulp = rel / 1e-4                                                  # fp32 epilogue from bf16 operands: summation noise only
    else:
        ulp = rel / 2e-5
    return ulp, rel


def _shapes_for(cfg):
    """Small shapes that are awkward for tile `cfg` (B, H, W, Cin, Cout, kwargs)."""
    if cfg == 68:       # K-split resident weights: Cin 256, 64-channel slices, 8 x 8-pixel tiles, rotating epilogue owner
        return [
            (1, 11, 21, 256, 64, dict(residual=True)),                  # ragged tile grid, one slice
            (2, 8, 8, 256, 128, dict(residual=False, relu=False)),      # one tile per image, two slices
            (1, 17, 26, 256, 192, dict(residual=True, in_extra=64, out_extra=64)),
            (16, 24, 80, 256, 256, dict(residual=True)),                # layer3 at the bench shape: 4 slices, 7.5 tiles per workgroup
            (3, 40, 72, 256, 256, dict(residual=False, bn=False)),
        ]
    if cfg == 58:       # point-wise expansion streaming kernel: 1x1 / s1, Cin 64 | 128, 256-channel slices, 32-pixel blocks, no LDS
        return [
            (1, 7, 13, 64, 256, dict(k=1, pad=0, residual=True)),                        # M = 91: ragged last block, fewer blocks than waves
            (2, 9, 31, 128, 512, dict(k=1, pad=0, residual=True, bn=False)),             # two slices, two K chunks
            (1, 16, 32, 64, 512, dict(k=1, pad=0, residual=False, relu=False)),          # no residual (down-sample branch)
            (1, 10, 18, 128, 256, dict(k=1, pad=0, residual=True, in_extra=64, out_extra=64)),   # channel-slice views
            (8, 36, 160, 64, 256, dict(k=1, pad=0, residual=True)),                      # many blocks per workgroup (steady-state prefetch path)
            (4, 18, 80, 128, 1024, dict(k=1, pad=0, residual=False)),                    # four slices
            (2, 18, 80, 256, 1024, dict(k=1, pad=0, residual=True)),                     # Cin 256 (one wave per SIMD), four slices
            (1, 5, 9, 256, 256, dict(k=1, pad=0, residual=False, relu=False)),
        ]
    if cfg == 69:       # small-channel streaming kernel: Cin 16 | 32 | 64, Cout <= 32, stride 1 | 2, 16-bit or fp32 output, no residual
        return [
            (1, 13, 45, 16, 16, dict()),                                      # ragged tile grid
            (2, 24, 70, 16, 32, dict(stride=2)),                              # stride 2: 12 x 35 outputs
            (1, 17, 33, 32, 32, dict(relu=False)),
            (2, 18, 66, 32, 16, dict(stride=2, bn=False)),
            (1, 21, 67, 64, 32, dict(out_f32=True, bn=False, relu=False)),    # DCN offset conv (27 channels padded to 32), fp32 out
            (1, 9, 40, 64, 28, dict(out_f32=True, relu=False)),               # Cout % 4 == 0 < 32: masked stores
            (3, 64, 96, 16, 16, dict(in_extra=16, out_extra=16)),             # channel-slice views, many tiles per workgroup
            (2, 32, 64, 64, 16, dict()),
            # MORE tiles than workgroup slots (512): every workgroup walks 2 - 3 tiles, the steady state of the halo ring AND of the
            # fragment ring across the tile boundary (round 2 had no such case for Cin 16 | 32 and shipped a ring that rotated there)
            (2, 256, 512, 16, 16, dict()),                                   # 1024 tiles, DLA level 0 shape family
            (3, 512, 1024, 16, 32, dict(stride=2)),                          # 1536 tiles, DLA level 1 (stride 2)
            (2, 320, 512, 32, 32, dict(relu=False)),                          # 1280 tiles, Cin 32 (NF = 18)
            (5, 128, 440, 64, 32, dict(out_f32=True, bn=False, relu=False)),  # 1120 tiles, DCN offset conv at the KM3D s4 shape
        ]
    if cfg == 70:       # narrow-output streaming kernel: 3x3 / s1 / p1, Cin % 64 == 0 (>= 128), Cout = 32 (w_frag needs 32 filters)
        return [
            (1, 13, 45, 128, 32, dict(out_f32=True, bn=False, relu=False)),   # ragged tile grid, two chunks; DCN offset conv form
            (2, 16, 55, 512, 32, dict(out_f32=True, bn=False, relu=False)),   # DLA level 5 shape: eight chunks, one tile per workgroup
            (1, 9, 40, 192, 32, dict()),                                      # 16-bit output, BN + ReLU epilogue, three chunks
            (1, 17, 33, 256, 32, dict(relu=False, in_extra=64, out_extra=32)),  # channel-slice views
            (2, 18, 80, 2176, 32, dict(out_f32=True, bn=False, relu=False)),  # stereo base head: 34 chunks
            # MORE tiles than workgroup slots (256): every workgroup walks 4 - 5 tiles x chunks, the carried accumulator is re-zeroed
            (5, 64, 220, 128, 32, dict(out_f32=True, bn=False, relu=False)),  # 1120 tiles, KM3D s8 offset conv
            (9, 32, 110, 256, 32, dict()),                                    # 576 tiles, 16-bit output
        ]
    if cfg == 61:       # register-resident weights: Cin 128 (128-channel slices) | 256 (64-channel slices, K halves added in LDS)
        return [
            (1, 11, 37, 128, 128, dict(residual=True)),                 # ragged tile grid, fewer tiles than CUs
            (2, 8, 16, 256, 64, dict(residual=False, relu=False)),      # one tile per image, one slice
            (1, 17, 50, 256, 192, dict(residual=True, in_extra=64, out_extra=64)),    # 3 slices, channel-slice views
            (3, 40, 72, 128, 256, dict(residual=True, bn=False)),       # 2 slices, several tiles per workgroup lane
            (16, 24, 80, 256, 256, dict(residual=True)),                # layer3 at the bench shape: 4 slices, 3.75 rounds
            (2, 48, 160, 128, 128, dict(residual=False)),               # many tiles per workgroup (steady-state vmcnt path)
        ]
    if cfg == 144:      # split-K over 128 x 128 tiles (phase 1: tiles x splits workgroups park fp32 partials, phase 2: add + epilogue)
        return [
            (1, 13, 27, 64, 400, dict(residual=True)),                      # 9 slices -> 2 splits of 5 + 4; ragged M, partial last N tile
            (2, 7, 45, 200, 360, dict(residual=False, relu=False, in_extra=56, out_extra=40)),   # tap-major K walk entered mid-tap
            (1, 9, 33, 128, 300, dict(k=1, pad=0, residual=True)),          # 1x1: two slices, one per split
            (1, 15, 21, 64, 290, dict(stride=2, residual=False)),           # stride 2, scalar epilogue
            (1, 6, 50, 256, 288, dict(out_f32=True, bn=False, relu=False)), # fp32 output
            (1, 24, 80, 1408, 256, dict(bn=False)),                         # the batch-1 cls conv: 30 tiles x 17 splits, chunk-major walk
            (1, 12, 40, 1152, 1152, dict(residual=True)),                   # whole-line epilogue in phase 2, splits of unequal length
        ]
    if cfg in HALO_TILES:
        return [
            (1, 11, 37, 64, 136, dict(residual=True)),                 # ragged patch grid, partial N tile
            (2, 8, 16, 128, 256, dict(residual=False, relu=False)),    # exactly one patch per image
            (1, 17, 50, 192, 128, dict(residual=True, in_extra=64, out_extra=128)),
        ]
    if cfg in NARROW:
        n = NARROW[cfg]
        return [
            (1, 13, 27, 64, n, dict(residual=True)),
            (2, 9, 31, 24, n - 5, dict(residual=False, bn=False)),     # Cout % 4 != 0 -> scalar epilogue
            (1, 10, 18, 72, n - 8, dict(k=1, pad=0, in_extra=8, out_extra=8)),
        ]
    return [
        (1, 13, 27, 64, 400, dict(residual=True)),                      # M = 351, two / three / four N tiles, last one partial
        (2, 7, 45, 200, 360, dict(residual=False, relu=False, in_extra=56, out_extra=40)),   # Cin not a K-slice multiple (tap-major walk)
        (1, 9, 33, 128, 300, dict(k=1, pad=0, residual=True)),          # 1x1
        (1, 15, 21, 64, 290, dict(stride=2, residual=False)),           # stride 2, Cout % 4 != 0 -> scalar epilogue
        (1, 6, 50, 256, 288, dict(out_f32=True, bn=False, relu=False)), # fp32 output (final head convs)
    ]


@pytest.mark.parametrize('cfg', PRODUCTION_TILES)
def test_every_production_tile_bf16_vs_oracle(cfg):
    for i, (B, H, W, Cin, Cout, kw) in enumerate(_shapes_for(cfg)):
        ulp, rel = run_case(B, H, W, Cin, Cout, dtype=torch.bfloat16, seed=100 + i, cfg=cfg, **kw)
        assert ulp <= 1.0, 'tile %d shape %s: %.2f bf16 ulp (rel %.2e)' % (cfg, (B, H, W, Cin, Cout, kw), ulp, rel)


@pytest.mark.parametrize('cfg', PRODUCTION_TILES)
def test_every_production_tile_fp16_vs_oracle(cfg):
    """fp16 storage (VD3D_F16, BASELINE config 5): the same tiles on v_mfma_f32_*_f16, each output within ONE fp16 ulp."""
    for i, (B, H, W, Cin, Cout, kw) in enumerate(_shapes_for(cfg)):
        ulp, rel = run_case(B, H, W, Cin, Cout, dtype=torch.float16, seed=300 + i, cfg=cfg, **kw)
        assert ulp <= 1.0, 'tile %d shape %s: %.2f fp16 ulp (rel %.2e)' % (cfg, (B, H, W, Cin, Cout, kw), ulp, rel)


@pytest.mark.parametrize('cfg', [c for c in PRODUCTION_TILES if c not in BF16_TILES])
def test_every_production_tile_fp32_vs_oracle(cfg):
    for i, (B, H, W, Cin, Cout, kw) in enumerate(_shapes_for(cfg)):
        ulp, rel = run_case(B, H, W, Cin, Cout, dtype=torch.float32, seed=200 + i, cfg=cfg, **kw)
        assert rel <= 2e-5, 'tile %d shape %s: rel %.2e' % (cfg, (B, H, W, Cin, Cout, kw), rel)


def test_forced_tile_that_cannot_run_the_shape_is_an_error():
    """The product library never computes with a kernel that is wrong for the shape: a halo tile forced on a 1x1 conv, a
    bf16-only tile forced in fp32 mode and an id that is not a production tile all fail loudly."""
    from visualdet3d_amd._lib import Vd3dError
    with pytest.raises(Vd3dError):
        run_case(1, 8, 16, 64, 128, k=1, pad=0, cfg=21)
    with pytest.raises(Vd3dError):
        run_case(1, 8, 16, 64, 352, dtype=torch.float32, cfg=50)
    for bad in (93, 1, 7):                # 93: a timing ablation of the tuning build; 1: the retired v1 kernel
        with pytest.raises(Vd3dError):
            run_case(1, 8, 16, 64, 128, cfg=bad)
    ulp, _ = run_case(1, 8, 16, 64, 128, cfg=0)       # and the override is really back to the heuristic afterwards
    assert ulp <= 1.0


# ---- the bench's own layer shapes, natural dispatch (BASELINE config 2: 8 pairs of 384 x 1280 -> 16 stacked images) --------
BENCH_SHAPES = [
    # name, B, H, W, Cin, Cout, kwargs                                   tile the heuristic picks (conv_igemm.hip dispatch)
    ('head 1408->1408 + res', 8, 24, 80, 1408, 1408, dict(residual=True)),         # 256x352 16x16x32 strips
    ('head 1408->1408', 8, 24, 80, 1408, 1408, dict(residual=False)),
    ('neck 1152->1152 + res', 8, 24, 80, 1152, 1152, dict(residual=True)),         # 256x288 16x16x32 strips, ring of 6
    ('reg out 1408->576 f32', 8, 24, 80, 1408, 576, dict(out_f32=True, bn=False, relu=False)),   # 128x192
    ('cls 1408->256', 8, 24, 80, 1408, 256, dict(bn=False)),                        # halo 8x16x128
    ('cls 256->144 f32', 8, 24, 80, 256, 144, dict(out_f32=True, bn=False, relu=False)),
    ('layer3 256->256 + res', 16, 24, 80, 256, 256, dict(residual=True)),           # K-split resident weights (240 tiles of 256 x 128 would be ONE round: measured slower in the model)
    ('layer2 128->128 + res', 16, 48, 160, 128, 128, dict(residual=True)),          # halo 8x32x128
    ('layer1 64->64 + res', 16, 96, 320, 64, 64, dict(residual=True)),              # resident weights
    ('layer2.0 64->128 s2', 16, 96, 320, 64, 128, dict(stride=2)),                  # 128x128
    ('layer3.0 128->256 s2', 16, 48, 160, 128, 256, dict(stride=2)),
    ('ghost 384->384', 8, 24, 80, 384, 384, dict()),
    ('neck 288->288 + res', 8, 24, 80, 288, 288, dict(residual=True)),              # 128x144 16x16x32
    ('cls 256->256', 8, 24, 80, 256, 256, dict(bn=False)),                          # register-resident weights, 4 slices
    ('r50 layer3 256->256 + res', 64, 18, 80, 256, 256, dict(residual=True)),      # 256 x 128 tiles, 2.81 rounds
    ('r50 head 2176->2176', 16, 18, 80, 2176, 2176, dict(residual=True)),           # 256x320 16x16x32 strips (6.8 of them)
]


@pytest.mark.parametrize('case', BENCH_SHAPES, ids=[c[0] for c in BENCH_SHAPES])
def test_bench_shapes_natural_dispatch_bf16_vs_oracle(case):
    name, B, H, W, Cin, Cout, kw = case
    torch.set_num_threads(min(64, torch.get_num_threads()))
    ulp, rel = run_case(B, H, W, Cin, Cout, dtype=torch.bfloat16, seed=7, cfg=0, **kw)
    assert ulp <= 1.0, '%s: %.2f bf16 ulp (rel %.2e)' % (name, ulp, rel)


# ---- the shapes of a batch-1 call (the reference's contract: one frame per test_forward): natural dispatch picks the split-K path ----
B1_SHAPES = [
    ('b1 head 1408->1408 + res', 1, 24, 80, 1408, 1408, dict(residual=True)),       # 165 tiles x 3 splits
    ('b1 neck 1152->1152 + res', 1, 24, 80, 1152, 1152, dict(residual=True)),
    ('b1 reg out 1408->576 f32', 1, 24, 80, 1408, 576, dict(out_f32=True, bn=False, relu=False)),
    ('b1 cls 1408->256', 1, 24, 80, 1408, 256, dict(bn=False)),
    ('b1 neck 288->288 + res', 1, 24, 80, 288, 288, dict(residual=True)),
    ('b1 mono cls 512->512', 1, 24, 80, 512, 512, dict(bn=False)),
    ('b1 mono cls out 512->64 f32', 1, 24, 80, 512, 64, dict(out_f32=True, bn=False, relu=False)),    # 128 x 64 split tiles
    ('b1 look-ground 256->1', 1, 24, 80, 256, 1, dict(bn=False, relu=False)),
    ('b1 mono reg out 256->384 f32', 1, 24, 80, 256, 384, dict(out_f32=True, bn=False, relu=False)),
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('case', B1_SHAPES, ids=[c[0] for c in B1_SHAPES])
def test_batch1_shapes_natural_dispatch_vs_oracle(case, dtype):
    from visualdet3d_amd import _lib
    name, B, H, W, Cin, Cout, kw = case
    ulp, rel = run_case(B, H, W, Cin, Cout, dtype=dtype, seed=11, cfg=0, **kw)
    if dtype == torch.float32:
        assert rel <= 2e-5, '%s: rel %.2e' % (name, rel)
    else:
        assert ulp <= 1.0, '%s: %.2f ulp (rel %.2e)' % (name, ulp, rel)


def test_splitk_is_what_batch1_dispatch_picks_and_is_deterministic():
    """vd3d_conv2d_workspace_bytes > 0 exactly for the low-parallelism deep-K shapes (and 0 for the full-chip batched shapes: nothing
    changes for them); without a workspace the same call runs unsplit and agrees to fp32 summation noise; two split runs are
    bit-identical (partials are added in split order, no atomics)."""
    import ctypes as C
    from visualdet3d_amd import _lib, hip_ops as ops
    g = torch.Generator().manual_seed(5)

    def params(B, H, W, Cin, Cout):
        x = torch.randn(B, H, W, Cin, generator=g).cuda().to(torch.bfloat16)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda()
        return x, ops.pack_conv(w, None, None, torch.bfloat16, 1, 1, 1)

    x, pc = params(1, 24, 80, 1408, 1408)
    a = ops.conv2d(x, pc, relu=True)
    b = ops.conv2d(x, pc, relu=True)
    assert torch.equal(a, b)
    orig = _lib.lib().vd3d_conv2d_workspace_bytes
    seen = []
    assert list(pc.ws_need.values()) == [6 * 40 * 256 * 288 * 4]     # hip_ops asks the library once per launch geometry and remembers the answer
    try:
        pc.ws_need.clear()
        _lib.lib().vd3d_conv2d_workspace_bytes = lambda p: (seen.append(orig(p)), 0)[1]     # hip_ops then passes no workspace: unsplit
        c = ops.conv2d(x, pc, relu=True)
    finally:
        _lib.lib().vd3d_conv2d_workspace_bytes = orig
        pc.ws_need.clear()
    assert seen and seen[0] == 6 * 40 * 256 * 288 * 4, seen          # 8 x 5 strip tiles of 256 x 288, six splits (plan_splitk_strip)
    with _lib.test_switch('VD3D_NO_STRIP_SPLIT'):                    # the 128 x 128-tile plan: 15 x 11 tiles, three splits
        assert orig is not None and _lib.lib().vd3d_conv2d_workspace_bytes is orig
        seen.clear()
        try:
            _lib.lib().vd3d_conv2d_workspace_bytes = lambda p: (seen.append(orig(p)), seen[-1])[1]
            a2 = ops.conv2d(x, pc, relu=True)
        finally:
            _lib.lib().vd3d_conv2d_workspace_bytes = orig
        assert seen[0] == 3 * 165 * 128 * 128 * 4, seen
        d2 = (a.float() - a2.float()).abs().max().item() / a2.float().abs().max().item()
        assert d2 < 2.0 ** -7, d2
    d = (a.float() - c.float()).abs().max().item() / c.float().abs().max().item()
    assert 0 < d < 2.0 ** -7, d            # a different summation order (so it really was another path), within one bf16 ulp
    x8, pc8 = params(8, 24, 80, 1408, 1408)
    seen.clear()
    try:
        _lib.lib().vd3d_conv2d_workspace_bytes = lambda p: (seen.append(orig(p)), 0)[1]
        ops.conv2d(x8, pc8, relu=True)
    finally:
        _lib.lib().vd3d_conv2d_workspace_bytes = orig
    assert seen == [0]


@pytest.mark.parametrize('cfg', [44, 42, 30])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_whole_line_epilogue_of_the_tile_kernels(cfg, dtype):
    """conv_epilogue_lines: short-K layers with Cout % 64 == 0 re-lay their 16-bit outputs through LDS so that store
    instructions cover whole 128-byte lines (ConvArgs::line_store).  Against the oracle, and bit-identical to the accumulator-layout
    stores it replaces (VD3D_NO_LINE_STORE=1): same arithmetic, only the store shape differs."""
    from visualdet3d_amd import _lib
    cases = [
        (2, 9, 31, 256, 128, dict(k=1, pad=0, residual=True)),                    # ragged M (558 pixels), two 64-channel strips
        (1, 13, 27, 64, 192, dict(residual=True, out_extra=64)),                  # 3x3, channel-slice view (pixel stride 256)
        (3, 16, 32, 128, 64, dict(k=1, pad=0, residual=False, relu=False)),       # one strip: the other waves of the N tile idle
        (1, 7, 45, 512, 256, dict(k=1, pad=0, stride=2, residual=False)),         # strided 1x1 (down-sample branch)
    ]
    for i, (B, H, W, Cin, Cout, kw) in enumerate(cases):
        ulp, rel = run_case(B, H, W, Cin, Cout, dtype=dtype, seed=400 + i, cfg=cfg, **kw)
        assert ulp <= 1.0, 'tile %d shape %s: %.2f ulp (rel %.2e)' % (cfg, (B, H, W, Cin, Cout, kw), ulp, rel)
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 9, 31, 256, generator=g).cuda().to(dtype)
    w = (torch.randn(128, 256, 1, 1, generator=g) * 0.1).cuda()
    res = torch.randn(2, 9, 31, 128, generator=g).cuda().to(dtype)
    pc = ops.pack_conv(w, None, None, dtype, 1, 0, 1)
    with forced_tile(cfg):
        a = ops.conv2d(x, pc, residual=res, relu=True)
        with _lib.test_switch('VD3D_NO_LINE_STORE'):
            b = ops.conv2d(x, pc, residual=res, relu=True)
    assert torch.equal(a, b)


@pytest.mark.parametrize('cfg', [79, 42, 44])
def test_grouped_tile_order_of_huge_1x1_gemms(cfg):
    """ConvArgs::group_m (1x1 GEMMs whose pixel matrix exceeds the on-die caches: the DCN column GEMM of BASELINE config 3) only
    permutes which workgroup computes which tile: forced on small shapes (VD3D_FORCE_GROUP_M=1), results must equal the plain order
    bit for bit -- tile counts that are and are not multiples of the group (4 pixel tiles), one and several N tiles."""
    from visualdet3d_amd import _lib, hip_ops as ops
    g = torch.Generator().manual_seed(21)
    for (B, H, W, Cin, Cout) in [(2, 18, 40, 192, 544), (1, 23, 31, 128, 272), (3, 16, 32, 64, 1088)]:
        x = torch.randn(B, H, W, Cin, generator=g).cuda().to(torch.bfloat16)
        w = (torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1).cuda()
        pc = ops.pack_conv(w, None, None, torch.bfloat16, 1, 0, 1)
        with forced_tile(cfg):
            a = ops.conv2d(x, pc, relu=True)
            with _lib.test_switch('VD3D_FORCE_GROUP_M'):
                b = ops.conv2d(x, pc, relu=True)
        assert torch.equal(a, b), (cfg, B, H, W, Cin, Cout)
    ulp, rel = run_case(2, 18, 40, 192, 544, k=1, pad=0, residual=True, dtype=torch.bfloat16, seed=77, cfg=cfg)
    assert ulp <= 1.0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,W,Cb', [(1, 16, 64, 32), (2, 37, 91, 32), (1, 9, 130, 16), (3, 64, 250, 32), (4, 512, 1760, 32)])
def test_level_pair_kernel_is_bit_identical_to_two_launches(dtype, B, H, W, Cb):
    """vd3d_conv2d_pair (DLA level0 -> level1, backbones/dla.py:118-121: conv 3x3 / s1 16 -> 16 + BN + ReLU, conv 3x3 / s2 16 -> 32 + BN +
    ReLU) against the two small-channel launches it replaces (each of which tests above pin to the oracle): same MFMA shapes, tap
    order and rounding points -> identical bits.  Odd sizes (ragged tiles on both levels, odd H / W under the stride), more tiles
    than workgroups (the last case is BASELINE config 5's own shape: 14 080 tiles)."""
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, H, W, 16, generator=g).cuda().to(dtype)

    def mk(cin, cout, stride, seed):
        gg = torch.Generator().manual_seed(seed)
        w = torch.randn(cout, cin, 3, 3, generator=gg) * (2.0 / (9 * cin)) ** 0.5
        bn = (torch.rand(cout, generator=gg) + 0.5, torch.randn(cout, generator=gg) * 0.1, torch.randn(cout, generator=gg) * 0.1,
              torch.rand(cout, generator=gg) + 0.5, 1e-5)
        return ops.pack_conv(w.cuda(), None, tuple(t.cuda() if torch.is_tensor(t) else t for t in bn), dtype, stride, 1, 1)

    pa, pb = mk(16, 16, 1, 1), mk(16, Cb, 2, 2)
    assert ops.conv2d_pair_supported(pa, pb)
    want = ops.conv2d(ops.conv2d(x, pa, relu=True), pb, relu=True)
    got = ops.conv2d_pair(x, pa, pb)
    torch.cuda.synchronize()
    assert got.shape == want.shape
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(8, 24, 80, 1408, 1408), (8, 24, 80, 1152, 1152), (3, 17, 41, 1408, 1408), (2, 18, 80, 2176, 2176)],
                         ids=['c2 head', 'c2 neck', 'ragged', 'c3 head'])
def test_staggered_dma_schedule_of_the_strips_is_bit_identical(shape, dtype):
    """The 352 / 288 column strips issue their LDS-DMA pieces on two schedules (waves 0-3 right behind the slice's barrier, waves 4-7 at a
    later fragment: conv_igemm.hip launch_strip352 / launch_strip288); VD3D_CONV_NO_STAGGER=1 puts every wave on the first.  Where a piece
    is issued changes nothing about what is computed: same operands, same k order -> the outputs are BIT-IDENTICAL (a piece that
    did not land before its barrier would show here)."""
    from visualdet3d_amd import _lib, hip_ops as ops
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn(B, H, W, Cin, generator=g).cuda().to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda()
    res = torch.randn(B, H, W, Cout, generator=g).cuda().to(dtype)
    pc = ops.pack_conv(w, None, None, dtype, 1, 1, 1)
    outs = []
    for tile in (50, 54, 52):
        with forced_tile(tile):
            a = ops.conv2d(x, pc, residual=res, relu=True)
            with _lib.test_switch('VD3D_CONV_NO_STAGGER'):
                b = ops.conv2d(x, pc, residual=res, relu=True)
            for _ in range(3):                                    # run to run as well
                assert torch.equal(ops.conv2d(x, pc, residual=res, relu=True).view(torch.int16), a.view(torch.int16))
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), 'tile %d: staggered schedule differs (max %.3e)' % (tile, (a.float() - b.float()).abs().max().item())
        outs.append(a)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_two_way_split_on_the_288_strips(dtype):
    """Config 2's reg-tower output conv (1408 -> 576 at 8 x 24 x 80, fp32 output): its 256 x 288 strip tiles fill less than half the chip (120 tiles), so
    natural dispatch runs the strips with K split in two + the reduction launch (conv_igemm.hip plan_splitk_strip).  Against the oracle at the fp32-output bar,
    against the unsplit path (VD3D_NO_STRIP_SPLIT=1) to summation noise, run-to-run bit-identical; and a 16-bit-output variant with residual + ReLU."""
    from visualdet3d_amd import _lib, hip_ops as ops
    ulp, rel = run_case(8, 24, 80, 1408, 576, dtype=dtype, seed=3, cfg=0, out_f32=True, bn=False, relu=False)
    assert rel <= 1e-4, rel
    ulp, rel = run_case(8, 24, 80, 1408, 576, dtype=dtype, seed=4, cfg=0, residual=True)
    assert ulp <= 1.0, (ulp, rel)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(8, 24, 80, 1408, generator=g).cuda().to(dtype)
    pc = ops.pack_conv((torch.randn(576, 1408, 3, 3, generator=g) * (2.0 / (9 * 1408)) ** 0.5).cuda(), None, None, dtype, 1, 1, 1)
    seen = []
    orig = _lib.lib().vd3d_conv2d_workspace_bytes
    try:
        _lib.lib().vd3d_conv2d_workspace_bytes = lambda p: (seen.append(orig(p)), seen[-1])[1]
        a = ops.conv2d(x, pc, relu=False, out_f32=True)
    finally:
        _lib.lib().vd3d_conv2d_workspace_bytes = orig
    assert seen and seen[-1] == 2 * 120 * 256 * 288 * 4, seen          # the split path was what ran
    b = ops.conv2d(x, pc, relu=False, out_f32=True)
    with _lib.test_switch('VD3D_NO_STRIP_SPLIT'):
        c = ops.conv2d(x, pc, relu=False, out_f32=True)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert ((a - c).abs().max() / c.abs().max()).item() < 2e-5


def test_two_way_split_when_the_strips_fill_their_rounds_badly():
    """Round 5, the second case of plan_splitk_strip: config 3's 2176 -> 576 reg-tower output conv at 32 x 18 x 80 (fp32 output) -- 180 x 2 = 360 strip tiles are
    1.4 rounds of 256 CUs; with K (306 slices) split in two the same strips are 720 workgroups = 2.81 rounds.  The split path is what runs (workspace query),
    the result meets the fp32-output bar against the oracle, equals the unsplit path to summation noise and is run-to-run bit-identical."""
    from visualdet3d_amd import _lib, hip_ops as ops
    torch.set_num_threads(min(64, torch.get_num_threads()))
    ulp, rel = run_case(32, 18, 80, 2176, 576, dtype=torch.bfloat16, seed=5, cfg=0, out_f32=True, bn=False, relu=False)
    assert rel <= 1e-4, rel
    g = torch.Generator().manual_seed(10)
    x = torch.randn(32, 18, 80, 2176, generator=g).cuda().to(torch.bfloat16)
    pc = ops.pack_conv((torch.randn(576, 2176, 3, 3, generator=g) * (2.0 / (9 * 2176)) ** 0.5).cuda(), None, None, torch.bfloat16, 1, 1, 1)
    a = ops.conv2d(x, pc, relu=False, out_f32=True)
    assert pc.ws_need and max(pc.ws_need.values()) == 2 * 360 * 256 * 288 * 4, pc.ws_need          # the two-way split of the 360 strip tiles
    b = ops.conv2d(x, pc, relu=False, out_f32=True)
    with _lib.test_switch('VD3D_NO_STRIP_SPLIT'):
        c = ops.conv2d(x, pc, relu=False, out_f32=True)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert ((a - c).abs().max() / c.abs().max()).item() < 2e-5

