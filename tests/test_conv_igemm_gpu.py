"""GPU: the fused implicit-GEMM conv kernel (vd3d_conv2d_igemm) against torch CPU fp32 convolution.
fp32 mode (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains): rtol 1e-5 of the output scale.
bf16 mode: inputs/weights rounded to bf16 on both sides, fp32 accumulate, output rounded to bf16 -> 1 bf16 ulp."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(B, H, W, Cin, Cout, k, stride, pad, dil, residual, relu, dtype, bias=True, bn=True, in_extra=0, out_extra=0, out_f32=False, seed=0):
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1 if bias else None
    bnp = None
    if bn:
        bnp = (torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1, torch.randn(Cout, generator=g) * 0.1,
               torch.rand(Cout, generator=g) + 0.5, 1e-5)
    rnd = (lambda t: t.to(torch.bfloat16).float()) if dtype == torch.bfloat16 else (lambda t: t)
    # reference on CPU
    y = F.conv2d(rnd(x), rnd(w), None, stride, pad, dil)
    scale = torch.ones(Cout)
    shift = b.clone() if bias else torch.zeros(Cout)
    if bn:
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
    # HIP
    dev = 'cuda'
    xin = torch.zeros(B, H, W, Cin + in_extra, dtype=dtype, device=dev)
    xin[..., in_extra:] = x.permute(0, 2, 3, 1).to(dev).to(dtype)
    xv = xin[..., in_extra:]
    pc = ops.pack_conv(w.to(dev), b.to(dev) if bias else None, tuple(t.to(dev) if torch.is_tensor(t) else t for t in bnp) if bn else None,
                       dtype, stride, pad, dil)
    odt = torch.float32 if out_f32 else dtype
    obuf = torch.full((B, Ho, Wo, Cout + out_extra), 7.0, dtype=odt, device=dev)
    ov = obuf[..., :Cout]
    rv = res.permute(0, 2, 3, 1).contiguous().to(dev).to(dtype) if residual else None
    out = ops.conv2d(xv, pc, out=ov, residual=rv, relu=relu, out_f32=out_f32)
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    if out_extra:
        assert bool((obuf[..., Cout:] == 7.0).all()), 'kernel wrote outside its channel slice'
    scale_ref = y.abs().max().item()
    err = (got - y).abs().max().item() / scale_ref
    return err


SHAPES = [
    # B, H, W, Cin, Cout, k, stride, pad, dil, residual, relu
    (2, 24, 80, 64, 64, 3, 1, 1, 1, True, True),
    (1, 24, 40, 128, 256, 3, 2, 1, 1, False, True),
    (2, 12, 20, 256, 256, 3, 1, 1, 1, True, True),
    (1, 13, 27, 64, 128, 1, 2, 0, 1, False, False),     # 1x1 stride-2 downsample, ragged M
    (1, 24, 80, 24, 24, 3, 1, 1, 1, False, True),       # Cin < K slice (ghost primary conv)
    (1, 12, 40, 72, 72, 3, 1, 1, 1, True, True),
    (1, 6, 20, 288, 288, 3, 1, 1, 1, True, True),
    (1, 6, 20, 384, 144, 3, 1, 1, 1, False, False),     # Cout not a tile multiple
    (1, 9, 11, 64, 27, 3, 1, 1, 1, False, False),       # Cout % 4 != 0 -> scalar epilogue (DCN offset conv)
    (1, 10, 18, 64, 64, 3, 1, 2, 2, False, True),       # dilation 2
    (3, 6, 20, 256, 8, 1, 1, 0, 1, False, True),        # cost-volume down-sample
]


@pytest.mark.parametrize('shape', SHAPES)
def test_conv_fp32(shape):
    err = _case(*shape, dtype=torch.float32)
    assert err < 2e-5, err


@pytest.mark.parametrize('shape', SHAPES)
def test_conv_bf16(shape):
    err = _case(*shape, dtype=torch.bfloat16)
    assert err < 1e-2, err  # bf16 output rounding: 2^-8 relative


def test_conv_channel_slices_and_fp32_out():
    # reads a channel slice, writes a channel slice of a wider buffer, fp32 output from bf16 compute
    err = _case(2, 12, 20, 96, 96, 3, 1, 1, 1, False, True, torch.bfloat16, in_extra=24, out_extra=96)
    assert err < 1e-2
    err = _case(1, 6, 20, 256, 144, 3, 1, 1, 1, False, False, torch.bfloat16, bn=False, out_f32=True)
    assert err < 2e-3  # only inputs are rounded; fp32 epilogue
    err = _case(2, 12, 20, 96, 96, 3, 1, 1, 1, True, True, torch.float32, in_extra=24, out_extra=96)
    assert err < 2e-5


def test_conv_big_k():
    # the 1408-channel head conv shape at reduced spatial size: K = 12672, 3 N tiles
    err = _case(1, 6, 20, 1408, 320, 3, 1, 1, 1, True, True, torch.float32)
    assert err < 5e-5
    err = _case(1, 6, 20, 1408, 320, 3, 1, 1, 1, True, True, torch.bfloat16)
    assert err < 1e-2


def test_stem_and_pools():
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 64, 96, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5, 1e-5)
    s = bn[0] / torch.sqrt(bn[3] + 1e-5)
    t = bn[1] - bn[2] * s
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 1e-2)):
        rnd = (lambda v: v.to(torch.bfloat16).float()) if dtype == torch.bfloat16 else (lambda v: v)
        y = F.relu(F.conv2d(rnd(x), rnd(w), None, 2, 3) * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1))
        pc = ops.pack_stem_conv(w.cuda(), tuple(v.cuda() if torch.is_tensor(v) else v for v in bn), dtype)
        out = ops.stem_conv(x.cuda(), pc, dtype)
        got = out.float().cpu().permute(0, 3, 1, 2)
        assert got.shape == y.shape
        assert ((got - y).abs().max() / y.abs().max()).item() < tol
        yr = rnd(y)
        mp = ops.maxpool3x3s2(out).float().cpu().permute(0, 3, 1, 2)
        assert torch.equal(mp, F.max_pool2d(out.float().cpu().permute(0, 3, 1, 2), 3, 2, 1))
        ap = ops.avgpool2x2(out).float().cpu().permute(0, 3, 1, 2)
        ref = rnd(F.avg_pool2d(out.float().cpu().permute(0, 3, 1, 2), 2))
        assert ((ap - ref).abs().max() / ref.abs().max()).item() < (1e-6 if dtype == torch.float32 else 1e-2)


def test_dwconv_and_copy():
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(4)
    C = 96
    x = torch.randn(2, C, 12, 20, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g) * 0.3
    bn = (torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5, 1e-5)
    s = bn[0] / torch.sqrt(bn[3] + 1e-5)
    t = bn[1] - bn[2] * s
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-2)):
        rnd = (lambda v: v.to(torch.bfloat16).float()) if dtype == torch.bfloat16 else (lambda v: v)
        y = F.relu(F.conv2d(rnd(x), w, None, 1, 1, 1, groups=C) * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1))
        pd = ops.pack_dwconv(w.cuda(), tuple(v.cuda() if torch.is_tensor(v) else v for v in bn))
        buf = torch.zeros(2, 12, 20, 3 * C, dtype=dtype, device='cuda')
        buf[..., C:2 * C] = x.permute(0, 2, 3, 1).cuda().to(dtype)
        ops.dwconv3x3(buf[..., C:2 * C], pd, out=buf[..., 2 * C:], relu=True)
        got = buf[..., 2 * C:].float().cpu().permute(0, 3, 1, 2)
        assert ((got - y).abs().max() / y.abs().max()).item() < tol
        ops.copy_channels(buf[..., 2 * C:], buf[..., :C])
        assert torch.equal(buf[..., :C], buf[..., 2 * C:])


def test_conv_resident_weights_64():
    """The persistent resident-weight kernel (3x3 s1 p1, 64 -> 64, bf16): fewer tiles than CUs, many tiles per workgroup,
    with/without residual, ReLU, BN; channel-slice input and output."""
    assert _case(1, 8, 16, 64, 64, 3, 1, 1, 1, False, False, torch.bfloat16, bias=False, bn=False) < 1e-2       # one tile
    assert _case(1, 16, 48, 64, 64, 3, 1, 1, 1, True, False, torch.bfloat16, seed=1) < 1e-2
    assert _case(3, 96, 320, 64, 64, 3, 1, 1, 1, True, True, torch.bfloat16, seed=2) < 1e-2                      # 720 tiles
    assert _case(2, 96, 320, 64, 64, 3, 1, 1, 1, False, True, torch.bfloat16, bn=False, seed=3) < 1e-2
    assert _case(2, 24, 32, 64, 64, 3, 1, 1, 1, True, True, torch.bfloat16, in_extra=64, out_extra=64, seed=4) < 1e-2
    assert _case(2, 20, 40, 64, 64, 3, 1, 1, 1, True, True, torch.bfloat16, out_extra=64, seed=5) < 1e-2      # ragged H and W
    assert _case(1, 5, 7, 64, 64, 3, 1, 1, 1, False, True, torch.bfloat16, seed=6) < 1e-2                       # smaller than a tile


def test_stem_conv_pool_fused():
    """vd3d_stem_conv_pool (conv 7x7/s2 + BN + ReLU + maxpool 3x3/s2 in one kernel, bf16) against torch CPU on bf16-rounded
    operands, and against the unfused HIP path (same MFMA chain and rounding point before the pool: identical up to an
    occasional 1-ulp bf16 difference from mul/add contraction in the two epilogues)."""
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(11)
    for B, H, W in [(1, 32, 64), (3, 96, 320), (2, 64, 192)]:
        img = torch.randn(B, 3, H, W, generator=g)
        w = torch.randn(64, 3, 7, 7, generator=g) * 0.12
        bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1,
              torch.rand(64, generator=g) + 0.5, 1e-5)
        rnd = lambda t: t.to(torch.bfloat16).float()
        y = F.conv2d(rnd(img), rnd(w), None, 2, 3)
        s = bn[0] / torch.sqrt(bn[3] + bn[4])
        y = F.relu(y * s.view(1, -1, 1, 1) + (bn[1] - bn[2] * s).view(1, -1, 1, 1))
        want = F.max_pool2d(rnd(y), 3, 2, 1)
        pc = ops.pack_stem_conv(w.cuda(), tuple(t.cuda() if torch.is_tensor(t) else t for t in bn), torch.bfloat16)
        halves = [img[:1].cuda().contiguous(), img[1:].cuda().contiguous()] if B > 1 else img.cuda()
        got = ops.stem_conv_pool(halves, pc, torch.bfloat16)
        unfused = ops.maxpool3x3s2(ops.stem_conv(img.cuda(), pc, torch.bfloat16))
        torch.cuda.synchronize()
        assert got.shape == (B, H // 4, W // 4, 64)
        d = (got.float() - unfused.float()).abs()
        assert float((d > 0).float().mean()) < 1e-4 and bool((d <= unfused.float().abs() * 2.0 ** -7 + 1e-30).all()), d.max()
        err = (got.float().cpu().permute(0, 3, 1, 2) - want).abs().max().item() / want.abs().max().item()
        assert err < 1e-2, err


@pytest.mark.parametrize('shape', [
    (2, 16, 64, 128, 128, 3, 1, 1, 1, True, True),      # halo kernel 8x32 px x 128 ch (layer2)
    (2, 16, 32, 256, 256, 3, 1, 1, 1, True, True),      # halo kernel 8x16 px x 256 ch (layer3)
    (1, 8, 16, 256, 256, 3, 1, 1, 1, False, False),     # one tile
    (1, 16, 32, 1408, 256, 3, 1, 1, 1, False, True),    # long K, 8x16 px x 128 ch (cls conv)
    (1, 24, 80, 64, 128, 3, 1, 1, 1, False, True),      # single 64-channel chunk (9 steps)
    (1, 64, 220, 128, 128, 3, 1, 1, 1, True, True),     # ragged W (220 = 6.9 x 32): KM3D level at 512 x 1760
    (1, 30, 110, 256, 256, 3, 1, 1, 1, False, True),    # ragged H and W on the 8x16 tile
])
def test_conv_halo_kernel_shapes(shape):
    assert _case(*shape, dtype=torch.bfloat16) < 1e-2
    assert _case(*shape, dtype=torch.float32) < 5e-5


def test_conv_batch_chunking_for_large_views(monkeypatch):
    """Inputs whose view spans more than the 32-bit buffer range are processed in batch chunks (hit by KM3D's head at B = 16,
    512 x 1760): force the path with a small limit and compare with the one-launch result."""
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(5, 16, 32, 192, generator=g).cuda().to(torch.bfloat16)
    w = torch.randn(64, 64, 3, 3, generator=g).cuda() * 0.05
    pc = ops.pack_conv(w, None, None, torch.bfloat16, 1, 1, 1)
    res = torch.randn(5, 16, 32, 64, generator=g).cuda().to(torch.bfloat16)
    xv = x[..., 64:128]                                   # channel slice of a wider buffer
    want = ops.conv2d(xv, pc, residual=res, relu=True)
    monkeypatch.setattr(ops, '_MAX_IN_BYTES', 2 * 16 * 32 * 192 * 2 + 100)
    got = ops.conv2d(xv, pc, residual=res, relu=True)
    assert torch.equal(got, want)


@pytest.mark.parametrize('B0,B1,H,W', [(2, 2, 128, 192), (3, 0, 96, 320), (8, 8, 384, 1280)])
def test_fused_stem_reads_the_fp32_image_directly(B0, B1, H, W):
    """vd3d_stem_conv_pool_f32 (conv 7x7/s2 + BN + ReLU + max pool straight from the fp32 NCHW image(s): no packed copy, no pack
    launch) against the round-2 path (vd3d_pack_image_nhwc4 + vd3d_stem_conv_pool): the bf16 rounding of the image is the same single
    rounding and the MFMA sequence is unchanged -> BIT-IDENTICAL pooled maps; stereo (two tensors stacked on the batch axis by the
    kernel), one tensor, ragged last workgroup round, the bench size."""
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(B0 * 10 + B1)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5, 1e-5)
    pc = ops.pack_stem_conv(w.cuda(), tuple(t.cuda() if torch.is_tensor(t) else t for t in bn), torch.bfloat16)
    imgs = [torch.randn(B0, 3, H, W, generator=g).cuda()] + ([torch.randn(B1, 3, H, W, generator=g).cuda()] if B1 else [])
    a = ops.stem_conv_pool(imgs, pc, torch.bfloat16)
    b = ops.stem_conv_pool(imgs, pc, torch.bfloat16, packed_first=True)
    torch.cuda.synchronize()
    assert a.shape == b.shape == (B0 + B1, H // 4, W // 4, 64)
    assert torch.equal(a, b), (a.float() - b.float()).abs().max().item()


def test_dwconv_runs_of_four_pixels_equal_the_one_pixel_kernel_bit_for_bit():
    """vd3d_dwconv3x3 (round 6): a thread owns four output pixels of a row (weights / folded BN from LDS); `VD3D_DWCONV_PLAIN` is the one-pixel kernel.  Same taps
    in the same order in fp32 -> identical bits: the ghost modules' shapes (24 channels at stride 4 as a slice of the 72-channel buffer, 96, 384), widths that are
    not multiples of four, one-pixel-wide and one-row images, every storage type."""
    from visualdet3d_amd import _lib, hip_ops as ops
    g = torch.Generator().manual_seed(9)
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        for (B, H, W, C, tot, off_in, off_out) in ((2, 24, 80, 24, 72, 24, 48), (1, 13, 37, 96, 288, 96, 192), (1, 7, 2, 384, 384, 0, 0), (2, 1, 9, 24, 24, 0, 0),
                                                  (1, 5, 1, 8, 8, 0, 0), (8, 96, 320, 24, 72, 24, 48)):
            w = torch.randn(C, 1, 3, 3, generator=g) * 0.3
            bn = (torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5, 1e-5)
            pd = ops.pack_dwconv(w.cuda(), tuple(v.cuda() if torch.is_tensor(v) else v for v in bn))
            buf = torch.randn(B, H, W, tot, generator=g).cuda().to(dtype)
            src = buf[..., off_in:off_in + C]
            same = off_in == off_out
            outs = []
            for plain in (False, True):
                dst = torch.full((B, H, W, tot), 3.0, dtype=dtype, device='cuda')
                o = dst[..., off_out:off_out + C]
                if plain:
                    with _lib.test_switch('VD3D_DWCONV_PLAIN'):
                        ops.dwconv3x3(src, pd, out=o, relu=not same)
                else:
                    ops.dwconv3x3(src, pd, out=o, relu=not same)
                torch.cuda.synchronize()
                outs.append(dst)
            assert torch.equal(outs[0], outs[1]), (dtype, B, H, W, C)
            if tot > C:
                assert bool((outs[0][..., :off_out] == 3.0).all()) and bool((outs[0][..., off_out + C:] == 3.0).all()), 'wrote outside its channel slice'
