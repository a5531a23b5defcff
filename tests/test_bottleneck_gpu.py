"""GPU: vd3d_conv2d_bottleneck (a whole ResNet Bottleneck of the 64-wide stage in one launch, csrc/conv_bottleneck.hip) against

  * the CPU oracle of the block -- torch fp32 convolutions on operands rounded exactly like the HIP path rounds them, every intermediate
    rounded ONCE to the storage type where the separate launches (and oracle/detector_oracle.py's `rnd`) round it (reference semantics:
    backbones/resnet.py:55-91): every element within one ulp of the storage type + the fp32 summation-order noise;
  * the separate launches of the same block (conv1, conv2, [downsample], conv3 + residual through vd3d_conv2d_igemm): the same roundings, so
    the two differ only where an intermediate sits on a rounding boundary: a handful of one-ulp differences.

Identity block (256 -> 64 -> 64 -> 256, residual = x) and the stage's first block (64 -> 64 -> 64 -> 256 + downsample conv), bf16 and fp16,
tiles that divide the image, ragged images, images smaller than a tile, many tiles per workgroup (config 3's 72 x 320 map)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bn(g, c):
    return (torch.rand(c, generator=g) * 0.4 + 0.8, torch.randn(c, generator=g) * 0.05, torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5, 1e-5)


def _fold(bn):
    s = bn[0] / torch.sqrt(bn[3] + bn[4])
    return s, bn[1] - bn[2] * s


def _make(cin, ds, seed):
    g = torch.Generator().manual_seed(seed)
    w = lambda o, i, k: torch.randn(o, i, k, k, generator=g) * (2.0 / (i * k * k)) ** 0.5
    P = dict(w1=w(64, cin, 1), bn1=_bn(g, 64), w2=w(64, 64, 3), bn2=_bn(g, 64), w3=w(256, 64, 1), bn3=_bn(g, 256))
    P['bn3'] = (P['bn3'][0] * 0.4,) + P['bn3'][1:]
    if ds:
        P['wd'], P['bnd'] = w(256, cin, 1), _bn(g, 256)
    return P, g


def _oracle(x, P, rnd):
    """x: NCHW fp32 (already rounded).  The block with every stored tensor of the unfused path rounded once."""
    def cbr(t, w, bn, pad, relu):
        s, h = _fold(bn)
        y = F.conv2d(t, rnd(w), None, 1, pad) * s.view(1, -1, 1, 1) + h.view(1, -1, 1, 1)
        return rnd(F.relu(y) if relu else y)
    t1 = cbr(x, P['w1'], P['bn1'], 0, True)
    t2 = cbr(t1, P['w2'], P['bn2'], 1, True)
    s3, h3 = _fold(P['bn3'])
    y = F.conv2d(t2, rnd(P['w3']), None, 1, 0) * s3.view(1, -1, 1, 1) + h3.view(1, -1, 1, 1)
    res = cbr(x, P['wd'], P['bnd'], 0, False) if 'wd' in P else x
    return F.relu(y + res)


def _packs(P, dt):
    from visualdet3d_amd import hip_ops as ops
    dev = lambda bn: tuple(t.cuda() if torch.is_tensor(t) else t for t in bn)
    pc1 = ops.pack_conv(P['w1'].cuda(), None, dev(P['bn1']), dt, 1, 0, 1)
    pc2 = ops.pack_conv(P['w2'].cuda(), None, dev(P['bn2']), dt, 1, 1, 1)
    pc3 = ops.pack_conv(P['w3'].cuda(), None, dev(P['bn3']), dt, 1, 0, 1)
    pcd = ops.pack_conv(P['wd'].cuda(), None, dev(P['bnd']), dt, 1, 0, 1) if 'wd' in P else None
    return pc1, pc2, pc3, pcd


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('ds', [False, True])
@pytest.mark.parametrize('B,H,W', [(2, 16, 32), (1, 13, 37), (1, 5, 7), (3, 24, 80)])
def test_fused_bottleneck_matches_the_oracle_and_the_separate_launches(dt, ds, B, H, W):
    from visualdet3d_amd import hip_ops as ops
    cin = 64 if ds else 256
    P, g = _make(cin, ds, seed=B * 100 + H + (7 if ds else 0))
    rnd = lambda t: t.to(dt).float()
    x = rnd(torch.randn(B, cin, H, W, generator=g).abs() * 0.8)              # (a block's input is a ReLU output)
    want = _oracle(x, P, rnd)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().to(dt)
    pc1, pc2, pc3, pcd = _packs(P, dt)
    assert ops.conv2d_bottleneck_supported(xd, pc1, pc2, pc3, pcd)
    got = ops.conv2d_bottleneck(xd, pc1, pc2, pc3, pcd)
    t1 = ops.conv2d(xd, pc1, relu=True)
    t2 = ops.conv2d(t1, pc2, relu=True)
    res = ops.conv2d(xd, pcd, relu=False) if ds else xd
    sep = ops.conv2d(t2, pc3, residual=res, relu=True)
    torch.cuda.synchronize()
    gotc = got.float().cpu().permute(0, 3, 1, 2)
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    scale = want.abs().max().item()
    # one ulp of the storage type (the output's own rounding) + one ulp-of-an-intermediate propagated (an intermediate on a rounding boundary moves
    # the output by ~ its weight x one ulp) + fp32 summation-order noise
    err = (gotc - want).abs()
    bound = want.abs() * ulp + scale * (4 * ulp * 0.05 + 3e-5)
    assert bool((err <= bound).all()), 'fused vs oracle: max excess %.3e (scale %.3f)' % ((err - bound).max().item(), scale)
    d = (got.float() - sep.float()).abs()
    frac = float((d > 0).float().mean())
    assert frac < 2e-2 and bool((d <= sep.float().abs() * 2 * ulp + scale * 2.5 * ulp * 0.05).all()), (frac, d.max().item())


def test_fused_bottleneck_at_config3_size_many_tiles_per_workgroup():
    """64 x 72 x 320 is config 3's layer-1 map (11 520 tiles over 512 persistent workgroups); here 8 x 72 x 320 (1 440 tiles: ~3 per workgroup, both ring
    parities, the XCD walk) against the separate launches, and the plain tile walk bit-identical to the XCD-aware one."""
    from visualdet3d_amd import _lib, hip_ops as ops
    dt = torch.bfloat16
    for ds in (False, True):
        cin = 64 if ds else 256
        P, g = _make(cin, ds, seed=11)
        xd = (torch.randn(8, 72, 320, cin, generator=g).abs() * 0.8).cuda().to(dt)
        pc1, pc2, pc3, pcd = _packs(P, dt)
        got = ops.conv2d_bottleneck(xd, pc1, pc2, pc3, pcd)
        with _lib.test_switch('VD3D_PLAIN_TILE_WALK'):
            got_plain = ops.conv2d_bottleneck(xd, pc1, pc2, pc3, pcd)
        t2 = ops.conv2d(ops.conv2d(xd, pc1, relu=True), pc2, relu=True)
        sep = ops.conv2d(t2, pc3, residual=ops.conv2d(xd, pcd, relu=False) if ds else xd, relu=True)
        torch.cuda.synchronize()
        assert torch.equal(got, got_plain)
        d = (got.float() - sep.float()).abs()
        scale = sep.float().abs().max().item()
        assert float((d > 0).float().mean()) < 2e-2 and bool((d <= sep.float().abs() * 2.0 ** -6 + scale * 1e-3).all()), d.max().item()


def test_fused_bottleneck_at_the_full_config3_batch():
    """64 x 72 x 320 (config 3's stacked left / right batch: 755 MB of input, 11 520 tiles, 22.5 per workgroup): both variants against their separate launches --
    byte offsets up to 2^29.5, every workgroup through many tiles and both ring parities."""
    from visualdet3d_amd import hip_ops as ops
    dt = torch.bfloat16
    for ds in (False, True):
        cin = 64 if ds else 256
        P, g = _make(cin, ds, seed=13)
        xd = (torch.randn(4, 72, 320, cin, generator=g).abs() * 0.8).cuda().to(dt).repeat(16, 1, 1, 1)
        xd[1::2] *= 0.5                                   # (not 16 identical copies)
        pc1, pc2, pc3, pcd = _packs(P, dt)
        got = ops.conv2d_bottleneck(xd, pc1, pc2, pc3, pcd)
        t2 = ops.conv2d(ops.conv2d(xd, pc1, relu=True), pc2, relu=True)
        sep = ops.conv2d(t2, pc3, residual=ops.conv2d(xd, pcd, relu=False) if ds else xd, relu=True)
        torch.cuda.synchronize()
        d = (got.float() - sep.float()).abs()
        scale = sep.float().abs().max().item()
        assert float((d > 0).float().mean()) < 2e-2 and bool((d <= sep.float().abs() * 2.0 ** -6 + scale * 1e-3).all()), d.max().item()
        del got, t2, sep, d
        torch.cuda.empty_cache()


def test_unsupported_shapes_are_refused():
    from visualdet3d_amd import _lib, hip_ops as ops
    P, g = _make(256, False, 3)
    pc1, pc2, pc3, _ = _packs(P, torch.bfloat16)
    x = torch.zeros(1, 8, 16, 128, dtype=torch.bfloat16, device='cuda')
    assert not ops.conv2d_bottleneck_supported(x, pc1, pc2, pc3)
    with pytest.raises(AssertionError):
        ops.conv2d_bottleneck(x, pc1, pc2, pc3)
