"""GPU: BASELINE config 2 AT SIZE (8 pairs of 384 x 1280, bf16 -- what bench.py times), TEACHER-FORCED stage by stage.

The bf16-rounded oracle (oracle/detector_oracle.py, pinned to the reference by tests/golden) records every fused operation of the
path with its input and output (``stage_taps``).  Each HIP stage is then fed the ORACLE's input of that stage -- not the HIP
path's own upstream result -- under natural dispatch at the bench's own shapes, and compared with the oracle's output of that
stage.  A 1-ulp flip can therefore not cascade through the ~60 layers, and every stage is held to

    |hip - oracle| <= 2 bf16 ulp (2 * 2^-7 |oracle|)  +  3e-5 of the stage's output scale (fp32 summation order of a K-long dot product)

(fp32-output head convs: 1e-4 of the output scale).  This replaces the loose end-to-end bars (3e-2 on logits) as the statement of
per-stage correctness in the timed dtype: stem + pool, all 26 + 2 backbone convs (layer1 / 2 / 3 on the stacked L | R batch), both PSM
cosine volumes at 8 x 96 x 320 and 8 x 48 x 160, the concat volume + both 3-D convs, every ghost / pyramid block (primary conv,
depth-wise conv, average pool, BasicBlock convs) and every conv of the cls and reg towers."""
import pytest
import torch

from tests.conftest import c2_bf16_case

pytestmark = pytest.mark.gpu

ULPS = 2.0
BF16_ULP = 2.0 ** -7
SUM_TERM = 3e-5


def _nhwc(x, dtype=torch.bfloat16):
    return x.permute(0, 2, 3, 1).contiguous().cuda().to(dtype)


def _nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2)


def _score(got, want, out_round=True):
    """-> (error in units of the bar (<= 1 passes), max ulp-normalised error, error relative to the output scale)."""
    sc = want.abs().max().item()
    d = (got - want).abs()
    rel = d.max().item() / max(sc, 1e-30)
    if not out_round:                                        # fp32 epilogue from bf16 operands: summation noise only
        return rel / 1e-4, 0.0, rel
    bar = (d / (want.abs() * (ULPS * BF16_ULP) + SUM_TERM * sc)).max().item()
    ulps = (d / (want.abs() * BF16_ULP + SUM_TERM * sc)).max().item()
    return bar, ulps, rel


def _run_stage(t, mods, B):
    """One oracle tap record -> (got NCHW fp32 on the host, want)."""
    from visualdet3d_amd import hip_ops as ops
    from visualdet3d_amd.networks.lib import fused
    kind, key = t['kind'], t['key']
    dt = torch.bfloat16
    if kind == 'conv':
        conv = mods[key]
        bn = mods[t['bn']] if t['bn'] else None
        pc = ops.pack_conv(conv.weight, conv.bias, fused.bn_tuple(bn) if bn is not None else None, dt, conv.stride[0], conv.padding[0], conv.dilation[0])
        res = _nhwc(t['residual']) if t['residual'] is not None else None
        return _nchw(ops.conv2d(_nhwc(t['x']), pc, residual=res, relu=t['relu'], out_f32=not t['out_round'])), t['y']
    if kind == 'dwconv':
        dw, bn = mods[key], mods[t['bn']]
        return _nchw(ops.dwconv3x3(_nhwc(t['x']), ops.pack_dwconv(dw.weight, fused.bn_tuple(bn)), relu=t['relu'])), t['y']
    if kind == 'stem':
        bb = mods[key]
        n = t['x'].shape[0] // 2
        pc = ops.pack_stem_conv(bb.conv1.weight, fused.bn_tuple(bb.bn1), dt)
        imgs = [t['x'][:n].contiguous().cuda(), t['x'][n:].contiguous().cuda()]       # left | right, stacked by the stem's image pack
        assert ops.stem_pool_supported(t['x'].shape[2], t['x'].shape[3], dt, pc.Cout)
        return _nchw(ops.stem_conv_pool(imgs, pc, dt)), t['y']
    if kind == 'psm_cosine':
        return _nchw(ops.psm_cosine(_nhwc(t['left']), _nhwc(t['right']), t['y'].shape[1])), t['y']
    if kind == 'avgpool':
        return _nchw(ops.avgpool2x2(_nhwc(t['x']))), t['y']
    if kind == 'costvol_build':
        vol = ops.costvol_build(_nhwc(t['left']), _nhwc(t['right']), t['y'].shape[2])             # [B, D, H, W, 2F]
        return vol.float().cpu().permute(0, 4, 1, 2, 3), t['y']
    if kind == 'conv3d':
        c3, b3 = mods[key], mods[key[:-1] + str(int(key[-1]) + 1)]
        p3 = ops.pack_conv3d(c3.weight, c3.bias, fused.bn_tuple(b3))
        x = t['x'].permute(0, 2, 3, 4, 1).contiguous().cuda().to(dt)                              # [B, D, H, W, Cin]
        if not t['last']:
            return ops.conv3d_3x3x3(x, p3, relu=True).float().cpu().permute(0, 4, 1, 2, 3), t['y']
        Bv, Fo, D, H, W = t['y'].shape
        out = torch.empty((Bv, H, W, Fo * D), dtype=dt, device='cuda')
        ops.conv3d_3x3x3(x, p3, relu=True, out_nhwc=out)                                          # channel = f * D + d
        return _nchw(out), t['y'].reshape(Bv, Fo * D, H, W)
    if kind == 'cost_volume':
        # the product module's fused launch (concat volume + 2 x Conv3d + BN3d + ReLU + reshape) from the down-sampled features
        cv = mods[key]
        c0, b0, c1, b1 = cv.conv3d[0], cv.conv3d[1], cv.conv3d[3], cv.conv3d[4]
        p0 = ops.pack_conv3d(c0.weight, c0.bias, fused.bn_tuple(b0))
        p1 = ops.pack_conv3d(c1.weight, c1.bias, fused.bn_tuple(b1))
        Bv, _, H, W = t['y'].shape
        out = torch.empty((Bv, H, W, cv.output_channel), dtype=dt, device='cuda')
        l, r = _nhwc(t['left']), _nhwc(t['right'])
        assert cv.fuse_volume and ops.cost_volume_fused_supported(l, cv.depth_channel)
        ops.cost_volume_fused(l, r, p0, p1, cv.depth_channel, out=out)       # what the bench runs: ONE launch, volume never in HBM
        return _nchw(out), t['y']
    raise AssertionError('unknown tap kind ' + kind)


def test_config2_batch8_bf16_every_stage_teacher_forced():
    case = c2_bf16_case()
    m, taps, B = case['model'], case['taps'], case['B']
    mods = dict(m.named_modules())
    kinds = {}
    report, failures = [], []
    with torch.no_grad():
        for t in taps:
            got, want = _run_stage(t, mods, B)
            torch.cuda.synchronize()
            assert got.shape == want.shape, (t['kind'], t['key'], got.shape, want.shape)
            bar, ulps, rel = _score(got, want, t.get('out_round', True))
            if t['kind'] == 'costvol_build':
                assert torch.equal(got, want), 'concat volume is a pure copy: must be bit-exact'
            limit = 1.0
            if t['kind'] == 'cost_volume':
                # the module's fused launch = concat volume -> conv3d -> conv3d CHAINED: a 1-ulp flip of the bf16 intermediate
                # volume feeds 216 products of the second conv, so small outputs move by many of THEIR ulps (measured 12) while the
                # error stays ~1 ulp of the output SCALE (measured 5.2e-3): held to 1e-2 of the scale; each of the two convs is
                # held to the ulp bar on its own by the `conv3d` records above (the unfused kernels, same MFMA formulation)
                bar = rel / 1e-2
            kinds[t['kind']] = kinds.get(t['kind'], 0) + 1
            line = '%-13s %-58s %-24s %5.2f ulp  rel %.2e' % (t['kind'], t['key'], tuple(want.shape), ulps, rel)
            report.append(line)
            if not bar <= limit:
                failures.append(line)
    print('\n[C2 B=8 bf16 teacher-forced stages: %d]\n' % len(taps) + '\n'.join(report))
    # the tap list must cover the whole path: stem, 2 x 13 backbone convs + 2 down-sample convs, the cost volume's 1x1 conv (L and R), 2 cosine volumes, the CostVolume
    # module, 3 ghost modules (primary + depth-wise), 2 pools, 3 pyramid BasicBlocks, 3 cls + 4 reg tower convs
    assert kinds.get('stem') == 1 and kinds.get('psm_cosine') == 2 and kinds.get('conv3d') == 2 and kinds.get('dwconv') == 3
    assert kinds.get('avgpool') == 2 and kinds.get('cost_volume') == 1 and kinds.get('conv') == 46
    assert not failures, 'stages outside 2 bf16 ulp + 3e-5 x scale:\n' + '\n'.join(failures)
