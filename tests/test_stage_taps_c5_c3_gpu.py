"""GPU: BASELINE configs 5 and 3 AT SIZE in their TIMED dtypes, TEACHER-FORCED stage by stage (what tests/test_stage_taps_gpu.py does
for config 2).

  * C5 = KM3D DLA-34, 512 x 1760, fp16, batch 16 (bench.py `other_configs[C5]`: same cfg, weights and images);
  * C3 = YOLOStereo3D ResNet-50 core + base DCNv2 head, 288 x 1280, bf16, batch 32 (`other_configs[C3]`).

The 16-bit-rounded oracle (oracle/detector_oracle.py, pinned to the reference by tests/golden) runs TWO frames on the host and records
every fused operation with its input and output.  Each HIP stage is then launched AT THE BENCH'S BATCH -- the oracle's input of that stage
repeated to 16 / 32 frames, so that the dispatcher picks the at-size kernels and their at-size paths (multi-tile persistent loops,
`ConvArgs::group_m`, the 1.8 GB `dcn_columns` matrix + the 19 584-deep GEMM, `conv_pw` with residual, the level pair, the IDA-Up phase
kernel, the persistent fused head) -- and EVERY replica of the output is compared with the oracle's output of the stage:

    |hip - oracle| <= 2 ulp of the format (fp16: 2 * 2^-10 |oracle|, bf16: 2 * 2^-7 |oracle|)  +  3e-5 of the stage's output scale

(fp32-output convs: 1e-4 of the output scale; DCNv2 blocks: 2 ulp + 5e-4 of the scale -- the sampling positions come from an fp32
conv whose summation order differs; the fused KM3D head, two chained GEMMs with a 16-bit intermediate, and the fused cost volume: relative
to the output scale).  Reference: backbones/dla.py:317-326, dla_utils.py:42-155, heads/km3d_head.py:132-153,353-357,
heads/detection_3d_head.py:47-88, backbones/resnet.py:55-91."""
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu

ULPS = 2.0
SUM_TERM = 3e-5
ULP = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}


def _rep_nhwc(t, dtype, rep):
    x = t.cuda().permute(0, 2, 3, 1).contiguous().to(dtype)
    return x.repeat(rep, 1, 1, 1) if rep > 1 else x


def _score(got_nhwc, want_nchw, rep, dtype, out_round=True, sum_term=SUM_TERM):
    """got: [rep * n, H, W, C] on the GPU, replica-major; want: [n, C, H, W] on the host.  -> (error in units of the bar, in single
    ulps, relative to the output scale), worst over every replica."""
    w = want_nchw.cuda().permute(0, 2, 3, 1).contiguous().float()
    n = w.shape[0]
    assert got_nhwc.shape[0] == rep * n and tuple(got_nhwc.shape[1:]) == tuple(w.shape[1:]), (tuple(got_nhwc.shape), tuple(w.shape), rep)
    sc = max(w.abs().max().item(), 1e-30)
    worst_bar = worst_ulp = worst_rel = 0.0
    for r in range(rep):                                      # replica by replica: bounded temporaries at 16 x 512 x 1760
        d = (got_nhwc[r * n:(r + 1) * n].float() - w).abs()
        worst_rel = max(worst_rel, d.max().item() / sc)
        if out_round:
            worst_bar = max(worst_bar, (d / (w.abs() * (ULPS * ULP[dtype]) + sum_term * sc)).max().item())
            worst_ulp = max(worst_ulp, (d / (w.abs() * ULP[dtype] + sum_term * sc)).max().item())
    if not out_round:
        return worst_rel / 1e-4, 0.0, worst_rel
    return worst_bar, worst_ulp, worst_rel


def _conv_stage(t, mods, dt, rep):
    from visualdet3d_amd import hip_ops as ops
    from visualdet3d_amd.networks.lib import fused
    conv = mods[t['key']]
    bn = mods[t['bn']] if t['bn'] else None
    pc = ops.pack_conv(conv.weight, conv.bias, fused.bn_tuple(bn) if bn is not None else None, dt, conv.stride[0], conv.padding[0], conv.dilation[0])
    res = _rep_nhwc(t['residual'], dt, rep) if t['residual'] is not None else None
    return ops.conv2d(_rep_nhwc(t['x'], dt, rep), pc, residual=res, relu=t['relu'], out_f32=not t['out_round'])


def _dcn_stage(mod, bn, t, dt, rep):
    """DCNv2 + BN + ReLU block at size from the oracle's input -> (block output, the block's own fp32 offset logits)."""
    from visualdet3d_amd import hip_ops as ops
    x = _rep_nhwc(t['x'], dt, rep)
    got = mod.forward_nhwc(x, bn=bn, relu=True)
    conv = mod.conv_offset
    pco = mod._cache.get(('off', dt), [conv.weight, conv.bias], None)        # packed by the forward above
    logits = ops.conv2d(x, pco, relu=False, out_f32=True)[..., :t['logits'].shape[1]]
    return got, logits


class _Report:
    def __init__(self, title):
        self.title, self.lines, self.failures, self.kinds = title, [], [], {}

    def add(self, kind, key, shape, bar, ulps, rel, limit=1.0):
        self.kinds[kind] = self.kinds.get(kind, 0) + 1
        line = '%-11s %-52s %-22s %5.2f ulp  rel %.2e' % (kind, key, tuple(shape), ulps, rel)
        self.lines.append(line)
        if not bar <= limit:
            self.failures.append(line + '   (%.2f x bar)' % bar)

    def finish(self):
        print('\n[%s: %d stages]\n' % (self.title, len(self.lines)) + '\n'.join(self.lines))
        assert not self.failures, 'stages outside their bar:\n' + '\n'.join(self.failures)


def test_config5_km3d_fp16_batch16_every_stage_teacher_forced():
    from oracle import detector_oracle as orc
    from visualdet3d_amd import hip_ops as ops
    from visualdet3d_amd.networks.lib import fused
    from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT
    import visualdet3d_amd.networks.detectors  # noqa: F401
    from visualdet3d_amd.utils import synthetic as syn
    dt, N, REP, H, W = torch.float16, 2, 8, 512, 1760
    cfg = syn.km3d_cfg(output_w=W // 4)
    m = DETECTOR_DICT[cfg.name](cfg)
    sd = syn.seeded_state_dict(m.state_dict(), seed=1, head_std=0.0005)           # bench.py time_other_config(C5)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = dt
    img = syn.mono_image(N, H, W, seed=3)
    P2, _ = syn.kitti_calib(W, batch=N)
    torch.set_num_threads(min(64, torch.get_num_threads()))
    taps = []
    with torch.no_grad():
        orc.km3d_forward(sd, cfg, img, P2, rnd=orc.fp16_round, stage_taps=taps)
    mods = dict(m.named_modules())
    rep = _Report('C5 KM3D fp16 16 x 512 x 1760 teacher-forced')
    first = {}
    with torch.no_grad():
        for t in taps:
            kind, key = t['kind'], t['key']
            if kind == 'conv' and key.startswith('bbox_head.head_layers.'):
                continue                                        # the nine branches run as ONE fused launch: the km3d_head record below
            if kind == 'conv' and key.endswith('base_layer.0'):
                conv, bn = mods[key], mods[t['bn']]
                pc = ops.pack_image_conv(conv.weight, fused.bn_tuple(bn), dt, 1, 3)
                got = ops.image_conv(t['x'].repeat(REP, 1, 1, 1).contiguous().cuda(), pc, relu=True)    # fp32 NCHW image, as the detector hands it over
                rep.add('image_conv', key, t['y'].shape, *_score(got, t['y'], REP, dt))
            elif kind == 'conv':
                got = _conv_stage(t, mods, dt, REP)
                rep.add(kind, key, t['y'].shape, *_score(got, t['y'], REP, dt, t['out_round']))
                if key.endswith('level0.0') or key.endswith('level1.0'):
                    first[key.split('.')[-2]] = t
            elif kind == 'maxpool':
                got = ops.maxpool2x2(_rep_nhwc(t['x'], dt, REP))
                bar, ulps, rel = _score(got, t['y'], REP, dt)
                rep.add(kind, key, t['y'].shape, 0.0 if rel == 0.0 else 2.0, ulps, rel)          # a pure selection: bit-exact
            elif kind == 'dcn':
                blk = mods[key]                                  # dla_utils.DeformConv: .conv (ModulatedDeformConvPack), .actf[0] (BN)
                got, logits = _dcn_stage(blk.conv, blk.actf[0], t, dt, REP)
                rep.add(kind, key, t['y'].shape, *_score(got, t['y'], REP, dt, sum_term=5e-4))
                rep.add('dcn_offset', key + '.conv.conv_offset', t['logits'].shape, *_score(logits, t['logits'], REP, dt, out_round=False))
            elif kind == 'dwconvT':
                up = mods[key]
                w = up.weight.detach().float().reshape(up.weight.shape[0], -1).t().contiguous()
                got = ops.dwconv_transpose(_rep_nhwc(t['x'], dt, REP), w, t['f'], add=_rep_nhwc(t['add'], dt, REP))
                rep.add(kind, key, t['y'].shape, *_score(got, t['y'], REP, dt))
            elif kind == 'km3d_head':
                head = mods['bbox_head']
                assert head.fuse_head
                maps = head.forward_nhwc(_rep_nhwc(t['x'], dt, REP))          # the persistent fused launch (3x3 convs + ReLU + 1x1 convs)
                for name, want in t['y'].items():
                    _, _, rel = _score(maps[name], want, REP, dt, out_round=False)
                    # two chained GEMMs with an fp16 intermediate: a 1-ulp flip of a hidden value moves the fp32 output by ~2^-11 of one of
                    # 256 products -- held to 1e-3 of the map's scale (measured below)
                    rep.add(kind, key + '.' + name, want.shape, rel / 1e-3, 0.0, rel)
            else:
                raise AssertionError('unknown tap kind ' + kind)
            torch.cuda.synchronize()
        # level0 -> level1 as ONE launch (what the detector runs): bit-identical to the two teacher-forced launches chained
        t0, t1 = first['level0'], first['level1']
        bb = mods['core.backbone']
        pa = ops.pack_conv(bb.level0[0].weight, None, fused.bn_tuple(bb.level0[1]), dt, 1, 1, 1)
        pb = ops.pack_conv(bb.level1[0].weight, None, fused.bn_tuple(bb.level1[1]), dt, 2, 1, 1)
        assert ops.conv2d_pair_supported(pa, pb)
        x0 = _rep_nhwc(t0['x'], dt, REP)
        pair = ops.conv2d_pair(x0, pa, pb)
        chain = ops.conv2d(ops.conv2d(x0, pa, relu=True), pb, relu=True)
        assert torch.equal(pair.view(torch.int16), chain.view(torch.int16)), 'level pair differs from its two launches at 16 x 512 x 1760'
    # coverage of the path: base layer, level0 / level1, 36 tree convs (project / block / root), 5 pools... (DLA-34), 16 DCNv2 blocks and
    # their offset convs, 8 IDA-Up transposed convs, the nine head maps
    k = rep.kinds
    assert k.get('image_conv') == 1 and k.get('conv') == 38 and k.get('maxpool') == 6 and k.get('dcn') == 16 and k.get('dcn_offset') == 16, k
    assert k.get('dwconvT') == 8 and k.get('km3d_head') == 9, k
    rep.finish()


def test_config3_r50_dcn_head_bf16_batch32_every_stage_teacher_forced():
    from oracle import detector_oracle as orc
    from tests.test_stage_taps_gpu import _run_stage as _c2_stage
    from visualdet3d_amd import hip_ops as ops
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3DBaseHead
    from visualdet3d_amd.networks.lib import fused
    from visualdet3d_amd.utils import synthetic as syn
    dt, N, REP, H, W = torch.bfloat16, 2, 16, 288, 1280
    tmp = tempfile.mkdtemp()
    cfg = syn.stereo3d_cfg(tmp, depth=50, score_thr=0.5, nms_iou_thr=0.4)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    m = Stereo3DBaseHead(cfg)
    sd = syn.seeded_state_dict(m.state_dict(), seed=6, head_std=0.006)           # bench.py OTHER_CONFIGS[C3]
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = dt
    L, R = syn.stereo_pair(N, H, W, seed=3)
    torch.set_num_threads(min(64, torch.get_num_threads()))
    taps = []
    with torch.no_grad():
        c = orc.Ctx(sd, orc.bf16_round, taps)
        feats, _ = orc.stereo_core(c, L, R, 50)
        orc.dcn_head(c, feats, len(cfg.obj_types) + 1)
    mods = dict(m.named_modules())
    rep = _Report('C3 R50 + DCNv2 head bf16 32 x 288 x 1280 teacher-forced')

    def rp(x):                                                    # NCHW host tensor of N (or 2N) frames -> replica-major batch
        return x.repeat(REP, *([1] * (x.dim() - 1)))

    with torch.no_grad():
        for t in taps:
            kind, key = t['kind'], t['key']
            if kind == 'conv':
                got = _conv_stage(t, mods, dt, REP)
                rep.add(kind, key, t['y'].shape, *_score(got, t['y'], REP, dt, t['out_round']))
            elif kind == 'dcn_head':
                # 2176 -> 2176 DCNv2: `dcn_columns` (1.8 GB of bf16 columns at this batch) + the 19 584-deep 1x1 GEMM in grouped tile order
                mod = mods[key]
                assert mod.out_channels > mod.columns_above
                got, logits = _dcn_stage(mod, mods[t['bn']], t, dt, REP)
                rep.add(kind, key, t['y'].shape, *_score(got, t['y'], REP, dt, sum_term=5e-4))
                rep.add('dcn_offset', key + '.conv_offset', t['logits'].shape, *_score(logits, t['logits'], REP, dt, out_round=False))
            elif kind == 'stem':
                bb = mods[key]
                n = t['x'].shape[0] // 2
                pc = ops.pack_stem_conv(bb.conv1.weight, fused.bn_tuple(bb.bn1), dt)
                imgs = [rp(t['x'][:n]).contiguous().cuda(), rp(t['x'][n:]).contiguous().cuda()]
                assert ops.stem_pool_supported(H, W, dt, pc.Cout)
                out = ops.stem_conv_pool(imgs, pc, dt)                                   # [left replicas | right replicas]
                half = REP * n
                got = torch.cat([out[:half].reshape(REP, n, *out.shape[1:]), out[half:].reshape(REP, n, *out.shape[1:])], dim=1).reshape(out.shape)
                rep.add(kind, key, t['y'].shape, *_score(got, t['y'], REP, dt))
            else:
                # the stereo neck's own kernels (cosine volumes, cost volume, ghost depth-wise convs, pools): the config-2 runner on the
                # replicated record
                tt = dict(t)
                for f in ('x', 'left', 'right', 'residual', 'y'):
                    if tt.get(f) is not None:
                        tt[f] = rp(tt[f])
                got, want = _c2_stage(tt, mods, N * REP)                                 # NCHW (or volume) fp32 on the host, replica-major
                assert got.shape == want.shape, (kind, key, got.shape, want.shape)
                sc = want.abs().max().item()
                d = (got - want).abs()
                rel = d.max().item() / max(sc, 1e-30)
                if kind == 'costvol_build':
                    assert torch.equal(got, want)
                bar = rel / 1e-2 if kind == 'cost_volume' else (d / (want.abs() * (ULPS * ULP[dt]) + SUM_TERM * sc)).max().item()
                rep.add(kind, key, t['y'].shape, bar, (d / (want.abs() * ULP[dt] + SUM_TERM * sc)).max().item(), rel)
            torch.cuda.synchronize()
    k = rep.kinds
    assert k.get('stem') == 1 and k.get('conv') == 58 and k.get('dcn_head') == 1 and k.get('dcn_offset') == 1, k
    assert k.get('psm_cosine') == 2 and k.get('cost_volume') == 1 and k.get('conv3d') == 2 and k.get('dwconv') == 3 and k.get('avgpool') == 2, k
    rep.finish()
