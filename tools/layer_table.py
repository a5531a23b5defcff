"""Per-launch table of vd3d_conv2d_igemm calls (shape, time, TF/s, GB/s) for one of the tools/bench_configs.py models:
    python tools/layer_table.py "C5 KM3D"      (serial launches, HIP events, side-stream overlaps off)"""
import sys

import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import bench_configs as bc  # noqa: E402
from visualdet3d_amd import hip_ops as ops  # noqa: E402

name = [n for n in bc.CONFIGS if sys.argv[1] in n][0]
m, inputs = bc.build(bc.CONFIGS[name])
for attr in ('bbox_head',):
    if hasattr(getattr(m, attr, None), 'overlap_towers'):
        m.bbox_head.overlap_towers = False
if hasattr(m.core, 'overlap_neck'):
    m.core.overlap_neck = False
rec = []
orig = ops.conv2d
orig_pair = ops.conv2d_pair
orig_bn = ops.conv2d_bottleneck


def timed(x, pc, out=None, residual=None, relu=False, out_f32=False):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    o = orig(x, pc, out=out, residual=residual, relu=relu, out_f32=out_f32)
    e.record()
    B, Ho, Wo, Co = o.shape
    fl = 2.0 * B * Ho * Wo * Co * pc.kh * pc.kw * pc.Cin
    by = x.shape[0] * x.shape[1] * x.shape[2] * pc.Cin * x.element_size() + o.numel() * o.element_size() + (residual.numel() * residual.element_size() if residual is not None else 0)
    rec.append(('%dx%d s%d %4d->%4d @ %dx%dx%d%s' % (pc.kh, pc.kw, pc.stride, pc.Cin, pc.Cout, B, Ho, Wo, ' f32' if out_f32 else ''), fl, by, s, e))
    return o


def timed_pair(x, pa, pb, relu_a=True, relu_b=True):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    o = orig_pair(x, pa, pb, relu_a=relu_a, relu_b=relu_b)
    e.record()
    B, H, W, _ = x.shape
    _, Ho, Wo, Co = o.shape
    fl = 2.0 * B * 9 * (H * W * pa.Cout * pa.Cin + Ho * Wo * Co * pb.Cin)
    rec.append(('3x3 s1 %d->%d + 3x3 s2 %d->%d @ %dx%dx%d' % (pa.Cin, pa.Cout, pb.Cin, Co, B, H, W), fl, x.numel() * x.element_size() + o.numel() * o.element_size(), s, e))
    return o


def timed_bottleneck(x, pc1, pc2, pc3, pc_ds=None, out=None):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    o = orig_bn(x, pc1, pc2, pc3, pc_ds, out=out)
    e.record()
    B, H, W, Cx = x.shape
    fl = 2.0 * B * H * W * (Cx * 64 + 576 * 64 + 64 * 256 + (Cx * 256 if pc_ds is not None else 0))
    rec.append(('bottleneck %d->64->64->256%s @ %dx%dx%d (one launch)' % (Cx, ' + ds' if pc_ds is not None else '', B, H, W), fl,
                x.numel() * x.element_size() + o.numel() * o.element_size(), s, e))
    return o


with torch.no_grad():
    m.forward_device(*inputs)
    torch.cuda.synchronize()
    ops.conv2d = timed
    ops.conv2d_pair = timed_pair
    ops.conv2d_bottleneck = timed_bottleneck
    rec.clear()
    m.forward_device(*inputs)
    torch.cuda.synchronize()
ops.conv2d = orig
ops.conv2d_pair = orig_pair
ops.conv2d_bottleneck = orig_bn
tot = 0.0
for d, fl, by, s, e in rec:
    t = s.elapsed_time(e) * 1e-3
    tot += t
    print('%-40s %8.2f GF %8.1f us %7.1f TF/s %7.1f GB/s' % (d, fl / 1e9, t * 1e6, fl / t / 1e12, by / t / 1e9))
print('total conv2d time %.2f ms over %d launches' % (tot * 1e3, len(rec)))
