"""Where does a 16-bit mode leave the fp32 mode?  Runs one detector forward per compute dtype with every hip_ops entry that produces
an activation hooked, and prints, call by call, the output's |max| and its difference from the fp32 run (relative to the fp32
output's scale).  The first call whose error jumps names the kernel / shape to look at.
    python tools/diag_dtype_divergence.py km3d 512 1760 [fp16 bf16]"""
import sys

import torch

sys.path.insert(0, '.')
from tests.common import load_golden  # noqa: E402
from visualdet3d_amd import hip_ops as ops  # noqa: E402
from visualdet3d_amd.utils import synthetic as syn  # noqa: E402

HOOKED = ['conv2d', 'image_conv', 'maxpool2x2', 'dwconv_transpose', 'deform_conv_general', 'stem_conv_pool', 'stem_conv', 'maxpool3x3s2',
          'avgpool2x2', 'dwconv3x3', 'psm_cosine', 'conv3d_3x3x3', 'km3d_head_fused']


def run(model, inputs, dtype):
    rec = []
    orig = {n: getattr(ops, n) for n in HOOKED}

    def wrap(n):
        def f(*a, **k):
            o = orig[n](*a, **k)
            outs = o if isinstance(o, (list, tuple)) else [o]
            for t in outs:
                rec.append((n, tuple(t.shape), t.float().clone()))
            return o
        return f
    for n in HOOKED:
        setattr(ops, n, wrap(n))
    try:
        model.compute_dtype = dtype
        with torch.no_grad():
            model.forward_device(*inputs)
        torch.cuda.synchronize()
    finally:
        for n in HOOKED:
            setattr(ops, n, orig[n])
    return rec


def main():
    kind, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dts = [dict(fp16=torch.float16, bf16=torch.bfloat16)[d] for d in (sys.argv[4:] or ['fp16', 'bf16'])]
    assert kind == 'km3d'
    from visualdet3d_amd.networks.detectors import KM3D
    cfg = syn.km3d_cfg(output_w=W // 4)
    m = KM3D(cfg)
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), seed=7))
    m = m.cuda().eval()
    m.bbox_head.fuse_head = False           # same launch sequence in every dtype (the fused head is compared separately below)
    img = syn.mono_image(1, H, W, seed=13).cuda()
    P2, _ = syn.kitti_calib(W, batch=1)
    inputs = (img, P2.cuda())
    ref = run(m, inputs, torch.float32)
    for dt in dts:
        got = run(m, inputs, dt)
        print('==== %s vs fp32: %d / %d calls' % (dt, len(got), len(ref)))
        j = 0
        for i, (n, shp, t) in enumerate(got):
            while j < len(ref) and (ref[j][0] != n or ref[j][1] != shp):
                j += 1
            if j >= len(ref):
                print('%3d %-20s %-24s |max| %10.3e   (no fp32 counterpart)' % (i, n, shp, t.abs().max().item()))
                j = 0
                continue
            r = ref[j][2]
            j += 1
            e = ((t - r).abs().max() / r.abs().max().clamp_min(1e-30)).item()
            flag = '  <-----' if e > 3e-2 else ''
            print('%3d %-20s %-24s |max| %10.3e  fp32 |max| %10.3e  rel err %.2e  nonfinite %d%s'
                  % (i, n, shp, t.abs().max().item(), r.abs().max().item(), e, int((~torch.isfinite(t)).sum()), flag))
        m.bbox_head.fuse_head = True
        fused = run(m, inputs, dt)
        m.bbox_head.fuse_head = False
        heads_f = [t for n, s, t in fused if n == 'km3d_head_fused']
        heads_u = [t for n, s, t in got[-len(heads_f):]] if heads_f else []
        for a, b in zip(heads_f, heads_u):
            print('    fused head map %-22s vs unfused: rel %.2e' % (tuple(a.shape), ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()))


if __name__ == '__main__':
    main()
