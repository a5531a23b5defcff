"""Tile sweep of the ResNet-50 bottleneck 1x1 convolutions (HBM-bound: K is one or a few 64-deep slices) on MI355X:
    python tools/bench_conv1x1.py [tile ids...]      (0 = the dispatcher's own pick)"""
import sys

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops, _lib  # noqa: E402

SHAPES = [  # name, B, H, W, Cin, Cout, residual
    ('64->256 +res @72x320', 64, 72, 320, 64, 256, True),
    ('64->256 @72x320', 64, 72, 320, 64, 256, False),
    ('256->64 @72x320', 64, 72, 320, 256, 64, False),
    ('64->64 @72x320', 64, 72, 320, 64, 64, False),
    ('128->512 +res @36x160', 64, 36, 160, 128, 512, True),
    ('512->128 @36x160', 64, 36, 160, 512, 128, False),
    ('256->1024 +res @18x80', 64, 18, 80, 256, 1024, True),
    ('1024->256 @18x80', 64, 18, 80, 1024, 256, False),
]
cfgs = [int(a) for a in sys.argv[1:]] or [0, 58, 42, 30]
torch.manual_seed(0)
for name, B, H, W, Cin, Cout, res in SHAPES:
    x = torch.randn(B, H, W, Cin, device='cuda').to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') * (2.0 / Cin) ** 0.5
    pc = ops.pack_conv(w, None, None, torch.bfloat16, 1, 0, 1)
    r = torch.randn(B, H, W, Cout, device='cuda').to(torch.bfloat16) if res else None
    by = (x.numel() + B * H * W * Cout * (2 if res else 1)) * 2
    line = '%-24s' % name
    ref = None
    for c in cfgs:
        _lib.lib().vd3d_test_force_conv_tile(c)
        try:
            out = ops.conv2d(x, pc, residual=r, relu=True)
            torch.cuda.synchronize()
        except Exception:
            line += '  cfg%-2d   ERR      ' % c
            continue
        if ref is None:
            ref = out.float()
        elif (out.float() - ref).abs().max().item() > 0.1:
            line += '  cfg%-2d  WRONG     ' % c
            continue
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            ops.conv2d(x, pc, residual=r, relu=True, out=out)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) * 1e-4
        line += '  cfg%-2d %6.1f us %4.2f TB/s' % (c, t * 1e6, by / t / 1e12)
    _lib.lib().vd3d_test_force_conv_tile(0)
    print(line, flush=True)
