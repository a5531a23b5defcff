"""Micro-benchmark of DLA level0 + level1 (BASELINE config 5: 16 x 512 x 1760 x 16 channels, fp16): the fused pair kernel
(vd3d_conv2d_pair) against the two small-channel launches.    python tools/bench_levelpair.py [reps]"""
import sys

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dt = torch.float16
x = torch.randn(16, 512, 1760, 16, device='cuda').to(dt)


def mk(cin, cout, stride):
    w = torch.randn(cout, cin, 3, 3, device='cuda') * (2.0 / (9 * cin)) ** 0.5
    bn = (torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda') * 0.1, torch.randn(cout, device='cuda') * 0.1,
          torch.rand(cout, device='cuda') + 0.5, 1e-5)
    return ops.pack_conv(w, None, bn, dt, stride, 1, 1)


pa, pb = mk(16, 16, 1), mk(16, 32, 2)
for label, fn in (('pair', lambda: ops.conv2d_pair(x, pa, pb)), ('two launches', lambda: ops.conv2d(ops.conv2d(x, pa, relu=True), pb, relu=True))):
    out = fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) * 1e-3 / reps
    by = (x.numel() + out.numel()) * 2
    print('%-14s %8.1f us  %6.2f TB/s (input + level1 output once)' % (label, t * 1e6, by / t / 1e12))
