"""MI355X: the fused 64-wide Bottleneck (vd3d_conv2d_bottleneck) against its separate launches at BASELINE config 3's layer-1 size
(64 x 72 x 320), and at config 2-like sizes:  python tools/bench_bottleneck.py [B H W]"""
import sys

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops  # noqa: E402

B, H, W = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (64, 72, 320)
dt = torch.bfloat16
torch.manual_seed(0)


def pack(o, i, k, relu_bn=True):
    w = torch.randn(o, i, k, k, device='cuda') * (2.0 / (i * k * k)) ** 0.5
    bn = (torch.rand(o, device='cuda') * 0.4 + 0.8, torch.randn(o, device='cuda') * 0.05, torch.randn(o, device='cuda') * 0.1, torch.rand(o, device='cuda') + 0.5, 1e-5)
    return ops.pack_conv(w, None, bn, dt, 1, k // 2, 1)


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for ds in (False, True):
    cin = 64 if ds else 256
    x = (torch.randn(B, H, W, cin, device='cuda').abs() * 0.8).to(dt)
    pc1, pc2, pc3 = pack(64, cin, 1), pack(64, 64, 3), pack(256, 64, 1)
    pcd = pack(256, cin, 1) if ds else None
    out = torch.empty(B, H, W, 256, dtype=dt, device='cuda')

    def separate():
        t2 = ops.conv2d(ops.conv2d(x, pc1, relu=True), pc2, relu=True)
        res = ops.conv2d(x, pcd, relu=False) if ds else x
        return ops.conv2d(t2, pc3, residual=res, relu=True, out=out)

    def fused():
        return ops.conv2d_bottleneck(x, pc1, pc2, pc3, pcd, out=out)

    a = separate().clone()
    b = fused().clone()
    torch.cuda.synchronize()
    diff = (a.float() - b.float()).abs()
    M = B * H * W
    gf = 2.0 * M * (cin * 64 + 576 * 64 + 64 * 256 + (cin * 256 if ds else 0)) / 1e9
    by = M * (cin + 256) * 2
    ts, tf = timeit(separate), timeit(fused)
    print('%s block @ %dx%dx%d: separate %7.1f us | fused %7.1f us = %5.1f TF/s, %4.2f TB/s algorithmic (x once + out once)  | differing elements %.2e, max %.3g'
          % ('first (64 in + downsample)' if ds else 'identity (256 in)', B, H, W, ts, tf, gf / tf * 1e3, by / tf * 1e-6, float((diff > 0).float().mean()), diff.max().item()), flush=True)
