"""Would fusing conv1 + conv2 of a ResNet layer-1 BasicBlock (64 -> 64, 3x3, backbones/resnet.py:23-52) pay?  
A fused kernel removes the intermediate's round trip (63 MB written + 63 MB read per block at 16 x 96 x 320) but re-computes conv1 on the
halo of every conv2 tile: (TH + 2)(TW + 2) / (TH TW) = 1.41x for 8 x 16 tiles, 1.27x for 16 x 16, 1.19x for 16 x 32 (the register-resident
weights of `conv_resident64` leave room for 8 x 16).  What bounds one conv today decides it: if the launch sat on the HBM roof the fused
block would win the bytes; this script measures the launch at the bench shape, with and without the residual read, in a loop whose whole
working set (126 / 189 MB) stays in the 256 MB Infinity Cache -- bytes per second and TF/s next to both roofs.
    python tools/bench_block_fusion_bound.py"""
import sys

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops  # noqa: E402

g = torch.Generator().manual_seed(0)
w = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5
bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5, 1e-5)
pc = ops.pack_conv(w.cuda(), None, tuple(t.cuda() if torch.is_tensor(t) else t for t in bn), torch.bfloat16, 1, 1, 1)
B, reps = 16, 40
x = torch.randn(B, 96, 320, 64, generator=g).cuda().to(torch.bfloat16)
res = torch.randn(B, 96, 320, 64, generator=g).cuda().to(torch.bfloat16)
out = torch.empty_like(x)
rows = {}
for rnd in range(2):
    for use_res in (False, True):
        for _ in range(5):
            ops.conv2d(x, pc, out=out, residual=res if use_res else None, relu=True)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps):
            ops.conv2d(x, pc, out=out, residual=res if use_res else None, relu=True)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) * 1e-3 / reps
        gf = 2.0 * B * 96 * 320 * 64 * 576 / 1e9
        mb = x.numel() * 2 * (3 if use_res else 2) / 1e6
        rows[use_res] = t
        print('64 -> 64 @ 16 x 96 x 320 %-9s %6.1f us  %6.1f TF/s (%.0f %% of 2.5 PF)  %5.0f MB = %.2f TB/s (%.0f %% of 6.3)'
              % ('+residual' if use_res else '', t * 1e6, gf / t / 1e3, gf / t / 1e3 / 25, mb, mb / t / 1e6, mb / t / 1e6 / 6.3 * 100))
today = rows[False] + rows[True]
print('block today: %.1f + %.1f = %.1f us (+ one launch gap)' % (rows[False] * 1e6, rows[True] * 1e6, today * 1e6))
print('neither roof binds one conv (MFMA pipe ~30 %% busy, < 3 TB/s): the launch is bound by its own per-tile pipeline (halo DMA -> 36 MFMAs per wave ->'
      ' epilogue), so a fused block costs that pipeline (1 + halo factor) times:')
for name, f in (('8 x 16', 1.41), ('16 x 16', 1.27), ('16 x 32', 1.19)):
    print('  fused, %-7s tiles: >= (1 + %.2f) x %.1f us = %.1f us  (today %.1f)' % (name, f, rows[False] * 1e6, (1 + f) * rows[False] * 1e6, today * 1e6))
