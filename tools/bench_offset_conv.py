"""Micro-benchmark of the DCN offset convs (3x3 / s1 / p1, Cin -> 27 padded to 32, fp32 logits) at the shapes of BASELINE configs 5
(KM3D DLA-34 at 16 x 512 x 1760, fp16) and 3 (stereo base head, 2176 channels at 32 x 18 x 80): the narrow-output streaming kernel
(tile id 70, the dispatch's choice) against the 256 x 32 tile kernel it replaced (forced tile 87).
    python tools/bench_offset_conv.py [reps]"""
import sys

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import _lib, hip_ops as ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [('C5 s8  16x64x220x128', 16, 64, 220, 128, torch.float16), ('C5 s16 16x32x110x256', 16, 32, 110, 256, torch.float16),
          ('C5 s32 16x16x55x512', 16, 16, 55, 512, torch.float16), ('C3 32x18x80x2176', 32, 18, 80, 2176, torch.bfloat16)]
for name, B, H, W, C, dt in SHAPES:
    x = torch.randn(B, H, W, C, device='cuda').to(dt)
    w = torch.randn(32, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    w[27:] = 0
    pc = ops.pack_conv(w, torch.zeros(32, device='cuda'), None, dt, 1, 1, 1)
    out = torch.empty(B, H, W, 32, device='cuda', dtype=torch.float32)
    ref = None
    for label, cfg in (('narrow 70', 70), ('tile 87', 87), ('natural', 0)):
        _lib.lib().vd3d_test_force_conv_tile(cfg)
        ops.conv2d(x, pc, out=out, relu=False, out_f32=True)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            ops.conv2d(x, pc, out=out, relu=False, out_f32=True)
        e.record()
        torch.cuda.synchronize()
        _lib.lib().vd3d_test_force_conv_tile(0)
        t = s.elapsed_time(e) * 1e-3 / reps
        by = x.numel() * 2 + out.numel() * 4
        print('%-22s %-10s %8.1f us  %6.2f TB/s (in + out once)  %6.1f TF/s   max diff vs narrow %.1e' % (
            name, label, t * 1e6, by / t / 1e12, 2.0 * B * H * W * 32 * 9 * C / t / 1e12, err))
