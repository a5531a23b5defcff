"""Micro-benchmark of the fused DCNv2 kernel (vd3d_deform_conv, NHWC fast path) on the KM3D up-path shapes (MI355X):
    python tools/bench_dcn.py [fp16|bf16] [reps]
Prints per-launch time (HIP events around `reps` back-to-back launches), TF/s and the algorithmic GB/s."""
import os
import sys

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops  # noqa: E402

SHAPES = [  # name, B, H, W, Cin, Cout
    ('ida 64->64 @128x440', 16, 128, 440, 64, 64),
    ('ida 128->64 @64x220', 16, 64, 220, 128, 64),
    ('ida 128->128 @64x220', 16, 64, 220, 128, 128),
    ('ida 256->128 @32x110', 16, 32, 110, 256, 128),
    ('ida 256->256 @32x110', 16, 32, 110, 256, 256),
    ('ida 512->256 @16x55', 16, 16, 55, 512, 256),
]
dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == 'fp16') else torch.bfloat16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(0)
for name, B, H, W, C, O in SHAPES:
    x = torch.randn(B, H, W, C, device='cuda').to(dt)
    w = torch.randn(O, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    pd = ops.pack_dcn_weight(w, dt)
    logits = torch.randn(B, H, W, 32, device='cuda') * float(os.environ.get('VD3D_DCN_SIGMA', '1.5'))   # spread of the offsets in pixels
    out = torch.empty(B, H, W, O, device='cuda', dtype=dt)
    run = lambda: ops.deform_conv_general(x, pd, logits[..., :18], logits[..., 18:27], out, 'nhwc', stride=(1, 1), padding=(1, 1),
                                          dilation=(1, 1), groups=1, deformable_groups=1, mask_sigmoid=True, relu=True)
    run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        run()
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) * 1e-3 / reps
    fl = 2.0 * B * H * W * O * 9 * C
    by = (x.numel() + out.numel()) * 2 + B * H * W * 27 * 4
    print('%-24s %8.1f us %7.1f TF/s %7.1f GB/s (algorithmic)' % (name, t * 1e6, fl / t / 1e12, by / t / 1e9), flush=True)
