"""Fused KM3D head (vd3d_km3d_head_fused) in isolation: 16 x 128 x 440 x 64 features, nine branches.
    python tools/bench_km3d_head.py [fp16|bf16]"""
import sys

import torch

sys.path.insert(0, '.')
from visualdet3d_amd.networks.heads.km3d_head import KM3DHead  # noqa: E402
from visualdet3d_amd.utils import synthetic as syn  # noqa: E402

dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == 'fp16') else torch.bfloat16
cfg = syn.km3d_cfg(output_w=440)
head = KM3DHead(**cfg.head).cuda().eval()
x = torch.randn(16, 128, 440, 64, device='cuda').to(dt)
fl = 2.0 * 16 * 128 * 440 * (2304 * 576 + 9 * 256 * 32)
for name in ('fused head',):
    with torch.no_grad():
        head.forward_nhwc(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            head.forward_nhwc(x)
        e.record()
        torch.cuda.synchronize()
    t = s.elapsed_time(e) * 1e-3 / 5
    print('%-18s %8.1f us  %6.1f TF/s' % (name, t * 1e6, fl / t / 1e12), flush=True)
