"""Micro-benchmark of the fused stem (conv 7x7/s2 + BN + ReLU + max pool) at the bench shape (16 x 3 x 384 x 1280, L | R):
the fp32-direct kernel against pack launches + the LDS-DMA kernel.    python tools/bench_stem.py [reps]"""
import sys

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(0)
w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5, 1e-5)
pc = ops.pack_stem_conv(w.cuda(), tuple(t.cuda() if torch.is_tensor(t) else t for t in bn), torch.bfloat16)
imgs = [torch.randn(8, 3, 384, 1280, generator=g).cuda(), torch.randn(8, 3, 384, 1280, generator=g).cuda()]
for name, kw in (('fp32 direct', {}), ('pack + LDS-DMA stem', dict(packed_first=True)), ('fp32 direct', {}), ('pack + LDS-DMA stem', dict(packed_first=True))):
    ops.stem_conv_pool(imgs, pc, torch.bfloat16, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.stem_conv_pool(imgs, pc, torch.bfloat16, **kw)
    e.record()
    torch.cuda.synchronize()
    print('%-22s %7.1f us per call (incl. the output allocation)' % (name, s.elapsed_time(e) * 1e3 / reps))
