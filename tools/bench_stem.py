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
from visualdet3d_amd import _lib  # noqa: E402
ref = ops.stem_conv_pool(imgs, pc, torch.bfloat16)
with _lib.test_switch('VD3D_STEM_WG4'):
    assert torch.equal(ops.stem_conv_pool(imgs, pc, torch.bfloat16).view(torch.int16), ref.view(torch.int16)), 'tile variants differ'
assert torch.equal(ops.stem_conv_pool(imgs, pc, torch.bfloat16, packed_first=True).view(torch.int16), ref.view(torch.int16)), 'packed path differs'
print('variants bit-identical; in 94.4 MB + out 62.9 MB')
for name, kw in (('fp32 direct 8x16 x1/CU', {}), ('fp32 direct 4x16 x2/CU', dict(sw=True)), ('pack + LDS-DMA stem', dict(packed_first=True)), ('fp32 direct 8x16 x1/CU', {}), ('fp32 direct 4x16 x2/CU', dict(sw=True))):
    sw = kw.pop('sw', False)
    _lib.lib().vd3d_test_set_switch(b'VD3D_STEM_WG4', int(sw))
    ops.stem_conv_pool(imgs, pc, torch.bfloat16, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.stem_conv_pool(imgs, pc, torch.bfloat16, **kw)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) * 1e-3 / reps
    print('%-24s %7.1f us per call (incl. the output allocation)  %.2f TB/s' % (name, t * 1e6, 157.3e6 / t / 1e12))
_lib.lib().vd3d_test_set_switch(b'VD3D_STEM_WG4', 0)
