"""Micro-benchmark of the PSM cosine cost volume at the shapes of BASELINE configs 2 (R34: C 64 / 128) and 3 (R50: C 256 / 512):
the MFMA kernel against the VALU kernel (VD3D_PSM_VALU=1 selects it).    python tools/bench_psm.py [reps]"""
import sys

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import _lib, hip_ops as ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [('C2 s4  8x96x320x64', 8, 96, 320, 64), ('C2 s8  8x48x160x128', 8, 48, 160, 128),
          ('C3 s4 32x72x320x256', 32, 72, 320, 256), ('C3 s8 32x36x160x512', 32, 36, 160, 512)]
for name, B, H, W, C in SHAPES:
    L = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    R = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    out = torch.empty(B, H, W, 24, device='cuda', dtype=torch.bfloat16)
    for label, sw in (('mfma', False), ('valu', True)):
        if sw:
            _lib.check(_lib.lib().vd3d_test_set_switch(b'VD3D_PSM_VALU', 1), 'switch')
        ops.psm_cosine(L, R, 24, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            ops.psm_cosine(L, R, 24, out=out)
        e.record()
        torch.cuda.synchronize()
        if sw:
            _lib.check(_lib.lib().vd3d_test_set_switch(b'VD3D_PSM_VALU', 0), 'switch')
        t = s.elapsed_time(e) * 1e-3 / reps
        by = (2 * L.numel() + out.numel()) * 2
        print('%-22s %-5s %8.1f us  %6.2f TB/s (L + R + cost once)' % (name, label, t * 1e6, by / t / 1e12))
