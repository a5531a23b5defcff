"""Micro-benchmark of IDA-Up's depth-wise transposed conv + add at the shapes of BASELINE config 5 (KM3D DLA-34, 16 x 512 x 1760, fp16):
the phase kernel against the generic per-element kernel (VD3D_DWCONVT_GENERIC).    python tools/bench_idaup.py [reps]"""
import sys

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import _lib, hip_ops as ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [('s32->s16 256ch f2', 16, 16, 55, 256, 2), ('s16->s8 128ch f2', 16, 32, 110, 128, 2), ('s8->s4 64ch f2', 16, 64, 220, 64, 2),
          ('s16->s4 64ch f4', 16, 32, 110, 64, 4)]
for name, B, H, W, C, f in SHAPES:
    x = torch.randn(B, H, W, C, device='cuda').half()
    wk = torch.randn(4 * f * f, C, device='cuda') * 0.3
    add = torch.randn(B, H * f, W * f, C, device='cuda').half()
    for label, sw in (('phase', False), ('generic', True)):
        if sw:
            _lib.check(_lib.lib().vd3d_test_set_switch(b'VD3D_DWCONVT_GENERIC', 1), 'switch')
        ops.dwconv_transpose(x, wk, f, add=add)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            ops.dwconv_transpose(x, wk, f, add=add)
        e.record()
        torch.cuda.synchronize()
        if sw:
            _lib.check(_lib.lib().vd3d_test_set_switch(b'VD3D_DWCONVT_GENERIC', 0), 'switch')
        t = s.elapsed_time(e) * 1e-3 / reps
        by = (x.numel() + 2 * add.numel()) * 2
        print('%-20s %-8s %8.1f us  %6.2f TB/s (in + add + out once)' % (name, label, t * 1e6, by / t / 1e12))
