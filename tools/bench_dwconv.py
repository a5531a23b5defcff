"""vd3d_dwconv3x3 on the ghost modules' shapes of the headline step: runs of four pixels per thread against the one-pixel kernel (VD3D_DWCONV_PLAIN)."""
import sys
import torch
sys.path.insert(0, '.')
from visualdet3d_amd import _lib, hip_ops as ops

for name, B, H, W, C, tot in (('s4 ghost 24', 8, 96, 320, 24, 72), ('s8 ghost 96', 8, 48, 160, 96, 288), ('merge ghost 384', 8, 24, 80, 384, 1152)):
    w = torch.randn(C, 1, 3, 3, device='cuda') * 0.3
    bn = (torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1, torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5, 1e-5)
    pd = ops.pack_dwconv(w, bn)
    buf = torch.randn(B, H, W, tot, device='cuda').to(torch.bfloat16)
    src, dst = buf[..., tot - 2 * C:tot - C], buf[..., tot - C:]
    res = {}
    for rnd in range(3):
        for plain in (True, False):
            def run():
                if plain:
                    with _lib.test_switch('VD3D_DWCONV_PLAIN'):
                        for _ in range(50):
                            ops.dwconv3x3(src, pd, out=dst, relu=True)
                else:
                    for _ in range(50):
                        ops.dwconv3x3(src, pd, out=dst, relu=True)
            run()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(); e.record(); torch.cuda.synchronize()
            res.setdefault(plain, []).append(s.elapsed_time(e) / 50 * 1e3)
    by = B * H * W * C * 2 * 2
    print('%-16s one pixel per thread %.1f us (%.0f GB/s)   runs of four %.1f us (%.0f GB/s)' % (name, min(res[True]), by / min(res[True]) / 1e3, min(res[False]), by / min(res[False]) / 1e3), flush=True)
