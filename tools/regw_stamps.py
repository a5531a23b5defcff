"""Tuning build only: cycle stamps of workgroup 0 / wave 0 of the register-resident-weights kernel (cfg 67) on the layer2 shape.
    VD3D_TUNING_LIB=1 python tools/regw_stamps.py"""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops, _lib
B, H, W, Cin, Cout = 16, 48, 160, 128, 128
torch.manual_seed(0)
x = torch.randn(B, H, W, Cin, device='cuda').to(torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.03
pc = ops.pack_conv(w, None, None, torch.bfloat16, 1, 1, 1)
res = torch.randn(B, H, W, Cout, device='cuda').to(torch.bfloat16)
lib = _lib.lib()
lib.vd3d_test_force_conv_tile(67)
for _ in range(3):
    ops.conv2d(x, pc, residual=res, relu=True)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
h = C.CDLL(_lib.LIB_PATH)
h.vd3d_tuning_regw_stamps(buf, 64)
t = [int(v) for v in buf]
names = ['start', 'prologue done (weights + unit 0 + ring)']
for k in range(4):
    names += ['tile %d top' % k, 'tile %d phase0 done' % k, 'tile %d phase1 done' % k, 'tile %d epilogue done' % k]
names += ['end']
for i, n in enumerate(names):
    if i < len(t) and t[i]:
        print('%-45s %10d  (+%d)' % (n, t[i] - t[0], t[i] - t[i - 1] if i else 0))
