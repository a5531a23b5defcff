"""Cycle stamps of one workgroup of dcn_bf64_kernel (library built with -DVD3D_STAMPS as libvd3d_hip_stamps.so; VD3D_TUNING_LIB=libvd3d_hip_stamps.so)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops, _lib  # noqa: E402

dt = torch.float16
torch.manual_seed(0)
B, H, W, Cc, O = 16, 128, 440, 64, 64
x = torch.randn(B, H, W, Cc, device='cuda').to(dt)
w = torch.randn(O, Cc, 3, 3, device='cuda') * (2.0 / (9 * Cc)) ** 0.5
pd = ops.pack_dcn_weight(w, dt)
logits = torch.randn(B, H, W, 32, device='cuda') * float(os.environ.get('VD3D_DCN_SIGMA', '0.8'))
out = torch.empty(B, H, W, O, device='cuda', dtype=dt)
for _ in range(3):
    ops.deform_conv_general(x, pd, logits[..., :18], logits[..., 18:27], out, 'nhwc', stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1,
                            deformable_groups=1, mask_sigmoid=True, relu=True)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 128)()
h = _lib.lib()
h.vd3d_debug_read_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
print('rc', h.vd3d_debug_read_stamps(buf))
names = ['start', 'window issued', 'logits requested', 'geometry written', 'vmcnt(0)', 'barrier', 'far test'] + ['tap %d' % t for t in range(9)] + ['parked', 'stored']
for wv in range(4):
    t0 = buf[wv * 32]
    print('wave %d: ' % wv + '  '.join('%s +%d' % (names[i], buf[wv * 32 + i] - (buf[wv * 32 + i - 1] if i else t0)) for i in range(18)), ' total', buf[wv * 32 + 17] - t0)
