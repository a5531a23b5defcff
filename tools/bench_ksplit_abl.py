"""conv_ksplit256 on ResNet-34 layer3's shape (16 x 24 x 80, 256 -> 256 + residual): the shipped kernel, its owner-epilogue form (VD3D_KSPLIT_OWNER_EPILOGUE=1) and,
with the tuning library (VD3D_TUNING_LIB=libvd3d_hip_tuning.so), the timing ablations VD3D_X_KSPLIT_ABL = 1 ... 8 of the owner-epilogue form (results wrong by construction)."""
import os
import sys
import torch
sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops

torch.manual_seed(0)
B, H, W, C = 16, 24, 80, 256
x = torch.randn(B, H, W, C, device='cuda').abs().to(torch.bfloat16)
w = torch.randn(C, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
pc = ops.pack_conv(w, None, None, torch.bfloat16, 1, 1, 1)
res = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
out = torch.empty_like(res)
best = 1e9
for rnd in range(4):
    for _ in range(5):
        ops.conv2d(x, pc, out=out, residual=res, relu=True)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        ops.conv2d(x, pc, out=out, residual=res, relu=True)
    e.record()
    torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 50 * 1e3)
print('VD3D_X_KSPLIT_ABL=%s VD3D_KSPLIT_OWNER_EPILOGUE=%s: %.1f us' % (os.environ.get('VD3D_X_KSPLIT_ABL', '0'), os.environ.get('VD3D_KSPLIT_OWNER_EPILOGUE', ''), best))
