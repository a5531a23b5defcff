"""``LookGround`` (lib/look_ground.py:11-71) with the reference's parameter names, on HIP kernels.

Three launches: (1) disp_create conv3x3 C->1 as an implicit GEMM (fp32 logits), (2) vd3d_look_ground_sample: tanh,
geometric prior, vertical bilinear gather of [x ; prior disparity] (grid_sample border / align_corners=True),
(3) the 1x1 ``extract`` conv as an implicit GEMM whose epilogue is ``relu(x + (conv + bias) * alpha)``."""
import ctypes as C

import torch
import torch.nn as nn

from ... import _lib
from ... import hip_ops as ops
from . import fused


class LookGround(nn.Module):
    def __init__(self, input_features, baseline=0.54, relative_elevation=1.65):
        super(LookGround, self).__init__()
        self.disp_create = nn.Sequential(nn.Conv2d(input_features, 1, 3, padding=1), nn.Tanh())
        self.extract = nn.Conv2d(1 + input_features, input_features, 1)
        self.baseline = baseline
        self.relative_elevation = relative_elevation
        self.alpha = nn.Parameter(torch.tensor([0.0], dtype=torch.float32))
        self._cache = fused.PackCache()

    def forward_nhwc(self, x, P2):
        B, H, W, Cc = x.shape
        dt = x.dtype
        conv = self.disp_create[0]
        pcd = self._cache.get(('disp', dt), [conv.weight, conv.bias], lambda: ops.pack_conv(conv.weight, conv.bias, None, dt, 1, 1, 1))
        disp = ops.conv2d(x, pcd, relu=False, out_f32=True)  # [B,H,W,1] fp32
        ve = 8 if ops.is16(dt) else 4
        cpad = (Cc + 1 + ve - 1) // ve * ve
        sampled = torch.empty((B, H, W, cpad), dtype=dt, device=x.device)
        P2 = P2.to(device=x.device, dtype=torch.float32).contiguous()
        _lib.check(_lib.lib().vd3d_look_ground_sample(ops._p(x), ops._p(disp), ops._p(P2), ops._p(sampled), B, H, W, Cc,
                                                      x.stride(2), sampled.stride(2), float(self.baseline),
                                                      float(self.relative_elevation), ops.dtype_code(dt), ops._stream()),
                   'vd3d_look_ground_sample')

        def build_extract():
            # torch.cat([disparity, x]) puts the prior at channel 0; the sampled buffer holds it at channel C
            w = self.extract.weight.detach().float()
            w2 = torch.zeros((Cc, cpad, 1, 1), dtype=torch.float32, device=w.device)
            w2[:, :Cc] = w[:, 1:]
            w2[:, Cc] = w[:, 0]
            pc = ops.pack_conv(w2, None, None, dt, 1, 0, 1)
            a = self.alpha.detach().float()
            pc.scale = a.expand(Cc).contiguous()
            pc.shift = (self.extract.bias.detach().float() * a).contiguous()
            return pc

        pce = self._cache.get(('ext', dt), [self.extract.weight, self.extract.bias, self.alpha], build_extract)
        return ops.conv2d(sampled, pce, residual=x, relu=True)

    def forward(self, inputs):
        x = fused.to_nhwc(inputs['features'])
        return fused.to_nchw(self.forward_nhwc(x, inputs['P2']))
