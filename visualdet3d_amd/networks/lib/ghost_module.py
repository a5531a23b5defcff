"""GhostNet modules (lib/ghost_module.py:16-64) on HIP kernels.

``ResGhostModule`` output is ``cat[x, x1, x2]``: the caller hands in the concat buffer whose first ``inp`` channels
already hold ``x``; the primary conv reads that slice and writes ``x1`` into the next slice, the depth-wise conv
reads ``x1`` and writes ``x2`` -- no concat copy."""
import math

import torch
import torch.nn as nn

from ... import hip_ops as ops
from . import fused


class GhostModule(nn.Module):
    def __init__(self, inp, oup, kernel_size=1, ratio=2, dw_size=3, stride=1, relu=True):
        super(GhostModule, self).__init__()
        assert stride == 1 and relu, 'only the stride-1 ReLU variant is on the hot path'
        self.inp, self.oup = inp, oup
        init_channels = math.ceil(oup / ratio)
        new_channels = init_channels * (ratio - 1)
        self.init_channels, self.new_channels = init_channels, new_channels
        self.primary_conv = nn.Sequential(
            nn.Sequential(),
            nn.Conv2d(inp, init_channels, kernel_size, 1, kernel_size // 2, bias=False),
            nn.BatchNorm2d(init_channels),
            nn.ReLU(inplace=True),
        )
        self.cheap_operation = nn.Sequential(
            nn.Conv2d(init_channels, new_channels, dw_size, 1, dw_size // 2, groups=init_channels, bias=False),
            nn.BatchNorm2d(new_channels),
            nn.ReLU(inplace=True),
        )
        assert new_channels == init_channels and dw_size == 3, 'depth-wise multiplier 1 (ratio 2) only'
        self._cache = fused.PackCache()

    def _packed(self, dtype):
        conv, bn = self.primary_conv[1], self.primary_conv[2]
        dw, dbn = self.cheap_operation[0], self.cheap_operation[1]
        pc = self._cache.get(('p', dtype), [conv.weight] + fused.bn_sources(bn),
                             lambda: ops.pack_conv(conv.weight, None, fused.bn_tuple(bn), dtype, 1, conv.padding[0], 1))
        pd = self._cache.get('d', [dw.weight] + fused.bn_sources(dbn), lambda: ops.pack_dwconv(dw.weight, fused.bn_tuple(dbn)))
        return pc, pd

    def _run(self, x, out, off):
        """x: NHWC input; writes x1 -> out[..., off:off+init], x2 -> out[..., off+init:off+init+new]."""
        pc, pd = self._packed(x.dtype)
        i, n = self.init_channels, self.new_channels
        x1 = ops.conv2d(x, pc, out=out[..., off:off + i], relu=True)
        ops.dwconv3x3(x1, pd, out=out[..., off + i:off + i + n], relu=True)
        return out

    def forward_nhwc(self, x, out=None):
        B, H, W, _ = x.shape
        tot = self.init_channels + self.new_channels
        if out is None:
            out = torch.empty((B, H, W, tot), dtype=x.dtype, device=x.device)
        self._run(x, out, 0)
        return out[..., :self.oup]

    def forward(self, x):
        return fused.to_nchw(self.forward_nhwc(fused.to_nhwc(x)).contiguous())


class ResGhostModule(GhostModule):
    def __init__(self, inp, oup, kernel_size=1, ratio=2, dw_size=3, relu=True, stride=1):
        assert ratio > 2
        super(ResGhostModule, self).__init__(inp, oup - inp, kernel_size, ratio - 1, dw_size, relu=relu, stride=stride)
        self.oup = oup
        self.downsampling = None

    def forward_nhwc(self, x, out=None):
        """``out``: concat buffer [B,H,W,>=oup] whose first ``inp`` channels alias ``x`` (then nothing is copied)."""
        B, H, W, Cx = x.shape
        tot = Cx + self.init_channels + self.new_channels
        assert tot == self.oup, 'shipped sizes fit exactly (ghost_module.py:63-64 slice is a no-op)'
        if out is None:
            out = torch.empty((B, H, W, tot), dtype=x.dtype, device=x.device)
        if out.data_ptr() != x.data_ptr():
            ops.copy_channels(x, out[..., :Cx])
            x = out[..., :Cx]
        self._run(x, out, Cx)
        return out[..., :self.oup]

    def forward(self, x):
        return fused.to_nchw(self.forward_nhwc(fused.to_nhwc(x)))
