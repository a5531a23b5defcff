"""``ConvBnReLU`` / ``AnchorFlatten`` with the reference's parameter names (lib/blocks.py:24-43,117-136)."""
import torch
import torch.nn as nn

from ... import hip_ops as ops
from . import fused


class ConvBnReLU(nn.Module):
    """conv + BN (+ReLU).  NB the reference ignores its ``relu`` argument (always ReLU, lib/blocks.py:36)."""

    def __init__(self, input_features=1, output_features=1, kernel_size=(1, 1), stride=[1, 1], padding='SAME',
                 dilation=1, groups=1, relu=True):
        super(ConvBnReLU, self).__init__()
        assert groups == 1
        pad_num = int((kernel_size[0] - 1) / 2) * dilation if padding.lower() == 'same' else 0
        self.sequence = nn.Sequential(
            nn.Conv2d(input_features, output_features, kernel_size=kernel_size, stride=stride, padding=pad_num,
                      dilation=dilation, groups=groups),
            nn.BatchNorm2d(output_features),
        )
        self.relu = True
        self._cache = fused.PackCache()

    def forward_nhwc(self, x, out=None):
        conv, bn = self.sequence[0], self.sequence[1]
        pc = self._cache.get(x.dtype, [conv.weight, conv.bias] + fused.bn_sources(bn),
                             lambda: ops.pack_conv(conv.weight, conv.bias, fused.bn_tuple(bn), x.dtype,
                                                   conv.stride[0], conv.padding[0], conv.dilation[0]))
        return ops.conv2d(x, pc, out=out, relu=self.relu)

    def forward(self, x):
        return fused.to_nchw(self.forward_nhwc(fused.to_nhwc(x)))


class AnchorFlatten(nn.Module):
    """[B, A*C, H, W] -> [B, H*W*A, C].  In NHWC the conv output already IS that layout: a pure view."""

    def __init__(self, num_output_channel):
        super(AnchorFlatten, self).__init__()
        self.num_output_channel = num_output_channel

    def forward_nhwc(self, x):
        return x.reshape(x.shape[0], -1, self.num_output_channel)

    def forward(self, x):
        x = x.permute(0, 2, 3, 1)
        return x.contiguous().view(x.shape[0], -1, self.num_output_channel)
