"""DLA up-sampling (backbones/dla_utils.py:42-155): every proj / node is DCNv2 3x3 + BN + ReLU (one fused kernel each,
``ModulatedDeformConvPack.forward_nhwc``), ``up`` is a depth-wise ConvTranspose2d with fixed bilinear weights, fused with the
``+ layers[i-1]`` that follows it."""
import math

import numpy as np
import torch
import torch.nn as nn

from ... import hip_ops as ops
from ..lib import fused
from ..lib.ops.dcn.deform_conv import ModulatedDeformConvPack


def fill_up_weights(up):
    w = up.weight.data
    f = math.ceil(w.size(2) / 2)
    c = (2 * f - 1 - f % 2) / (2. * f)
    for i in range(w.size(2)):
        for j in range(w.size(3)):
            w[0, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
    for c in range(1, w.size(0)):
        w[c, 0, :, :] = w[0, 0, :, :]


class DeformConv(nn.Module):
    """BN + ReLU after a modulated deformable conv (dla_utils.py:42-56)."""

    def __init__(self, chi, cho):
        super(DeformConv, self).__init__()
        self.actf = nn.Sequential(nn.BatchNorm2d(cho), nn.ReLU(inplace=True))
        self.conv = ModulatedDeformConvPack(chi, cho, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)

    def forward_nhwc(self, x):
        return self.conv.forward_nhwc(x, bn=self.actf[0], relu=True)

    def forward(self, x):
        return fused.to_nchw(self.forward_nhwc(fused.to_nhwc(x)))


class IDAUp(nn.Module):
    def __init__(self, o, channels, up_f):
        super(IDAUp, self).__init__()
        self._up_f = {}
        for i in range(1, len(channels)):
            c = channels[i]
            f = int(up_f[i])
            up = nn.ConvTranspose2d(o, o, f * 2, stride=f, padding=f // 2, output_padding=0, groups=o, bias=False)
            fill_up_weights(up)
            setattr(self, 'proj_' + str(i), DeformConv(c, o))
            setattr(self, 'up_' + str(i), up)
            setattr(self, 'node_' + str(i), DeformConv(o, o))
            self._up_f[i] = f
        self._cache = fused.PackCache()

    def forward_nhwc(self, layers, startp, endp):
        for i in range(startp + 1, endp):
            k = i - startp
            up, project, node = getattr(self, 'up_' + str(k)), getattr(self, 'proj_' + str(k)), getattr(self, 'node_' + str(k))
            f = self._up_f[k]
            w = self._cache.get(('up', k), [up.weight], lambda: up.weight.detach().float().reshape(up.weight.shape[0], -1).t().contiguous())
            summed = ops.dwconv_transpose(project.forward_nhwc(layers[i]), w, f, add=layers[i - 1])
            layers[i] = node.forward_nhwc(summed)


class DLAUp(nn.Module):
    def __init__(self, startp, channels, scales, in_channels=None):
        super(DLAUp, self).__init__()
        self.startp = startp
        if in_channels is None:
            in_channels = channels
        self.channels = channels
        channels = list(channels)
        scales = np.array(scales, dtype=int)
        for i in range(len(channels) - 1):
            j = -i - 2
            setattr(self, 'ida_{}'.format(i), IDAUp(channels[j], in_channels[j:], scales[j:] // scales[j]))
            scales[j + 1:] = scales[j]
            in_channels[j + 1:] = [channels[j] for _ in channels[j + 1:]]

    def forward_nhwc(self, layers):
        layers = list(layers)
        out = [layers[-1]]
        for i in range(len(layers) - self.startp - 1):
            getattr(self, 'ida_{}'.format(i)).forward_nhwc(layers, len(layers) - i - 2, len(layers))
            out.insert(0, layers[-1])
        return out


class DLASegUpsample(nn.Module):
    def __init__(self, input_channels, down_ratio=4, final_kernel=1, last_level=5, out_channel=0):
        super(DLASegUpsample, self).__init__()
        assert down_ratio in [2, 4, 8, 16]
        self.first_level = int(np.log2(down_ratio))
        self.last_level = last_level
        channels = list(input_channels)
        scales = [2 ** i for i in range(len(channels[self.first_level:]))]
        self.dla_up = DLAUp(self.first_level, channels[self.first_level:], scales)
        if out_channel == 0:
            out_channel = channels[self.first_level]
        self.ida_up = IDAUp(out_channel, channels[self.first_level:self.last_level],
                            [2 ** i for i in range(self.last_level - self.first_level)])

    def forward_nhwc(self, tensors):
        tensors = self.dla_up.forward_nhwc(tensors)
        y = [tensors[i] for i in range(self.last_level - self.first_level)]
        self.ida_up.forward_nhwc(y, 0, len(y))
        return y[-1]
