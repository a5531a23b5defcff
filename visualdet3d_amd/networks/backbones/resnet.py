"""ResNet-18/34/50/101/152 backbone with the reference's constructor, ``out_indices`` semantics and parameter
names (backbones/resnet.py:95-198,255-270), executed as fused implicit-GEMM HIP kernels on NHWC activations:

  conv + BN(eval) + ReLU                  -> one kernel (BN folded into the epilogue scale/shift)
  conv + BN + residual add + ReLU         -> one kernel (residual read in the epilogue)
  7x7/s2 stem                             -> NHWC4 image pack + implicit GEMM over (7 rows x 32 elements)

Inference only (eval-mode BN); the reference's freeze_stages / norm_eval only affect ``train()`` there."""
import os

import torch
import torch.nn as nn

from ... import hip_ops as ops
from ..lib import fused
from ..utils.registry import BACKBONE_DICT


def conv3x3(in_planes, out_planes, stride=1, dilation=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, bias=False, dilation=dilation)


def _conv_bn(cache, key, conv, bn, dtype):
    return cache.get((key, dtype), [conv.weight, conv.bias] + fused.bn_sources(bn),
                     lambda: ops.pack_conv(conv.weight, conv.bias, fused.bn_tuple(bn), dtype,
                                           conv.stride[0], conv.padding[0], conv.dilation[0]))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super(BasicBlock, self).__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes, dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self._cache = fused.PackCache()

    def forward_nhwc(self, x, out=None):
        dt = x.dtype
        y = ops.conv2d(x, _conv_bn(self._cache, 'c1', self.conv1, self.bn1, dt), relu=True)
        res = x
        if self.downsample is not None:
            # (the 1x1 / stride-2 downsample on a side stream under conv1 was measured: no gain, 1989 / 1996 vs 1997 / 2005 img/s)
            res = ops.conv2d(x, _conv_bn(self._cache, 'ds', self.downsample[0], self.downsample[1], dt), relu=False)
        return ops.conv2d(y, _conv_bn(self._cache, 'c2', self.conv2, self.bn2, dt), out=out, residual=res, relu=True)

    def forward(self, x):
        return fused.to_nchw(self.forward_nhwc(fused.to_nhwc(x)))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation, bias=False, dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self._cache = fused.PackCache()

    fuse_block = not os.environ.get('VD3D_NO_BLOCK_FUSION')    # the 64-wide stage (layer1) as ONE launch per block (vd3d_conv2d_bottleneck); False: the separate launches (A/B, tests)

    def forward_nhwc(self, x, out=None):
        dt = x.dtype
        pc1 = _conv_bn(self._cache, 'c1', self.conv1, self.bn1, dt)
        pc2 = _conv_bn(self._cache, 'c2', self.conv2, self.bn2, dt)
        pc3 = _conv_bn(self._cache, 'c3', self.conv3, self.bn3, dt)
        pcd = _conv_bn(self._cache, 'ds', self.downsample[0], self.downsample[1], dt) if self.downsample is not None else None
        if self.fuse_block and ops.conv2d_bottleneck_supported(x, pc1, pc2, pc3, pcd):
            # x read once (+ halo), the output written once, both 64-channel intermediates in LDS (csrc/conv_bottleneck.hip)
            return ops.conv2d_bottleneck(x, pc1, pc2, pc3, pcd, out=out)
        y = ops.conv2d(x, pc1, relu=True)
        y = ops.conv2d(y, pc2, relu=True)
        res = x
        if pcd is not None:
            res = ops.conv2d(x, pcd, relu=False)
        return ops.conv2d(y, pc3, out=out, residual=res, relu=True)

    def forward(self, x):
        return fused.to_nchw(self.forward_nhwc(fused.to_nhwc(x)))


class ResNet(nn.Module):
    planes = [64, 128, 256, 512]

    def __init__(self, block, layers, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(-1, 0, 1, 2, 3), frozen_stages=-1, norm_eval=True):
        self.inplanes = 64
        super(ResNet, self).__init__()
        assert 1 <= num_stages <= 4 and max(out_indices) < num_stages
        self.num_stages, self.strides, self.dilations = num_stages, strides, dilations
        self.out_indices, self.frozen_stages, self.norm_eval = out_indices, frozen_stages, norm_eval
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for i in range(num_stages):
            setattr(self, 'layer%d' % (i + 1),
                    self._make_layer(block, self.planes[i], layers[i], stride=self.strides[i], dilation=self.dilations[i]))
        self._cache = fused.PackCache()
        self.fuse_stem_pool = True

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        # NB: the first block of a stage does not receive `dilation` (backbones/resnet.py:147) -- kept.
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=dilation))
        return nn.Sequential(*layers)

    def forward_nhwc(self, img_nchw, dtype=None, on_stage=None):
        """img_nchw: [B,3,H,W] fp32 (the reference's input format), or a tuple of such tensors stacked on the batch axis.
        Returns the NHWC feature list.  ``on_stage(i, x)``: optional callback right after stage i's output has been
        enqueued (the stereo neck uses it to start the work that only depends on that stage on a side stream)."""
        dtype = dtype or fused.default_compute_dtype()
        pc = self._cache.get(('stem', dtype), [self.conv1.weight] + fused.bn_sources(self.bn1),
                             lambda: ops.pack_stem_conv(self.conv1.weight, fused.bn_tuple(self.bn1), dtype))
        outs = []
        imgs = [t.float().contiguous() for t in img_nchw] if isinstance(img_nchw, (list, tuple)) else img_nchw.float().contiguous()
        H, W = (imgs[0] if isinstance(imgs, list) else imgs).shape[2:]
        if self.fuse_stem_pool and -1 not in self.out_indices and ops.stem_pool_supported(H, W, dtype, pc.Cout):
            x = ops.stem_conv_pool(imgs, pc, dtype)      # conv1 + bn1 + relu + maxpool in one kernel
        else:
            x = ops.stem_conv(imgs, pc, dtype)
            if -1 in self.out_indices:
                outs.append(x)
            x = ops.maxpool3x3s2(x)
        for i in range(self.num_stages):
            for blk in getattr(self, 'layer%d' % (i + 1)):
                x = blk.forward_nhwc(x)
            if i in self.out_indices:
                outs.append(x)
            if on_stage is not None:
                on_stage(i, x)
        return outs

    def forward(self, img_batch):
        return [fused.to_nchw(o) for o in self.forward_nhwc(img_batch)]


_ARCH = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
         101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


@BACKBONE_DICT.register_module
def resnet(depth, pretrained=False, **kwargs):
    """Same factory signature as backbones/resnet.py:255-270.  ``pretrained=True`` downloads ImageNet weights in the
    reference; there is no network here, so weights come from ``load_state_dict`` (checkpoint keys match)."""
    if depth not in _ARCH:
        raise ValueError('Unsupported model depth, must be one of 18, 34, 50, 101, 152')
    block, layers = _ARCH[depth]
    return ResNet(block, layers, **kwargs)
