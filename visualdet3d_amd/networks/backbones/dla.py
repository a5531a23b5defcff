"""DLA-34 backbone (backbones/dla.py:37-61 BasicBlock, :156-230 Root/Tree, :233-331 DLA, :428-440 dlanet) with the
reference's module / parameter names, on fused implicit-GEMM HIP kernels (NHWC).

Only the BasicBlock family (DLA-34, the KM3D / MonoFlex backbone) is on the hot path; the Bottleneck / BottleneckX
variants (DLA-60/102/169) are not in BASELINE's configs and raise.  ``Root``'s ``torch.cat`` is assembled with the strided
slice-copy kernel (the aggregated maps are small), everything else is conv + BN (+residual) + ReLU in one launch."""
import torch
import torch.nn as nn

from ... import hip_ops as ops
from ..lib import fused
from ..utils.registry import BACKBONE_DICT
from .resnet import _conv_bn

import os
_PAIR = not os.environ.get('VD3D_NO_LEVEL_PAIR')     # A/B: level0 and level1 as two launches

BatchNorm = nn.BatchNorm2d


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super(BasicBlock, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=dilation, bias=False, dilation=dilation)
        self.bn1 = BatchNorm(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=dilation, bias=False, dilation=dilation)
        self.bn2 = BatchNorm(planes)
        self.stride = stride
        self._cache = fused.PackCache()

    def forward_nhwc(self, x, residual=None, out=None):
        """``out``: optional channel slice of a wider NHWC buffer (the Root's concatenation) to write the block's output into."""
        if residual is None:
            residual = x
        dt = x.dtype
        y = ops.conv2d(x, _conv_bn(self._cache, 'c1', self.conv1, self.bn1, dt), relu=True)
        return ops.conv2d(y, _conv_bn(self._cache, 'c2', self.conv2, self.bn2, dt), out=out, residual=residual, relu=True)


class Root(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, residual):
        super(Root, self).__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=1, bias=False, padding=(kernel_size - 1) // 2)
        self.bn = BatchNorm(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.residual = residual
        self._cache = fused.PackCache()

    def forward_nhwc(self, *xs, buf=None):
        """``buf``: the concatenation buffer when the caller had the first inputs written straight into its leading channel slices
        (``xs[i]`` is then a view of ``buf``: no copy for it)."""
        B, H, W, _ = xs[0].shape
        tot = sum(t.shape[3] for t in xs)
        if buf is None:
            buf = torch.empty((B, H, W, tot), dtype=xs[0].dtype, device=xs[0].device)
        assert buf.shape[3] == tot
        off = 0
        for t in xs:
            dst = buf[..., off:off + t.shape[3]]
            if t.data_ptr() != dst.data_ptr():
                ops.copy_channels(t, dst)
            off += t.shape[3]
        pc = _conv_bn(self._cache, 'c', self.conv, self.bn, buf.dtype)
        return ops.conv2d(buf, pc, residual=xs[0] if self.residual else None, relu=True)


class Tree(nn.Module):
    def __init__(self, levels, block, in_channels, out_channels, stride=1, level_root=False, root_dim=0, root_kernel_size=1,
                 dilation=1, root_residual=False):
        super(Tree, self).__init__()
        if root_dim == 0:
            root_dim = 2 * out_channels
        if level_root:
            root_dim += in_channels
        if levels == 1:
            self.tree1 = block(in_channels, out_channels, stride, dilation=dilation)
            self.tree2 = block(out_channels, out_channels, 1, dilation=dilation)
        else:
            self.tree1 = Tree(levels - 1, block, in_channels, out_channels, stride, root_dim=0,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels, root_dim=root_dim + out_channels,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
        if levels == 1:
            self.root = Root(root_dim, out_channels, root_kernel_size, root_residual)
        self.level_root = level_root
        self.root_dim = root_dim
        self.downsample = None
        self.project = None
        self.levels = levels
        if stride > 1:
            assert stride == 2
            self.downsample = nn.MaxPool2d(stride, stride=stride)
        if in_channels != out_channels:
            self.project = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, bias=False), BatchNorm(out_channels))
        self._cache = fused.PackCache()

    def forward_nhwc(self, x, residual=None, children=None):
        children = [] if children is None else children
        bottom = ops.maxpool2x2(x) if self.downsample else x
        if self.project:
            residual = ops.conv2d(bottom, _conv_bn(self._cache, 'proj', self.project[0], self.project[1], x.dtype), relu=False)
        else:
            residual = bottom
        if self.level_root:
            children.append(bottom)
        if self.levels == 1:
            # the two blocks write straight into the leading channel slices of the Root's concatenation [x2 | x1 | children...]
            C = self.root.conv.out_channels
            B, H, W, _ = bottom.shape
            buf = torch.empty((B, H, W, 2 * C + sum(t.shape[3] for t in children)), dtype=x.dtype, device=x.device)
            x1 = self.tree1.forward_nhwc(x, residual, out=buf[..., C:2 * C])
            x2 = self.tree2.forward_nhwc(x1, out=buf[..., :C])
            x = self.root.forward_nhwc(x2, x1, *children, buf=buf)
            return x
        x1 = self.tree1.forward_nhwc(x, residual)
        if self.levels == 1:
            x2 = self.tree2.forward_nhwc(x1)
            x = self.root.forward_nhwc(x2, x1, *children)
        else:
            children.append(x1)
            x = self.tree2.forward_nhwc(x1, children=children)
        return x


class DLA(nn.Module):
    """Down-scale per output index: -1: 1, 0: 1, 1: 2, 2: 4, 3: 8, 4: 16, 5: 32."""

    def __init__(self, levels, channels, num_classes=1000, block=BasicBlock, residual_root=False,
                 out_indices=(-1, 0, 1, 2, 3, 4, 5)):
        super(DLA, self).__init__()
        self.channels = channels
        self.out_indices = out_indices
        self.num_classes = num_classes
        self.base_layer = nn.Sequential(nn.Conv2d(3, channels[0], kernel_size=7, stride=1, padding=3, bias=False),
                                        BatchNorm(channels[0]), nn.ReLU(inplace=True))
        self.level0 = self._make_conv_level(channels[0], channels[0], levels[0])
        self.level1 = self._make_conv_level(channels[0], channels[1], levels[1], stride=2)
        self.level2 = Tree(levels[2], block, channels[1], channels[2], 2, level_root=False, root_residual=residual_root)
        self.level3 = Tree(levels[3], block, channels[2], channels[3], 2, level_root=True, root_residual=residual_root)
        self.level4 = Tree(levels[4], block, channels[3], channels[4], 2, level_root=True, root_residual=residual_root)
        self.level5 = Tree(levels[5], block, channels[4], channels[5], 2, level_root=True, root_residual=residual_root)
        self._cache = fused.PackCache()

    def _make_conv_level(self, inplanes, planes, convs, stride=1, dilation=1):
        modules = []
        for i in range(convs):
            modules.extend([nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride if i == 0 else 1, padding=dilation,
                                      bias=False, dilation=dilation), BatchNorm(planes), nn.ReLU(inplace=True)])
            inplanes = planes
        return nn.Sequential(*modules)

    def _conv_level_nhwc(self, seq, name, x):
        for i in range(0, len(seq), 3):
            x = ops.conv2d(x, _conv_bn(self._cache, (name, i), seq[i], seq[i + 1], x.dtype), relu=True)
        return x

    def forward_nhwc(self, img_nchw, dtype=None, first_needed=0):
        """``first_needed``: the first level whose output the caller reads (DLA-Up starts at level 2): earlier entries of the result
        may be None -- level0 -> level1 (one conv each) then run as ONE launch and level0's tensor is never written."""
        dtype = dtype or fused.default_compute_dtype()
        conv, bn = self.base_layer[0], self.base_layer[1]
        pc = self._cache.get(('base', dtype), [conv.weight] + fused.bn_sources(bn),
                             lambda: ops.pack_image_conv(conv.weight, fused.bn_tuple(bn), dtype, 1, 3))
        y = []
        x = ops.image_conv(img_nchw.float().contiguous(), pc, relu=True)
        if -1 in self.out_indices:
            y.append(x)
        start = 0
        if first_needed >= 1 and len(self.level0) == 3 and len(self.level1) == 3 and _PAIR:
            pa = _conv_bn(self._cache, (0, 0), self.level0[0], self.level0[1], x.dtype)
            pb = _conv_bn(self._cache, (1, 0), self.level1[0], self.level1[1], x.dtype)
            if ops.conv2d_pair_supported(pa, pb):
                x = ops.conv2d_pair(x, pa, pb)
                if 0 in self.out_indices:
                    y.append(None)
                if 1 in self.out_indices:
                    y.append(x)
                start = 2
        for i in range(start, 6):
            lvl = getattr(self, 'level{}'.format(i))
            x = self._conv_level_nhwc(lvl, i, x) if i < 2 else lvl.forward_nhwc(x)
            if i in self.out_indices:
                y.append(x)
        return y

    def forward(self, x):
        return [fused.to_nchw(t) for t in self.forward_nhwc(x)]

    def load_pretrained_model(self, data_name, name):
        raise RuntimeError('no network access: load DLA weights with load_state_dict (checkpoint keys match the reference)')


def dla34(pretrained=None, **kwargs):
    model = DLA([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], block=BasicBlock, **kwargs)
    if pretrained is not None and pretrained is not False:
        model.load_pretrained_model(pretrained, 'dla34')
    return model


@BACKBONE_DICT.register_module
def dlanet(depth, **kwargs):
    if depth == 34:
        return dla34(**kwargs)
    raise ValueError('only DLA-34 (BasicBlock trees) is on the MI355X hot path; got depth %r' % (depth,))
