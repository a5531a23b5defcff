"""Deformable convolution modules with the reference's API, parameter names and checkpoint behaviour
(lib/ops/dcn/deform_conv.py:52-489), forward only, on the HIP kernel of csrc/deform_conv.hip.

  * functional ``deform_conv`` / ``modulated_deform_conv``: NCHW fp32 tensors, same argument meaning as the reference's
    autograd Functions' forward (backward is training-only and out of scope: NotImplementedError);
  * ``DeformConv``, ``DeformConvPack``, ``ModulatedDeformConv``, ``ModulatedDeformConvPack`` (``_version = 2`` + the legacy
    ``*_offset`` key remap);
  * ``ModulatedDeformConvPack.forward_nhwc``: engine path -- offset conv as an implicit GEMM (fp32 logits), then ONE kernel
    that samples, applies sigmoid(mask), contracts on MFMA and applies bias + folded BN + ReLU."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.utils import _pair, _single

from ..... import hip_ops as ops
from ... import fused
from . import deform_conv_ext


def deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
    if input is not None and input.dim() != 4:
        raise ValueError('Expected 4D tensor as input, got {}D tensor instead.'.format(input.dim()))
    if not input.is_cuda:
        raise NotImplementedError
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    cur = min(im2col_step, input.shape[0])
    assert (input.shape[0] % cur) == 0, 'im2col step must divide batchsize'
    # same call as DeformConvFunction.forward (deform_conv.py:78-95): caller-allocated output, two empty scratch tensors,
    # W-before-H argument order
    B, _, H, W = input.shape
    O, _, kh, kw = weight.shape
    Ho = (H + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
    output = input.new_empty((B, O, Ho, Wo))
    bufs = [input.new_empty(0), input.new_empty(0)]
    deform_conv_ext.deform_conv_forward(input, weight, offset, output, bufs[0], bufs[1], weight.size(3), weight.size(2), stride[1], stride[0],
                                        padding[1], padding[0], dilation[1], dilation[0], groups, deformable_groups, cur)
    return output


def modulated_deform_conv(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    if not input.is_cuda:
        raise NotImplementedError
    # same call as ModulatedDeformConvFunction.forward (deform_conv.py:170-186); like the reference the module-level API takes
    # scalar stride / padding / dilation
    with_bias = bias is not None
    if not with_bias:
        bias = input.new_empty(1)  # fake tensor
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    B, _, H, W = input.shape
    O, _, kh, kw = weight.shape
    output = input.new_empty((B, O, (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1))
    bufs = [input.new_empty(0), input.new_empty(0)]
    deform_conv_ext.modulated_deform_conv_forward(input, weight, bias, bufs[0], offset, mask, output, bufs[1], kh, kw, sh, sw, ph, pw, dh, dw,
                                                  groups, deformable_groups, with_bias)
    return output


class DeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super(DeformConv, self).__init__()
        assert not bias
        assert in_channels % groups == 0 and out_channels % groups == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = _pair(stride), _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.transposed = False
        self.output_padding = _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // self.groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        # inputs smaller than the kernel are zero-padded, as in the reference (deform_conv.py:283-297)
        input_pad = (x.size(2) < self.kernel_size[0] or x.size(3) < self.kernel_size[1])
        if input_pad:
            pad_h = max(self.kernel_size[0] - x.size(2), 0)
            pad_w = max(self.kernel_size[1] - x.size(3), 0)
            x = F.pad(x, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
            offset = F.pad(offset, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
        out = deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups, self.deformable_groups)
        if input_pad:
            out = out[:, :, :out.size(2) - pad_h, :out.size(3) - pad_w].contiguous()
        return out


def _remap_legacy_offset_keys(state_dict, prefix, local_metadata):
    version = local_metadata.get('version', None)
    if version is None or version < 2:
        for leaf in ('weight', 'bias'):
            new, old = prefix + 'conv_offset.' + leaf, prefix[:-1] + '_offset.' + leaf
            if new not in state_dict and old in state_dict:
                state_dict[new] = state_dict.pop(old)


class DeformConvPack(DeformConv):
    """Offsets ``[y0, x0, y1, x1, ...]`` produced by an ordinary conv (zero-initialised)."""
    _version = 2

    def __init__(self, *args, **kwargs):
        super(DeformConvPack, self).__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels, self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride), padding=_pair(self.padding),
                                     dilation=_pair(self.dilation), bias=True)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()
        self._cache = fused.PackCache()

    def forward(self, x):
        offset = _offset_conv_nchw(self, x)
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups, self.deformable_groups)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        _remap_legacy_offset_keys(state_dict, prefix, local_metadata)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


def _offset_conv_nchw(mod, x):
    """conv_offset through the implicit-GEMM kernel in fp32 (offsets steer sampling positions: keep them exact)."""
    conv = mod.conv_offset
    ve = 4
    cin_pad = (conv.in_channels + ve - 1) // ve * ve
    pc = mod._cache.get(('off', torch.float32, cin_pad), [conv.weight, conv.bias],
                        lambda: ops.pack_conv(conv.weight, conv.bias, None, torch.float32, conv.stride[0], conv.padding[0],
                                              conv.dilation[0], cin_pad=cin_pad))
    B, Cc, H, W = x.shape
    xh = torch.zeros((B, H, W, cin_pad), dtype=torch.float32, device=x.device) if cin_pad != Cc else None
    if xh is None:
        xh = ops.nchw_f32_to_nhwc(x, torch.float32)
    else:
        ops.nchw_f32_to_nhwc(x, torch.float32, out=xh[..., :Cc])
    return ops.nhwc_to_nchw_f32(ops.conv2d(xh, pc, relu=False))


class ModulatedDeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super(ModulatedDeformConv, self).__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups = groups, deformable_groups
        self.with_bias = bias
        self.transposed = False
        self.output_padding = _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.init_weights()

    def init_weights(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                     self.groups, self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    _version = 2
    columns_above = 256     # forward_nhwc: out_channels beyond one fused tile -> sampled columns in HBM + strip GEMM

    def __init__(self, *args, **kwargs):
        super(ModulatedDeformConvPack, self).__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels, self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride), padding=_pair(self.padding),
                                     dilation=_pair(self.dilation), bias=True)
        self.init_weights()
        self._cache = fused.PackCache()

    def init_weights(self):
        super(ModulatedDeformConvPack, self).init_weights()
        if hasattr(self, 'conv_offset'):
            self.conv_offset.weight.data.zero_()
            self.conv_offset.bias.data.zero_()

    def forward(self, x):
        """Reference contract: NCHW fp32 in, NCHW fp32 out (deform_conv.py:459-466)."""
        if not x.is_cuda:
            raise NotImplementedError
        logits = _offset_conv_nchw(self, x)              # [B, 3K, Ho, Wo] fp32 = (o1 | o2 | mask logits)
        # the reference's chunk(3) + cat(o1, o2) + sigmoid(mask) (deform_conv.py:461-464) without any torch arithmetic: (o1 | o2)
        # IS the first 2K channels, and the kernel applies the sigmoid to the mask logits it reads (mask_sigmoid = 1)
        K = self.kernel_size[0] * self.kernel_size[1]
        G = self.deformable_groups
        x = x.float().contiguous()
        pd = self._cache.get(('w', torch.float32), [self.weight], lambda: ops.pack_dcn_weight(self.weight, torch.float32))
        out = torch.empty((x.shape[0], self.out_channels) + tuple(logits.shape[2:]), dtype=torch.float32, device=x.device)
        bias = self.bias.detach().float() if self.bias is not None else None
        return ops.deform_conv_general(x, pd, logits[:, :2 * K * G], logits[:, 2 * K * G:3 * K * G], out, 'nchw', bias=bias,
                                       stride=_pair(self.stride), padding=_pair(self.padding), dilation=_pair(self.dilation),
                                       groups=self.groups, deformable_groups=G, mask_sigmoid=True)

    def forward_nhwc(self, x, bn=None, relu=False, out=None):
        """Engine path on NHWC activations (bf16 | fp32): offset logits (fp32) -> fused sample + sigmoid(mask) + MFMA +
        bias + folded ``bn`` (an nn.BatchNorm2d in eval mode, optional) + ReLU."""
        assert self.deformable_groups == 1, 'every call site of the reference uses deformable_groups = 1 (SURVEY.md 7.3)'
        dt = x.dtype
        conv = self.conv_offset
        K = self.kernel_size[0] * self.kernel_size[1]

        def build_offset_conv():
            # 3*K = 27 output channels padded with zero filters to a multiple of 4 (32): 16-byte fp32 stores in the epilogue
            # instead of 27 scalar ones per pixel; the DCN kernels read offsets / mask through strides, so the pad is free
            O3 = conv.weight.shape[0]
            Op = (O3 + 3) // 4 * 4 if O3 % 4 else O3
            Op = max(Op, 32) if O3 <= 32 else Op
            w = torch.zeros((Op,) + tuple(conv.weight.shape[1:]), dtype=torch.float32, device=conv.weight.device)
            b = torch.zeros((Op,), dtype=torch.float32, device=conv.weight.device)
            w[:O3] = conv.weight.detach().float()
            b[:O3] = conv.bias.detach().float()
            return ops.pack_conv(w, b, None, dt, conv.stride[0], conv.padding[0], conv.dilation[0])

        pco = self._cache.get(('off', dt), [conv.weight, conv.bias], build_offset_conv)
        logits = ops.conv2d(x, pco, relu=False, out_f32=True)          # [B,Ho,Wo,>=3*K] fp32: (o1 | o2 | mask) == (offsets 0:2K | mask 2K:3K | pad)
        pd = self._cache.get(('w', dt), [self.weight], lambda: ops.pack_dcn_weight(self.weight, dt))
        scale = shift = None
        if bn is not None:
            scale, shift = self._cache.get('bn', fused.bn_sources(bn), lambda: ops.fold_bn(None, fused.bn_tuple(bn), self.out_channels, x.device))
        B, Ho, Wo, _ = logits.shape
        if self.out_channels > self.columns_above and self.groups == 1:
            # many output channels (stereo base head: 2176 -> 2176): the fused kernel would re-sample the columns once per 256
            # output channels (9 times, 12 ms at 32 x 18 x 80); write them once and contract on the strip tiles (3.9 ms)
            cols = ops.deform_columns(x, logits[..., :2 * K], logits[..., 2 * K:3 * K], self.kernel_size, _pair(self.stride), _pair(self.padding),
                                      _pair(self.dilation), mask_sigmoid=True)

            def build_gemm():
                w = self.weight.detach().float().permute(0, 2, 3, 1).reshape(self.out_channels, K * self.in_channels, 1, 1)
                bn_t = fused.bn_tuple(bn) if bn is not None else None
                return ops.pack_conv(w, self.bias, bn_t, dt, 1, 0, 1)

            pcg = self._cache.get(('gemm', dt, id(bn)), [self.weight, self.bias] + (fused.bn_sources(bn) if bn is not None else []), build_gemm)
            return ops.conv2d(cols, pcg, out=out, relu=relu)
        if out is None:
            out = torch.empty((B, Ho, Wo, self.out_channels), dtype=dt, device=x.device)
        bias = self.bias.detach().float() if self.bias is not None else None
        return ops.deform_conv_general(x, pd, logits[..., :2 * K], logits[..., 2 * K:3 * K], out, 'nhwc', bias=bias, scale=scale, shift=shift,
                                       stride=_pair(self.stride), padding=_pair(self.padding), dilation=_pair(self.dilation),
                                       groups=self.groups, deformable_groups=1, mask_sigmoid=True, relu=relu)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        _remap_legacy_offset_keys(state_dict, prefix, local_metadata)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
