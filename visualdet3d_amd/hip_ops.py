"""torch-tensor front end of the C-ABI in libvd3d_hip.so.

PyTorch is plumbing here (device memory, streams); every arithmetic op below is a hand-written HIP kernel.
Activation convention: NHWC tensors ``[B, H, W, C]`` (bf16 or fp32) with unit channel stride; a tensor may be a
channel-slice view of a wider buffer (``buf[..., off:off+C]``) -- that is how ``torch.cat`` along channels is
fused away: producers write straight into their slice of the concat buffer.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import VD3D_BF16, VD3D_F16, VD3D_F32, ConvParams, HeadParams, check


def dtype_code(dt):
    if dt == torch.bfloat16:
        return VD3D_BF16
    if dt == torch.float16:
        return VD3D_F16
    if dt == torch.float32:
        return VD3D_F32
    raise TypeError('HIP path supports bfloat16 / float16 / float32 activations, got %s' % dt)


def is16(dt):
    """bf16 or fp16: the two 16-bit storage formats (same kernels, vector widths and MFMA rate)."""
    return dt in (torch.bfloat16, torch.float16)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.Vd3dError('HIP op called with a CPU tensor: there is no CPU fallback in the product path')


def _nhwc_strides(x):
    """(pix_stride, row_stride, batch_stride) in elements of an NHWC (possibly channel-sliced) tensor."""
    assert x.dim() == 4 and x.stride(3) == 1, 'expected NHWC with unit channel stride'
    return x.stride(2), x.stride(1), x.stride(0)


def _dense_pixels(x):
    B, H, W, _ = x.shape
    ps, rs, bs = _nhwc_strides(x)
    return rs == W * ps and (B == 1 or bs == H * W * ps)


def _bytes_from(x):
    return x.untyped_storage().nbytes() - x.storage_offset() * x.element_size()


def new_act(B, H, W, C_, dtype, device):
    return torch.empty((B, H, W, C_), dtype=dtype, device=device)


# --------------------------------------------------------------------------------------------- conv
class PackedConv:
    """Conv weight packed for the implicit-GEMM kernel: ``[CoutPad, Kpad]`` with K = (ky, kx, c), plus the folded
    epilogue ``y = conv * scale + shift`` (bias and eval-mode BatchNorm)."""

    def __init__(self, w, scale, shift, Cin, Cout, kh, kw, stride, pad, dil, Kpad, CoutPad, dtype):
        self.w, self.scale, self.shift = w, scale, shift
        self.Cin, self.Cout, self.kh, self.kw = Cin, Cout, kh, kw
        self.stride, self.pad, self.dil = stride, pad, dil
        self.Kpad, self.CoutPad, self.dtype = Kpad, CoutPad, dtype
        self.w_frag = None      # optional MFMA register image of the same weights (vd3d_conv_params.weight_frag)
        self.ws_need = {}       # split-K scratch bytes per launch geometry (conv2d)


def fold_bn(bias, bn, cout, device):
    """(scale, shift) fp32 from an optional conv bias and optional eval-mode BN ``(gamma, beta, mean, var, eps)``."""
    shift = bias.detach().float() if bias is not None else torch.zeros(cout, device=device)
    scale = None
    if bn is not None:
        gamma, beta, mean, var, eps = bn
        scale = gamma.detach().float() / torch.sqrt(var.detach().float() + eps)
        shift = shift * scale + (beta.detach().float() - mean.detach().float() * scale)
    return (scale.contiguous() if scale is not None else None), shift.contiguous()


def pack_conv(weight, bias=None, bn=None, dtype=torch.bfloat16, stride=1, pad=0, dil=1, cin_pad=None):
    """weight: OIHW fp32 (on the GPU).  cin_pad: pad input channels with zeros up to this count (the activation
    buffer then has that many channels)."""
    O, I, kh, kw = weight.shape
    dev = weight.device
    ve = 8 if is16(dtype) else 4
    bke = 64 if is16(dtype) else 32
    Cin = cin_pad or I
    assert Cin % ve == 0, 'input channels must be a multiple of %d for %s' % (ve, dtype)
    w = weight.detach().float().permute(0, 2, 3, 1)  # O, kh, kw, I
    if Cin != I:
        w = torch.nn.functional.pad(w, (0, Cin - I))
    K = kh * kw * Cin
    Kpad = (K + bke - 1) // bke * bke
    CoutPad = (O + 127) // 128 * 128
    packed = torch.zeros((CoutPad, Kpad), dtype=dtype, device=dev)
    packed[:O, :K] = w.reshape(O, K).to(dtype)
    scale, shift = fold_bn(bias, bn, O, dev)
    pc = PackedConv(packed, scale, shift, Cin, O, kh, kw, stride, pad, dil, Kpad, CoutPad, dtype)
    if is16(dtype) and (kh, kw, stride, pad, dil) == (3, 3, 1, 1, 1) and O % 32 == 0 and (Cin in (64, 128, 256) or (Cin % 64 == 0 and O == 32)):
        # register image for the resident-weight kernels and the narrow-output streaming kernel (O = 32, any Cin % 64 == 0)
        # (layout: include/vd3d.h, vd3d_conv_params.weight_frag):
        # [O/32][Cin/64][tap*4 + ks][half*32 + lr][8]  <-  w[32*nb + lr][tap][64*kc + (2*ks + half)*8 + e]
        w7 = w.reshape(O // 32, 32, 9, Cin // 64, 4, 2, 8)                 # nb, lr, tap, kc, ks, half, e
        pc.w_frag = w7.permute(0, 3, 2, 4, 5, 1, 6).contiguous().to(dtype)
    elif is16(dtype) and (kh, kw, stride, pad) == (1, 1, 1, 0) and Cin in (64, 128, 256) and O % 256 == 0:
        # the same register image with one tap, for the point-wise streaming kernel: [O/32][Cin/64][ks][half*32 + lr][8]
        w7 = w.reshape(O // 32, 32, 1, Cin // 64, 4, 2, 8)
        pc.w_frag = w7.permute(0, 3, 2, 4, 5, 1, 6).contiguous().to(dtype)
    return pc


def pack_stem_conv(weight, bn, dtype):
    """7x7/s2 stem (backbones/resnet.py:118) as a (7 x 1)-tap conv over 32-element rows: the packed image
    (vd3d_pack_image_nhwc4) is NHWC4 with a 3-px zero border, so kernel row ky of output pixel (oy, ox) is the 8
    consecutive pixels starting at padded (2*oy + ky, 2*ox): 8 px * 4 ch = 32 contiguous elements (8th px and 4th
    channel carry zero weights)."""
    O, I, kh, kw = weight.shape
    assert (I, kh, kw) == (3, 7, 7)
    dev = weight.device
    w = torch.zeros((O, 7, 8, 4), dtype=torch.float32, device=dev)
    w[:, :, :7, :3] = weight.detach().float().permute(0, 2, 3, 1)
    K = 7 * 32
    bke = 64 if is16(dtype) else 32
    Kpad = (K + bke - 1) // bke * bke
    CoutPad = (O + 127) // 128 * 128
    packed = torch.zeros((CoutPad, Kpad), dtype=dtype, device=dev)
    packed[:O, :K] = w.reshape(O, K).to(dtype)
    scale, shift = fold_bn(None, bn, O, dev)
    return PackedConv(packed, scale, shift, 32, O, 7, 1, 2, 0, 1, Kpad, CoutPad, dtype)


_MAX_IN_BYTES = 0x7ffffff0      # the kernels address the input with 32-bit buffer offsets


def conv2d(x, pc, out=None, residual=None, relu=False, out_f32=False):
    """Fused conv (+bias/BN) (+residual) (+ReLU).  x: NHWC (C >= pc.Cin used), returns NHWC ``[B,Ho,Wo,Cout]``.  An input
    view spanning more than 2 GiB (e.g. a channel slice of KM3D's 9 x 256-channel head buffer at 16 x 128 x 440) is processed
    in batch chunks."""
    _require_cuda(x, pc.w, out, residual)
    B, H, W, Cx = x.shape
    assert x.dtype == pc.dtype and Cx == pc.Cin, (x.dtype, pc.dtype, Cx, pc.Cin)
    ips, irs, ibs = _nhwc_strides(x)
    span1 = ((H - 1) * irs + (W - 1) * ips + Cx) * x.element_size()          # bytes one sample's view reaches over
    if B > 1 and (B - 1) * ibs * x.element_size() + span1 > _MAX_IN_BYTES:
        nb = max(1, (_MAX_IN_BYTES - span1) // (ibs * x.element_size()) + 1)
        Ho_ = (H + 2 * pc.pad - pc.dil * (pc.kh - 1) - 1) // pc.stride + 1
        Wo_ = (W + 2 * pc.pad - pc.dil * (pc.kw - 1) - 1) // pc.stride + 1
        if out is None:
            out = torch.empty((B, Ho_, Wo_, pc.Cout), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
        for b0 in range(0, B, nb):
            conv2d(x[b0:b0 + nb], pc, out=out[b0:b0 + nb], residual=None if residual is None else residual[b0:b0 + nb],
                   relu=relu, out_f32=out_f32)
        return out
    Ho = (H + 2 * pc.pad - pc.dil * (pc.kh - 1) - 1) // pc.stride + 1
    Wo = (W + 2 * pc.pad - pc.dil * (pc.kw - 1) - 1) // pc.stride + 1
    odt = torch.float32 if out_f32 else x.dtype
    if out is None:
        out = torch.empty((B, Ho, Wo, pc.Cout), dtype=odt, device=x.device)
    assert out.shape == (B, Ho, Wo, pc.Cout) and out.dtype == odt and _dense_pixels(out)
    p = ConvParams()
    p.in_, p.weight, p.out = x.data_ptr(), pc.w.data_ptr(), out.data_ptr()
    p.scale = pc.scale.data_ptr() if pc.scale is not None else None
    p.shift = pc.shift.data_ptr() if pc.shift is not None else None
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == x.dtype and _dense_pixels(residual)
        p.residual, p.res_pix_stride = residual.data_ptr(), residual.stride(2)
    p.B, p.H, p.W, p.Cin = B, H, W, pc.Cin
    p.in_pix_stride, p.in_row_stride, p.in_batch_stride = ips, irs, ibs
    p.in_bytes = min(_bytes_from(x), (B - 1) * ibs * x.element_size() + span1)
    p.Ho, p.Wo, p.Cout = Ho, Wo, pc.Cout
    p.out_pix_stride = out.stride(2)
    p.kh, p.kw, p.stride, p.pad, p.dil = pc.kh, pc.kw, pc.stride, pc.pad, pc.dil
    p.Kpad, p.CoutPad, p.relu = pc.Kpad, pc.CoutPad, int(relu)
    p.dtype, p.out_f32 = dtype_code(x.dtype), int(out_f32)
    p.weight_frag = pc.w_frag.data_ptr() if pc.w_frag is not None else None
    # split-K scratch for low-parallelism shapes (batch-1 calls): the library says how much it wants (0 for everything the batched
    # configurations launch).  The answer depends only on the launch geometry (shape, strides, alignments, epilogue form) and the
    # library's test-hook state (+ whether a register image of the weights is attached, the activation type and the device: the routes that
    # split need `weight_frag`, the plan reads the device's CU count), so it is asked ONCE per (packed conv, geometry) and remembered on the packed conv; the scratch itself is
    # a fresh stream-ordered allocation per call -- inside a hipGraph capture it lives in the graph's pool
    wkey = (B, H, W, ips, irs, ibs, p.out_pix_stride, p.out & 127, p.residual and (p.res_pix_stride, p.residual & 15), p.out_f32,
            (p.scale or 0) & 15, (p.shift or 0) & 15, _lib.hook_epoch(), pc.w_frag is not None, x.dtype, x.device.index)
    need = pc.ws_need.get(wkey)
    if need is None:
        need = _lib.lib().vd3d_conv2d_workspace_bytes(C.byref(p))
        if need < 0:
            check(1, 'vd3d_conv2d_workspace_bytes')
        if len(pc.ws_need) > 64:
            pc.ws_need.clear()
        pc.ws_need[wkey] = need
    ws = None
    if need > 0:
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        p.splitk_ws, p.splitk_ws_bytes = ws.data_ptr(), need
    check(_lib.lib().vd3d_conv2d_igemm(C.byref(p), _stream()), 'vd3d_conv2d_igemm')
    return out


def conv2d_pair(x, pc_a, pc_b, relu_a=True, relu_b=True):
    """Two chained 3x3 convs in ONE launch (vd3d_conv2d_pair): ``pc_a`` stride 1, 16 -> 16, then ``pc_b`` stride 2, 16 -> <= 32, each
    + folded BN (+ ReLU) -- DLA level0 -> level1.  The intermediate tensor is never written.  Bit-identical to two ``conv2d`` calls."""
    _require_cuda(x, pc_a.w, pc_b.w)
    B, H, W, Cx = x.shape
    assert x.dtype == pc_a.dtype == pc_b.dtype and is16(x.dtype) and Cx == pc_a.Cin
    ips, irs, ibs = _nhwc_strides(x)
    span1 = ((H - 1) * irs + (W - 1) * ips + Cx) * x.element_size()
    assert (B - 1) * ibs * x.element_size() + span1 <= _MAX_IN_BYTES, 'conv2d_pair: input view beyond 2 GiB'
    Ho = (H + 2 * pc_b.pad - pc_b.dil * (pc_b.kh - 1) - 1) // pc_b.stride + 1
    Wo = (W + 2 * pc_b.pad - pc_b.dil * (pc_b.kw - 1) - 1) // pc_b.stride + 1
    out = torch.empty((B, Ho, Wo, pc_b.Cout), dtype=x.dtype, device=x.device)

    def params(pc, Hi, Wi, Hout, Wout, relu):
        p = ConvParams()
        p.in_, p.weight, p.out = x.data_ptr(), pc.w.data_ptr(), out.data_ptr()
        p.scale = pc.scale.data_ptr() if pc.scale is not None else None
        p.shift = pc.shift.data_ptr() if pc.shift is not None else None
        p.B, p.H, p.W, p.Cin = B, Hi, Wi, pc.Cin
        p.in_pix_stride, p.in_row_stride, p.in_batch_stride = ips, irs, ibs
        p.in_bytes = min(_bytes_from(x), (B - 1) * ibs * x.element_size() + span1)
        p.Ho, p.Wo, p.Cout = Hout, Wout, pc.Cout
        p.out_pix_stride = out.stride(2)
        p.kh, p.kw, p.stride, p.pad, p.dil = pc.kh, pc.kw, pc.stride, pc.pad, pc.dil
        p.Kpad, p.CoutPad, p.relu = pc.Kpad, pc.CoutPad, int(relu)
        p.dtype, p.out_f32 = dtype_code(x.dtype), 0
        return p

    pa, pb = params(pc_a, H, W, H, W, relu_a), params(pc_b, H, W, Ho, Wo, relu_b)
    check(_lib.lib().vd3d_conv2d_pair(C.byref(pa), C.byref(pb), _stream()), 'vd3d_conv2d_pair')
    return out


def conv2d_bottleneck_supported(x, pc1, pc2, pc3, pc_ds=None):
    """shapes vd3d_conv2d_bottleneck takes: the 64-wide Bottleneck (ResNet-50 / 101 / 152 layer1) on a dense 16-bit NHWC input."""
    k = lambda pc: (pc.kh, pc.kw, pc.stride, pc.pad, pc.dil)
    cin = 64 if pc_ds is not None else 256
    return (is16(x.dtype) and x.is_contiguous() and x.shape[3] == cin and pc1.dtype == pc2.dtype == pc3.dtype == x.dtype and
            k(pc1) == (1, 1, 1, 0, 1) and k(pc2) == (3, 3, 1, 1, 1) and k(pc3) == (1, 1, 1, 0, 1) and
            (pc1.Cin, pc1.Cout, pc2.Cin, pc2.Cout, pc3.Cin, pc3.Cout) == (cin, 64, 64, 64, 64, 256) and
            (pc_ds is None or (k(pc_ds) == (1, 1, 1, 0, 1) and (pc_ds.Cin, pc_ds.Cout) == (64, 256) and pc_ds.dtype == x.dtype)) and
            x.numel() * x.element_size() <= _MAX_IN_BYTES and x.shape[0] * x.shape[1] * x.shape[2] * 256 * 2 <= _MAX_IN_BYTES)


def conv2d_bottleneck(x, pc1, pc2, pc3, pc_ds=None, out=None):
    """A whole ResNet Bottleneck in ONE launch (vd3d_conv2d_bottleneck): relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + identity), identity = x
    (``pc_ds`` None) or bn(downsample 1x1 (x)).  Both 64-channel intermediates stay in LDS; rounding points are those of the separate launches."""
    _require_cuda(x, pc1.w, pc2.w, pc3.w, out)
    B, H, W, Cx = x.shape
    assert conv2d_bottleneck_supported(x, pc1, pc2, pc3, pc_ds)
    if out is None:
        out = torch.empty((B, H, W, 256), dtype=x.dtype, device=x.device)
    assert out.shape == (B, H, W, 256) and out.dtype == x.dtype and _dense_pixels(out)

    def params(pc, relu, residual=None):
        p = ConvParams()
        p.in_, p.weight, p.out = x.data_ptr(), pc.w.data_ptr(), out.data_ptr()
        p.scale = pc.scale.data_ptr() if pc.scale is not None else None
        p.shift = pc.shift.data_ptr() if pc.shift is not None else None
        p.B, p.H, p.W, p.Cin = B, H, W, pc.Cin
        p.in_pix_stride, p.in_row_stride, p.in_batch_stride = Cx, W * Cx, H * W * Cx
        p.in_bytes = x.numel() * x.element_size()
        p.Ho, p.Wo, p.Cout = H, W, pc.Cout
        p.out_pix_stride = out.stride(2)
        p.kh, p.kw, p.stride, p.pad, p.dil = pc.kh, pc.kw, pc.stride, pc.pad, pc.dil
        p.Kpad, p.CoutPad, p.relu = pc.Kpad, pc.CoutPad, int(relu)
        p.dtype, p.out_f32 = dtype_code(x.dtype), 0
        if residual is not None:
            p.residual, p.res_pix_stride = residual.data_ptr(), residual.stride(2)
        return p

    p1, p2, p3 = params(pc1, True), params(pc2, True), params(pc3, True, None if pc_ds is not None else x)
    pd = params(pc_ds, False) if pc_ds is not None else None
    check(_lib.lib().vd3d_conv2d_bottleneck(C.byref(p1), C.byref(p2), C.byref(p3), C.byref(pd) if pd is not None else None, _stream()),
          'vd3d_conv2d_bottleneck')
    return out


def conv2d_pair_supported(pc_a, pc_b):
    k = lambda pc: (pc.kh, pc.kw, pc.pad, pc.dil)
    return (is16(pc_a.dtype) and pc_a.dtype == pc_b.dtype and k(pc_a) == k(pc_b) == (3, 3, 1, 1) and pc_a.stride == 1 and pc_b.stride == 2 and
            pc_a.Cin == 16 and pc_a.Cout == 16 and pc_b.Cin == 16 and pc_b.Cout in (16, 32))


def _pack_stem_images(img_nchw, dtype):
    """NCHW fp32 image(s) -> bordered NHWC4 [B, H+6, W+8, 4].  A list / tuple of images is packed into consecutive batch
    slices (stereo: left then right -- no torch.cat copy)."""
    imgs = list(img_nchw) if isinstance(img_nchw, (list, tuple)) else [img_nchw]
    _require_cuda(*imgs)
    _, Cc, H, W = imgs[0].shape
    B = sum(int(t.shape[0]) for t in imgs)
    Hp, Wp = H + 6, W + 8  # 3 px border top/bottom/left, 5 px right (keeps rows 16-byte aligned, covers the 8-px tap)
    packed = torch.empty((B, Hp, Wp, 4), dtype=dtype, device=imgs[0].device)
    b0 = 0
    for t in imgs:
        assert t.shape[1:] == (3, H, W) and t.dtype == torch.float32 and t.is_contiguous()
        check(_lib.lib().vd3d_pack_image_nhwc4(_p(t), _p(packed[b0:]), int(t.shape[0]), H, W, 3, 3, 5, dtype_code(dtype), _stream()),
              'vd3d_pack_image_nhwc4')
        b0 += int(t.shape[0])
    return packed, B, H, W


def stem_pool_supported(H, W, dtype, Cout):
    return dtype == torch.bfloat16 and Cout == 64 and H % 4 == 0 and W % 4 == 0 and (H // 4) % 8 == 0 and (W // 4) % 16 == 0


_STEM_PACKED = bool(os.environ.get('VD3D_STEM_PACKED'))     # A/B: the round-2 path (pack launches + LDS-DMA stem)


def stem_conv_pool(img_nchw, pc, dtype, packed_first=False):
    """Stem conv 7x7/s2 + BN + ReLU + MaxPool(3, 2, 1) fused: returns the pooled NHWC map.  One or two fp32 NCHW image tensors
    (stereo: left, right -- stacked on the batch axis by the kernel) are read DIRECTLY (vd3d_stem_conv_pool_f32);
    ``packed_first=True``: the round-2 path (vd3d_pack_image_nhwc4 per tensor, then vd3d_stem_conv_pool on the packed copy)."""
    imgs = list(img_nchw) if isinstance(img_nchw, (list, tuple)) else [img_nchw]
    if packed_first or len(imgs) > 2 or _STEM_PACKED:
        packed, B, H, W = _pack_stem_images(img_nchw, dtype)
        assert stem_pool_supported(H, W, dtype, pc.Cout)
        out = torch.empty((B, H // 4, W // 4, 64), dtype=dtype, device=packed.device)
        check(_lib.lib().vd3d_stem_conv_pool(_p(packed), _p(pc.w), _p(pc.scale) if pc.scale is not None else None, _p(pc.shift), _p(out),
                                             B, H, W, pc.Kpad, 64, _stream()), 'vd3d_stem_conv_pool')
        return out
    _require_cuda(*imgs)
    _, Cc, H, W = imgs[0].shape
    for t in imgs:
        assert t.shape[1:] == (3, H, W) and t.dtype == torch.float32 and t.is_contiguous()
    assert stem_pool_supported(H, W, dtype, pc.Cout)
    B0, B1 = int(imgs[0].shape[0]), (int(imgs[1].shape[0]) if len(imgs) == 2 else 0)
    out = torch.empty((B0 + B1, H // 4, W // 4, 64), dtype=dtype, device=imgs[0].device)
    check(_lib.lib().vd3d_stem_conv_pool_f32(_p(imgs[0]), B0, _p(imgs[1]) if B1 else None, B1, _p(pc.w),
                                             _p(pc.scale) if pc.scale is not None else None, _p(pc.shift), _p(out), H, W, pc.Kpad, 64, _stream()),
          'vd3d_stem_conv_pool_f32')
    return out


def stem_conv(img_nchw, pc, dtype):
    """Stem: pack NCHW fp32 image to bordered NHWC4, then the 7x7/s2 conv + BN + ReLU as an implicit GEMM."""
    packed, B, H, W = _pack_stem_images(img_nchw, dtype)
    Hp, Wp = H + 6, W + 8
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    out = torch.empty((B, Ho, Wo, pc.Cout), dtype=dtype, device=packed.device)
    p = ConvParams()
    p.in_, p.weight, p.out = packed.data_ptr(), pc.w.data_ptr(), out.data_ptr()
    p.scale = pc.scale.data_ptr() if pc.scale is not None else None
    p.shift = pc.shift.data_ptr()
    # "pixel" = 4-element NHWC4 pixel; Cin = 32 contiguous elements per tap row; bounds never trigger (bordered)
    p.B, p.H, p.W, p.Cin = B, Hp, Wp, 32
    p.in_pix_stride, p.in_row_stride, p.in_batch_stride = 4, Wp * 4, Hp * Wp * 4
    p.in_bytes = _bytes_from(packed)
    p.Ho, p.Wo, p.Cout = Ho, Wo, pc.Cout
    p.out_pix_stride = pc.Cout
    p.kh, p.kw, p.stride, p.pad, p.dil = 7, 1, 2, 0, 1
    p.Kpad, p.CoutPad, p.relu = pc.Kpad, pc.CoutPad, 1
    p.dtype, p.out_f32 = dtype_code(dtype), 0
    check(_lib.lib().vd3d_conv2d_igemm(C.byref(p), _stream()), 'vd3d_conv2d_igemm(stem)')
    return out


# --------------------------------------------------------------------------------------------- elementwise
def maxpool3x3s2(x, out=None):
    _require_cuda(x)
    B, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    assert _dense_pixels(x) and _dense_pixels(out)
    check(_lib.lib().vd3d_maxpool3x3s2(_p(x), _p(out), B, H, W, Cc, x.stride(2), out.stride(2), dtype_code(x.dtype), _stream()),
          'vd3d_maxpool3x3s2')
    return out


def avgpool2x2(x, out=None):
    _require_cuda(x)
    B, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((B, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
    assert _dense_pixels(x) and _dense_pixels(out)
    check(_lib.lib().vd3d_avgpool2x2(_p(x), _p(out), B, H, W, Cc, x.stride(2), out.stride(2), dtype_code(x.dtype), _stream()),
          'vd3d_avgpool2x2')
    return out


class PackedDW:
    def __init__(self, w, scale, shift, C_):
        self.w, self.scale, self.shift, self.C = w, scale, shift, C_


def pack_dwconv(weight, bn):
    """weight [C,1,3,3] -> [9][C] fp32 + folded BN."""
    Cc = weight.shape[0]
    w = weight.detach().float().reshape(Cc, 9).t().contiguous()
    scale, shift = fold_bn(None, bn, Cc, weight.device)
    return PackedDW(w, scale, shift, Cc)


def dwconv3x3(x, pd, out=None, relu=True):
    _require_cuda(x)
    B, H, W, Cc = x.shape
    assert Cc == pd.C
    if out is None:
        out = torch.empty((B, H, W, Cc), dtype=x.dtype, device=x.device)
    assert _dense_pixels(x) and _dense_pixels(out)
    check(_lib.lib().vd3d_dwconv3x3(_p(x), _p(pd.w), _p(pd.scale), _p(pd.shift), _p(out), B, H, W, Cc,
                                    x.stride(2), out.stride(2), int(relu), dtype_code(x.dtype), _stream()), 'vd3d_dwconv3x3')
    return out


def copy_channels(x, out):
    _require_cuda(x, out)
    B, H, W, Cc = x.shape
    assert out.shape == x.shape and out.dtype == x.dtype and _dense_pixels(x) and _dense_pixels(out)
    check(_lib.lib().vd3d_copy_channels(_p(x), _p(out), B * H * W, Cc, x.stride(2), out.stride(2), dtype_code(x.dtype), _stream()),
          'vd3d_copy_channels')
    return out


def nhwc_to_nchw_f32(x):
    _require_cuda(x)
    B, H, W, Cc = x.shape
    assert _dense_pixels(x)
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
    check(_lib.lib().vd3d_nhwc_to_nchw_f32(_p(x), _p(out), B, H, W, Cc, x.stride(2), dtype_code(x.dtype), _stream()),
          'vd3d_nhwc_to_nchw_f32')
    return out


def nchw_f32_to_nhwc(x, dtype, out=None):
    _require_cuda(x)
    B, Cc, H, W = x.shape
    x = x.float().contiguous()
    if out is None:
        out = torch.empty((B, H, W, Cc), dtype=dtype, device=x.device)
    check(_lib.lib().vd3d_nchw_f32_to_nhwc(_p(x), _p(out), B, H, W, Cc, out.stride(2), dtype_code(dtype), _stream()),
          'vd3d_nchw_f32_to_nhwc')
    return out


# --------------------------------------------------------------------------------------------- stereo
def _bf16_or_f32(t, what):
    """the stereo-only kernels (PSM cosine volume, concat volume, 3-D convs) are instantiated for bf16 and fp32 only"""
    if t.dtype not in (torch.bfloat16, torch.float32):
        raise _lib.Vd3dError('%s: dtype %s not supported (bf16 | fp32 only; there is no fp16 instantiation)' % (what, t.dtype))


def psm_cosine(left, right, D, out=None):
    _require_cuda(left, right)
    _bf16_or_f32(left, 'psm_cosine')
    B, H, W, Cc = left.shape
    assert right.shape == left.shape and left.dtype == right.dtype
    assert _dense_pixels(left) and _dense_pixels(right) and left.stride(2) == right.stride(2)
    if out is None:
        out = torch.empty((B, H, W, D), dtype=left.dtype, device=left.device)
    assert out.shape == (B, H, W, D) and _dense_pixels(out)
    check(_lib.lib().vd3d_psm_cosine(_p(left), _p(right), _p(out), B, H, W, Cc, D, left.stride(2), out.stride(2),
                                     dtype_code(left.dtype), _stream()), 'vd3d_psm_cosine')
    return out


def costvol_build(left, right, D):
    _require_cuda(left, right)
    _bf16_or_f32(left, 'costvol_build')
    B, H, W, F = left.shape
    assert _dense_pixels(left) and _dense_pixels(right) and left.stride(2) == right.stride(2)
    vol = torch.empty((B, D, H, W, 2 * F), dtype=left.dtype, device=left.device)
    check(_lib.lib().vd3d_costvol_build(_p(left), _p(right), _p(vol), B, H, W, F, D, left.stride(2),
                                        dtype_code(left.dtype), _stream()), 'vd3d_costvol_build')
    return vol


class PackedConv3d:
    def __init__(self, w, scale, shift, Cin, Cout):
        self.w, self.scale, self.shift, self.Cin, self.Cout = w, scale, shift, Cin, Cout


def pack_conv3d(weight, bias, bn):
    """weight [O,I,3,3,3] -> [27][I][O] fp32 + folded bias/BN3d."""
    O, I = weight.shape[:2]
    w = weight.detach().float().permute(2, 3, 4, 1, 0).reshape(27, I, O).contiguous()
    scale, shift = fold_bn(bias, bn, O, weight.device)
    if scale is None:
        scale = torch.ones(O, device=weight.device)
    return PackedConv3d(w, scale, shift, I, O)


def conv3d_3x3x3(vol, pc, relu=True, out_nhwc=None):
    """vol: [B,D,H,W,Cin] channels-last.  out_nhwc given => write NHWC [B,H,W,Cout*D] (channel = f*D + d)."""
    _require_cuda(vol)
    _bf16_or_f32(vol, 'conv3d_3x3x3')
    B, D, H, W, Cin = vol.shape
    assert Cin == pc.Cin and vol.is_contiguous()
    if out_nhwc is not None:
        assert out_nhwc.shape == (B, H, W, pc.Cout * D) and _dense_pixels(out_nhwc)
        out, fd, ops = out_nhwc, 1, out_nhwc.stride(2)
    else:
        out, fd, ops = torch.empty((B, D, H, W, pc.Cout), dtype=vol.dtype, device=vol.device), 0, 0
    check(_lib.lib().vd3d_conv3d_3x3x3(_p(vol), _p(pc.w), _p(pc.scale), _p(pc.shift), _p(out), B, D, H, W, Cin, pc.Cout,
                                       int(relu), fd, ops, dtype_code(vol.dtype), _stream()), 'vd3d_conv3d_3x3x3')
    return out


# --------------------------------------------------------------------------------------------- head
def cost_volume_fused_supported(left, D):
    return left.dtype == torch.bfloat16 and left.shape[3] == 8 and D <= 24 and left.stride(2) % 8 == 0


def cost_volume_fused(left, right, p0, p1, D, out=None):
    """CostVolume after the 1x1 down-sample in one launch (vd3d_cost_volume_fused): left / right NHWC [B,H,W,8] bf16 ->
    NHWC [B,H,W,8*D] (channel = f*D + d).  The three-launch path (costvol_build + 2 x conv3d_3x3x3) is the fp32 / general one."""
    _require_cuda(left, right)
    B, H, W, F = left.shape
    assert right.shape == left.shape and cost_volume_fused_supported(left, D) and left.stride(2) == right.stride(2)
    assert _dense_pixels(left) and _dense_pixels(right) and (p0.Cin, p0.Cout, p1.Cin, p1.Cout) == (2 * F, F, F, F)
    if out is None:
        out = torch.empty((B, H, W, F * D), dtype=left.dtype, device=left.device)
    assert out.shape == (B, H, W, F * D) and out.dtype == left.dtype and _dense_pixels(out)
    check(_lib.lib().vd3d_cost_volume_fused(_p(left), _p(right), _p(p0.w), _p(p0.scale), _p(p0.shift), _p(p1.w), _p(p1.scale), _p(p1.shift),
                                            _p(out), B, H, W, F, D, left.stride(2), out.stride(2), dtype_code(left.dtype), _stream()),
          'vd3d_cost_volume_fused')
    return out


def _head_params(cls, reg, anchors, prior, P2, A, n_cls, n_types, img_hw, score_thr, nms_iou_thr, use_filter, y_min_max, x_max, max_cand,
                 max_det, workspace):
    p = HeadParams()
    B, N, _ = cls.shape
    p.cls, p.anchors, p.prior_mean_std, p.P2 = cls.data_ptr(), anchors.data_ptr(), prior.data_ptr(), P2.data_ptr()
    p.reg = reg.data_ptr() if reg is not None else None
    p.B, p.N, p.A, p.n_cls, p.n_types = B, N, A, n_cls, n_types
    p.img_h, p.img_w = int(img_hw[0]), int(img_hw[1])
    p.score_thr, p.nms_iou_thr = float(score_thr), float(nms_iou_thr)
    p.filter_y_min, p.filter_y_max, p.filter_x_max = float(y_min_max[0]), float(y_min_max[1]), float(x_max)
    p.use_filter, p.max_cand, p.max_det = int(use_filter), int(max_cand), int(max_det)
    p.workspace = workspace.data_ptr()
    return p


def head_select(cls, anchors, prior, P2, A, n_cls, n_types, img_hw, score_thr, nms_iou_thr, use_filter=True, y_min_max=(-0.5, 1.8),
                x_max=40.0, max_cand=4096, max_det=None, workspace=None):
    """Stage 1 of the head post-processing on the CURRENT stream (vd3d_head_select): needs only the class logits, so the detector
    runs it on the cls tower's side stream under the reg tower.  ``P2`` must be fp32 contiguous on the device; ``workspace`` as for
    ``head_postprocess``.  Finish with ``head_postprocess(..., preselected=True)`` (same arguments) once the streams have joined."""
    _require_cuda(cls, anchors, prior, P2, workspace)
    B, N, nc1 = cls.shape
    assert nc1 == n_cls + 1 and cls.dtype == torch.float32 and cls.is_contiguous() and P2.dtype == torch.float32 and P2.is_contiguous()
    assert workspace.numel() >= _lib.lib().vd3d_head_workspace_bytes(B, max_cand)
    p = _head_params(cls, None, anchors, prior, P2, A, n_cls, n_types, img_hw, score_thr, nms_iou_thr, use_filter, y_min_max, x_max,
                     max_cand, max_det or max_cand, workspace)
    check(_lib.lib().vd3d_head_select(C.byref(p), _stream()), 'vd3d_head_select')


def head_postprocess(cls, reg, anchors, prior, P2, A, n_cls, n_types, img_hw, score_thr, nms_iou_thr,
                     use_filter=True, y_min_max=(-0.5, 1.8), x_max=40.0, max_cand=4096, max_det=None, workspace=None, preselected=False):
    """Device-side get_bboxes for a whole batch.  Returns padded (scores [B,K], boxes [B,K,11], labels [B,K] i32,
    anchor_idx [B,K] i32, count [B] i32) -- all on the device, no host sync.  ``preselected``: ``head_select`` already ran with the
    same arguments and workspace (stream-ordered before this call): only the decode / NMS stage is launched."""
    _require_cuda(cls, reg, anchors, prior, P2)
    B, N, nc1 = cls.shape
    assert nc1 == n_cls + 1 and reg.shape == (B, N, 12) and cls.dtype == torch.float32 and reg.dtype == torch.float32
    assert cls.is_contiguous() and reg.is_contiguous() and anchors.is_contiguous() and prior.is_contiguous()
    assert anchors.shape == (N, 4) and prior.shape == (A, n_types, 6, 2) and P2.shape == (B, 3, 4)
    P2 = P2.float().contiguous()
    max_det = max_det or max_cand
    dev = cls.device
    need = _lib.lib().vd3d_head_workspace_bytes(B, max_cand)
    if workspace is None or workspace.numel() < need:
        assert not preselected, 'preselected=True needs the workspace head_select wrote'
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    scores = torch.empty((B, max_det), dtype=torch.float32, device=dev)
    boxes = torch.empty((B, max_det, 11), dtype=torch.float32, device=dev)
    labels = torch.empty((B, max_det), dtype=torch.int32, device=dev)
    aidx = torch.empty((B, max_det), dtype=torch.int32, device=dev)
    count = torch.empty((B,), dtype=torch.int32, device=dev)
    p = _head_params(cls, reg, anchors, prior, P2, A, n_cls, n_types, img_hw, score_thr, nms_iou_thr, use_filter, y_min_max, x_max,
                     max_cand, max_det, workspace)
    p.out_scores, p.out_boxes, p.out_labels, p.out_anchor, p.out_count = (
        scores.data_ptr(), boxes.data_ptr(), labels.data_ptr(), aidx.data_ptr(), count.data_ptr())
    if preselected:
        check(_lib.lib().vd3d_head_nms(C.byref(p), _stream()), 'vd3d_head_nms')
    else:
        check(_lib.lib().vd3d_head_postprocess(C.byref(p), _stream()), 'vd3d_head_postprocess')
    return scores, boxes, labels, aidx, count


def pack_detections(scores, boxes, labels, count, k, out=None):
    """Padded (scores [B,K], boxes [B,K,11], labels [B,K] i32, count [B] i32) -> one fp32 block [B, k+1, 13] (row k carries the
    count; rows past the count -- and, when K < k, rows past K -- are zero): vd3d_pack_detections, one launch, capturable in the
    step's hipGraph.  K and k are independent (KM3D's decode returns K = 100 rows; the gather record may be wider)."""
    _require_cuda(scores, boxes, labels, count, out)
    B, K = scores.shape
    k = int(k)
    assert boxes.shape == (B, K, 11) and labels.shape == (B, K) and count.shape == (B,)
    assert scores.dtype == boxes.dtype == torch.float32 and labels.dtype == count.dtype == torch.int32
    assert scores.is_contiguous() and boxes.is_contiguous() and labels.is_contiguous() and count.is_contiguous()
    if out is None:
        out = torch.empty((B, k + 1, 13), dtype=torch.float32, device=scores.device)
    assert out.shape == (B, k + 1, 13) and out.dtype == torch.float32 and out.is_contiguous()
    check(_lib.lib().vd3d_pack_detections(_p(scores), _p(boxes), _p(labels), _p(count), B, K, k, _p(out), _stream()), 'vd3d_pack_detections')
    return out


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms drop-in on the GPU: int64 keep indices in decreasing-score order."""
    _require_cuda(boxes, scores)
    n = boxes.shape[0]
    boxes = boxes.float().contiguous()
    scores = scores.float().contiguous()
    keep = torch.empty((max(n, 1),), dtype=torch.int32, device=boxes.device)
    count = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    check(_lib.lib().vd3d_nms(_p(boxes), _p(scores), n, float(iou_threshold), _p(keep), _p(count), None, _stream()), 'vd3d_nms')
    k = int(count.item())
    return keep[:k].long()


# --------------------------------------------------------------------------------------------- deformable conv
class PackedDCN:
    def __init__(self, w, Kpad, O, Cg, kh, kw, dtype):
        self.w, self.Kpad, self.O, self.Cg, self.kh, self.kw, self.dtype = w, Kpad, O, Cg, kh, kw, dtype


def pack_dcn_weight(weight, dtype):
    """OIHW fp32 -> [O][Kpad] tap-major in ``dtype`` (device kernel)."""
    _require_cuda(weight)
    O, Cg, kh, kw = weight.shape
    bke = 64 if is16(dtype) else 32
    Kpad = (kh * kw * Cg + bke - 1) // bke * bke
    w = weight.detach().float().contiguous()
    packed = torch.empty((O, Kpad), dtype=dtype, device=weight.device)
    check(_lib.lib().vd3d_dcn_pack_weight(_p(w), _p(packed), O, Cg, kh, kw, Kpad, dtype_code(dtype), _stream()), 'vd3d_dcn_pack_weight')
    return PackedDCN(packed, Kpad, O, Cg, kh, kw, dtype)


def _strides4(t, layout):
    """element strides {batch, channel, y, x} of a 4-D tensor given its logical layout ('nchw' or 'nhwc')."""
    if layout == 'nchw':
        return (t.stride(0), t.stride(1), t.stride(2), t.stride(3))
    return (t.stride(0), t.stride(3), t.stride(1), t.stride(2))


def deform_conv_general(x, pd, offset, mask, out, layout, bias=None, scale=None, shift=None, stride=(1, 1), padding=(0, 0),
                        dilation=(1, 1), groups=1, deformable_groups=1, mask_sigmoid=False, relu=False,
                        offset_layout=None, mask_layout=None):
    """vd3d_deform_conv on tensors of either layout.  x/out: activations (bf16|fp32, same dtype as the packed weight);
    offset/mask: fp32."""
    _require_cuda(x, offset, out)
    from ._lib import DcnParams
    if layout == 'nchw':
        B, Cc, H, W = x.shape
    else:
        B, H, W, Cc = x.shape
    assert x.dtype == pd.dtype == out.dtype and offset.dtype == torch.float32 and (mask is None or mask.dtype == torch.float32)
    p = DcnParams()
    p.in_, p.weight, p.out, p.offset = x.data_ptr(), pd.w.data_ptr(), out.data_ptr(), offset.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.scale = scale.data_ptr() if scale is not None else None
    p.shift = shift.data_ptr() if shift is not None else None
    p.mask = mask.data_ptr() if mask is not None else None
    p.B, p.C, p.H, p.W, p.O, p.kh, p.kw = B, Cc, H, W, pd.O, pd.kh, pd.kw
    p.stride_h, p.stride_w = stride
    p.pad_h, p.pad_w = padding
    p.dil_h, p.dil_w = dilation
    p.groups, p.deformable_groups, p.Kpad, p.dtype = groups, deformable_groups, pd.Kpad, dtype_code(x.dtype)
    p.mask_sigmoid, p.relu = int(mask_sigmoid), int(relu)
    for name, t, lay in (('in_strides', x, layout), ('out_strides', out, layout),
                         ('offset_strides', offset, offset_layout or layout), ('mask_strides', mask, mask_layout or layout)):
        s = _strides4(t, lay) if t is not None else (0, 0, 0, 0)
        setattr(p, name, (C.c_int64 * 4)(*s))
    check(_lib.lib().vd3d_deform_conv(C.byref(p), _stream()), 'vd3d_deform_conv')
    return out


def deform_columns(x, offset, mask, kernel, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mask_sigmoid=False, offset_layout='nhwc',
                   mask_layout='nhwc'):
    """Sampled, modulated DCN columns of NHWC ``x`` -> ``[B, Ho, Wo, kh*kw*C]`` in x's dtype (vd3d_deform_columns): the sampling half
    of a deformable conv whose output-channel count is large; contract with a 1x1 ``conv2d`` over K = kh*kw*C."""
    _require_cuda(x, offset, mask)
    from ._lib import DcnParams
    B, H, W, Cc = x.shape
    kh, kw = kernel
    Ho = (H + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
    assert offset.dtype == torch.float32 and (mask is None or mask.dtype == torch.float32) and x.stride(3) == 1
    cols = torch.empty((B, Ho, Wo, kh * kw * Cc), dtype=x.dtype, device=x.device)
    p = DcnParams()
    p.in_, p.offset = x.data_ptr(), offset.data_ptr()
    p.mask = mask.data_ptr() if mask is not None else None
    p.B, p.C, p.H, p.W, p.O, p.kh, p.kw = B, Cc, H, W, 0, kh, kw
    p.stride_h, p.stride_w = stride
    p.pad_h, p.pad_w = padding
    p.dil_h, p.dil_w = dilation
    p.groups, p.deformable_groups, p.Kpad, p.dtype = 1, 1, kh * kw * Cc, dtype_code(x.dtype)
    p.mask_sigmoid, p.relu = int(mask_sigmoid), 0
    for name, t, lay in (('in_strides', x, 'nhwc'), ('offset_strides', offset, offset_layout), ('mask_strides', mask, mask_layout)):
        s = _strides4(t, lay) if t is not None else (0, 0, 0, 0)
        setattr(p, name, (C.c_int64 * 4)(*s))
    check(_lib.lib().vd3d_deform_columns(C.byref(p), _p(cols), _stream()), 'vd3d_deform_columns')
    return cols


def deform_conv_forward_nchw(x, weight, bias, offset, mask, stride, padding, dilation, groups, deformable_groups, out=None):
    """The reference extension's call (NCHW fp32 contiguous tensors) through vd3d_deform_conv_forward.  ``out``: optional
    preallocated contiguous fp32 [B,O,Ho,Wo] written in place (the pybind surface's ``output`` argument)."""
    _require_cuda(x, weight, offset)
    x, weight, offset = x.float().contiguous(), weight.float().contiguous(), offset.float().contiguous()
    mask = mask.float().contiguous() if mask is not None else None
    bias = bias.float().contiguous() if bias is not None else None
    B, Cc, H, W = x.shape
    O, Cg, kh, kw = weight.shape
    Ho = (H + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
    assert offset.shape == (B, deformable_groups * 2 * kh * kw, Ho, Wo), (offset.shape, (B, deformable_groups * 2 * kh * kw, Ho, Wo))
    if out is None:
        out = torch.empty((B, O, Ho, Wo), dtype=torch.float32, device=x.device)
    assert out.shape == (B, O, Ho, Wo) and out.dtype == torch.float32 and out.is_contiguous() and out.is_cuda
    ws = torch.empty(_lib.lib().vd3d_deform_conv_workspace_bytes(O, Cc, groups, kh, kw), dtype=torch.uint8, device=x.device)
    check(_lib.lib().vd3d_deform_conv_forward(_p(x), _p(weight), _p(bias), _p(offset), _p(mask), _p(out), _p(ws), B, Cc, H, W, O, kh, kw,
                                              stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
                                              groups, deformable_groups, _stream()), 'vd3d_deform_conv_forward')
    return out


# --------------------------------------------------------------------------------------------- 3-channel image convs
def pack_image_conv(weight, bn, dtype, stride, pad):
    """k x k conv on a 3-channel image (ResNet stem 7x7/s2, DLA base layer 7x7/s1) as a (k x 1)-tap implicit GEMM over
    rows of 8 px x cpad channels of a zero-bordered NHWC-cpad image (vd3d_pack_image_nhwc).  cpad is chosen so that every
    output column starts a 16-byte aligned run: 4 for fp32 or stride 2, 8 for bf16 at stride 1."""
    O, I, kh, kw = weight.shape
    assert I == 3 and kh == kw and kw <= 8
    cpad = 4 if (dtype == torch.float32 or stride % 2 == 0) else 8
    dev = weight.device
    w = torch.zeros((O, kh, 8, cpad), dtype=torch.float32, device=dev)
    w[:, :, :kw, :3] = weight.detach().float().permute(0, 2, 3, 1)
    row = 8 * cpad
    K = kh * row
    bke = 64 if is16(dtype) else 32
    Kpad = (K + bke - 1) // bke * bke
    CoutPad = (O + 127) // 128 * 128
    packed = torch.zeros((CoutPad, Kpad), dtype=dtype, device=dev)
    packed[:O, :K] = w.reshape(O, K).to(dtype)
    scale, shift = fold_bn(None, bn, O, dev)
    pc = PackedConv(packed, scale, shift, row, O, kh, 1, stride, 0, 1, Kpad, CoutPad, dtype)
    pc.cpad, pc.k, pc.img_pad = cpad, kh, pad
    pc.w_frag7 = None
    if is16(dtype) and (kh, stride, pad) == (7, 1, 3) and O <= 16:
        # operand image of vd3d_image_conv7x7 (layout: include/vd3d.h): [ky][lane][e] <- w[o = l & 15][c = e & 3][ky][kx = 2 (l >> 4) + (e >> 2)]
        wf = torch.zeros((16, 4, 7, 8), dtype=torch.float32, device=dev)          # o, c, ky, kx (kx = 7 and c = 3 stay zero)
        wf[:O, :3, :, :7] = weight.detach().float()
        pc.w_frag7 = wf.reshape(16, 4, 7, 4, 2).permute(2, 3, 0, 4, 1).contiguous().to(dtype)   # ky, kq, o, kx & 1, c  ==  [7][64][8]
    return pc


def image_conv(img_nchw, pc, relu=True):
    _require_cuda(img_nchw)
    B, Cc, H, W = img_nchw.shape
    assert Cc == 3 and img_nchw.dtype == torch.float32 and img_nchw.is_contiguous()
    dtype, k, s, p, cpad = pc.dtype, pc.k, pc.stride, pc.img_pad, pc.cpad
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    if getattr(pc, 'w_frag7', None) is not None and pc.Cout % 4 == 0:
        # full-resolution 7x7 base layer: dedicated streaming kernel, no packed copy of the image
        out = torch.empty((B, H, W, pc.Cout), dtype=dtype, device=img_nchw.device)
        check(_lib.lib().vd3d_image_conv7x7(_p(img_nchw), _p(pc.w_frag7), _p(pc.scale) if pc.scale is not None else None, _p(pc.shift), _p(out),
                                            B, H, W, pc.Cout, pc.Cout, int(relu), dtype_code(dtype), _stream()), 'vd3d_image_conv7x7')
        return out
    pad_r = max(p, (Wo - 1) * s + 8 - W - p)
    Wp = W + p + pad_r
    if (Wp * cpad * packed_elem_size(dtype)) % 16:
        pad_r += 1
        Wp += 1
    pad_b = max(p, (Ho - 1) * s + k - H - p)
    Hp = H + p + pad_b
    packed = torch.empty((B, Hp, Wp, cpad), dtype=dtype, device=img_nchw.device)
    check(_lib.lib().vd3d_pack_image_nhwc(_p(img_nchw), _p(packed), B, H, W, p, pad_b, p, pad_r, cpad, dtype_code(dtype), _stream()),
          'vd3d_pack_image_nhwc')
    out = torch.empty((B, Ho, Wo, pc.Cout), dtype=dtype, device=img_nchw.device)
    q = ConvParams()
    q.in_, q.weight, q.out = packed.data_ptr(), pc.w.data_ptr(), out.data_ptr()
    q.scale = pc.scale.data_ptr() if pc.scale is not None else None
    q.shift = pc.shift.data_ptr()
    q.B, q.H, q.W, q.Cin = B, Hp, Wp, 8 * cpad
    q.in_pix_stride, q.in_row_stride, q.in_batch_stride = cpad, Wp * cpad, Hp * Wp * cpad
    q.in_bytes = _bytes_from(packed)
    q.Ho, q.Wo, q.Cout = Ho, Wo, pc.Cout
    q.out_pix_stride = pc.Cout
    q.kh, q.kw, q.stride, q.pad, q.dil = k, 1, s, 0, 1
    q.Kpad, q.CoutPad, q.relu = pc.Kpad, pc.CoutPad, int(relu)
    q.dtype, q.out_f32 = dtype_code(dtype), 0
    check(_lib.lib().vd3d_conv2d_igemm(C.byref(q), _stream()), 'vd3d_conv2d_igemm(image)')
    return out


def packed_elem_size(dtype):
    return 2 if is16(dtype) else 4


def maxpool2x2(x):
    _require_cuda(x)
    B, H, W, Cc = x.shape
    out = torch.empty((B, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
    assert _dense_pixels(x)
    check(_lib.lib().vd3d_maxpool2x2(_p(x), _p(out), B, H, W, Cc, x.stride(2), out.stride(2), dtype_code(x.dtype), _stream()),
          'vd3d_maxpool2x2')
    return out


def dwconv_transpose(x, weight_kk_c, f, add=None):
    """Depth-wise ConvTranspose2d(kernel 2f, stride f, pad f//2) (+ add).  weight_kk_c: [(2f)^2][C] fp32."""
    _require_cuda(x, weight_kk_c, add)
    B, H, W, Cc = x.shape
    K, pad = 2 * f, f // 2
    Ho, Wo = (H - 1) * f - 2 * pad + K, (W - 1) * f - 2 * pad + K
    out = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    assert _dense_pixels(x) and (add is None or (add.shape == out.shape and _dense_pixels(add)))
    check(_lib.lib().vd3d_dwconv_transpose(_p(x), _p(weight_kk_c), _p(add), _p(out), B, H, W, Cc, f, x.stride(2),
                                           add.stride(2) if add is not None else 0, out.stride(2), dtype_code(x.dtype), _stream()),
          'vd3d_dwconv_transpose')
    return out


def post_opt_batched(boxes, labels, counts, P2s, clamp_wh=(1280.0, 288.0), min_depth=3.0, target_label=0):
    """In-place hill-climbing yaw refinement of a padded batch (heads/detection_3d_head.py:294-308 +
    lib/fast_utils/hill_climbing.py): boxes [B,K,11] fp32, labels [B,K] int32, counts [B] int32 or None, P2s [B,3,4]."""
    _require_cuda(boxes, labels, P2s)
    assert boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.dim() == 3 and boxes.shape[2] == 11
    assert labels.dtype == torch.int32 and labels.is_contiguous() and labels.shape == boxes.shape[:2]
    B, K = labels.shape
    P2s = P2s.float().contiguous()
    assert P2s.shape == (B, 3, 4)
    if counts is not None:
        assert counts.dtype == torch.int32 and counts.shape == (B,) and counts.is_contiguous()
    check(_lib.lib().vd3d_post_opt(_p(boxes), _p(labels), _p(counts) if counts is not None else None, _p(P2s), B, K,
                                   float(clamp_wh[0]), float(clamp_wh[1]), float(min_depth), int(target_label), _stream()),
          'vd3d_post_opt')
    return boxes


def kitti_postpath(boxes, counts, P2s, xform):
    """Batched post-path geometry (pipelines/evaluators.py:112-129): boxes [B,K,11] fp32, counts [B] int32 or None,
    P2s [B,3,4], xform [B,4] = (shift_left, shift_top, scale_x, scale_y).  Returns [B,K,12] fp32 rows
    (x1,y1,x2,y2 in the original image, x3d, y_bottom, z, w, h, l, alpha, theta)."""
    _require_cuda(boxes, P2s, xform)
    assert boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.dim() == 3 and boxes.shape[2] == 11
    B, K = boxes.shape[:2]
    P2s = P2s.float().contiguous()
    xform = xform.float().contiguous()
    assert P2s.shape == (B, 3, 4) and xform.shape == (B, 4)
    if counts is not None:
        assert counts.dtype == torch.int32 and counts.shape == (B,) and counts.is_contiguous()
    out = torch.empty((B, K, 12), dtype=torch.float32, device=boxes.device)
    check(_lib.lib().vd3d_kitti_postpath(_p(boxes), _p(counts) if counts is not None else None, _p(P2s), _p(xform), _p(out), B, K,
                                         _stream()), 'vd3d_kitti_postpath')
    return out


def resized_shape(Hs, Ws, crop_top, size):
    """(Hr, Wr, scale) of Resize(size, preserve_aspect_ratio=True) after CropTop (stereo_augmentator.py:75-80)."""
    import numpy as np
    hc = Hs - crop_top
    scale = size[0] / hc
    return int(np.round(hc * scale).astype(int)), int(np.round(Ws * scale).astype(int)), scale


def adjust_calib(P, crop_top, scale):
    """CropTop (:236-242) then Resize (:114-120) applied to a [3,4] projection matrix, float64 like the reference's numpy."""
    import numpy as np
    P = np.array(P, dtype=np.float64, copy=True)
    P[1, 2] = P[1, 2] - crop_top
    P[1, 3] = P[1, 3] - crop_top * P[2, 3]
    P[0, :] = P[0, :] * scale
    P[1, :] = P[1, :] * scale
    return P


def preprocess_images(frames_u8, crop_top, size, mean, std, packed=False):
    """uint8 HWC frames (list of [Hs,Ws,3] cuda tensors, or one [B,Hs,Ws,3]) -> the network input: fp32 NCHW [B,3,H,W]
    (``packed=False``: what the reference's pipeline + collate_fn produce) or the bordered NHWC4 bf16 image of the fused stem
    (``packed=True``)."""
    frames = list(frames_u8)
    _require_cuda(*frames)
    H, W = int(size[0]), int(size[1])
    B = len(frames)
    dev = frames[0].device
    out = (torch.empty((B, H + 6, W + 8, 4), dtype=torch.bfloat16, device=dev) if packed
           else torch.empty((B, 3, H, W), dtype=torch.float32, device=dev))
    m = (C.c_float * 3)(*[float(v) for v in mean])
    sd = (C.c_float * 3)(*[float(v) for v in std])
    for b, f in enumerate(frames):
        assert f.dtype == torch.uint8 and f.dim() == 3 and f.shape[2] == 3 and f.is_contiguous()
        Hs, Ws = int(f.shape[0]), int(f.shape[1])
        Hr, Wr, _ = resized_shape(Hs, Ws, crop_top, size)
        check(_lib.lib().vd3d_preprocess_image(_p(f), Hs, Ws, int(crop_top), Hr, Wr, None if packed else _p(out[b]),
                                               _p(out[b]) if packed else None, H, W, m, sd, _stream()), 'vd3d_preprocess_image')
    return out


def km3d_head_fused(x, pc_first, w2_packed, b2, n_out):
    """Fused KM3D head (vd3d_km3d_head_fused): x NHWC bf16 [B,H,W,Cin], pc_first = packed concatenation of the nine 3x3 convs
    (Cout = 256 * heads, bias in shift), w2_packed [heads,32,256] bf16, b2 [heads,32] fp32 -> list of fp32 [B,H,W,n_h]."""
    _require_cuda(x, pc_first.w, w2_packed, b2)
    B, H, W, Cx = x.shape
    heads = len(n_out)
    assert is16(x.dtype) and pc_first.dtype == x.dtype and Cx == pc_first.Cin and pc_first.Cout == 256 * heads
    assert w2_packed.shape == (heads, 32, 256) and w2_packed.dtype == x.dtype and w2_packed.is_contiguous()
    assert b2.shape == (heads, 32) and b2.dtype == torch.float32 and b2.is_contiguous() and pc_first.scale is None
    outs = [torch.empty((B, H, W, int(n)), dtype=torch.float32, device=x.device) for n in n_out]
    ips, irs, ibs = _nhwc_strides(x)
    p = ConvParams()
    p.in_, p.weight, p.out = x.data_ptr(), pc_first.w.data_ptr(), outs[0].data_ptr()
    p.scale, p.shift = None, pc_first.shift.data_ptr()
    p.B, p.H, p.W, p.Cin = B, H, W, pc_first.Cin
    p.in_pix_stride, p.in_row_stride, p.in_batch_stride = ips, irs, ibs
    p.in_bytes = min(_bytes_from(x), ((B - 1) * ibs + (H - 1) * irs + (W - 1) * ips + Cx) * x.element_size())
    p.Ho, p.Wo, p.Cout = H, W, pc_first.Cout
    p.out_pix_stride = pc_first.Cout
    p.kh, p.kw, p.stride, p.pad, p.dil = 3, 3, 1, 1, 1
    p.Kpad, p.CoutPad, p.relu = pc_first.Kpad, pc_first.CoutPad, 1
    p.dtype, p.out_f32 = dtype_code(x.dtype), 0
    ptrs = (C.c_void_p * heads)(*[o.data_ptr() for o in outs])
    ns = (C.c_int32 * heads)(*[int(n) for n in n_out])
    check(_lib.lib().vd3d_km3d_head_fused(C.byref(p), _p(w2_packed), _p(b2), ptrs, ns, heads, _stream()), 'vd3d_km3d_head_fused')
    return outs
