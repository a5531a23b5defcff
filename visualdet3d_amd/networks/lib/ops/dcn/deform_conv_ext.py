"""``deform_conv_ext`` -- the reference's pybind11 extension module, name for name (lib/ops/dcn/src/deform_conv_ext.cpp:149-163),
on the HIP kernel behind ``vd3d_deform_conv_forward`` (include/vd3d.h).  A maintainer of the reference replaces
``from . import deform_conv_ext`` (lib/ops/dcn/deform_conv.py:50) by an import of this module and nothing else changes:

  * ``modulated_deform_conv_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w, stride_h,
    stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, with_bias) -> None``
    (deform_conv_ext.cpp:106-124 -> src/cuda/deform_conv_cuda.cpp:491-570): writes ``output`` IN PLACE; ``ones`` / ``columns``
    are the two empty scratch tensors the caller passes (deform_conv.py:181-186) -- the reference re-binds them locally
    (deform_conv_cuda.cpp:527,533), so the caller's stay empty; here they are simply never needed (the HIP kernel gathers
    straight into the MFMA operand, no ``columns`` matrix exists);
  * ``deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
    deformable_group, im2col_step) -> 1`` (deform_conv_ext.cpp:51-67 -> deform_conv_cuda.cpp:152-260): note the reference's
    **W-before-H** argument order (deform_conv.py:90-95);
  * the three backward entry points are training-only: ``NotImplementedError``.

Error behaviour follows the reference: CPU tensors -> ``RuntimeError('... is not implemented on CPU')``
(deform_conv_ext.cpp:66,123 ``AT_ERROR``); non-contiguous input / weight, kernel-size or channel mismatches ->
``RuntimeError`` with the reference's message (``TORCH_CHECK`` in deform_conv_cuda.cpp:62-150,498-513).  Kernels run on the
current torch stream (the reference: ``at::cuda::getCurrentCUDAStream()``, deform_conv_cuda_kernel.cu:788).
Inputs are fp32 or fp16/fp64 in the reference's dispatch; this boundary computes in fp32 (other float dtypes are converted
and the result cast back into ``output``)."""
import torch

from ..... import hip_ops as ops


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw):
    return (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1


def _write(output, result):
    """The reference ``view``s / resizes ``output`` to [B, O, Ho, Wo] and fills it; the caller allocated exactly that."""
    if tuple(output.shape) != tuple(result.shape):
        output.resize_(result.shape)
    output.copy_(result)


def modulated_deform_conv_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w, stride_h, stride_w,
                                  pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
    if not input.is_cuda:
        raise RuntimeError('modulated deform conv is not implemented on CPU')
    _check(input.is_contiguous(), 'input tensor has to be contiguous')
    _check(weight.is_contiguous(), 'weight tensor has to be contiguous')
    B, Cc, H, W = input.shape
    O, Cg, kh_, kw_ = weight.shape
    _check(kh_ == kernel_h and kw_ == kernel_w,
           'Input shape and kernel shape wont match: (%d x %d vs %d x %d).' % (kernel_h, kernel_w, kh_, kw_))
    _check(Cc == Cg * group, 'Input shape and kernel channels wont match: (%d vs %d).' % (Cc, Cg * group))
    Ho, Wo = _out_hw(H, W, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w)
    K = kernel_h * kernel_w
    _check(tuple(offset.shape) == (B, deformable_group * 2 * K, Ho, Wo) and tuple(mask.shape) == (B, deformable_group * K, Ho, Wo),
           'offset / mask shape does not match the output size [%d, %d] and deformable_group %d' % (Ho, Wo, deformable_group))
    # offset / mask contiguity is assumed, not checked, by the reference; contiguous copies are made if needed
    res = ops.deform_conv_forward_nchw(input, weight, bias if with_bias else None, offset, mask, (stride_h, stride_w), (pad_h, pad_w),
                                       (dilation_h, dilation_w), group, deformable_group,
                                       out=output if (output.dtype == torch.float32 and output.is_contiguous()
                                                      and tuple(output.shape) == (B, O, Ho, Wo)) else None)
    if res.data_ptr() != output.data_ptr():
        _write(output, res)
    return None


def deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
                        deformable_group, im2col_step):
    if not input.is_cuda:
        raise RuntimeError('deform conv is not implemented on CPU')
    # shape_check (deform_conv_cuda.cpp:62-150)
    _check(weight.dim() == 4, '4D weight tensor (nOutputPlane,nInputPlane,kH,kW) expected, but got: %d' % weight.dim())
    _check(weight.is_contiguous(), 'weight tensor has to be contiguous')
    _check(kW > 0 and kH > 0, 'kernel size should be greater than zero, but got kH: %d kW: %d' % (kH, kW))
    _check(weight.size(2) == kH and weight.size(3) == kW,
           'kernel size should be consistent with weight, but got kH: %d kW: %d weight.size(2): %d, weight.size(3): %d'
           % (kH, kW, weight.size(2), weight.size(3)))
    _check(dW > 0 and dH > 0, 'stride should be greater than zero, but got dH: %d dW: %d' % (dH, dW))
    _check(dilationW > 0 and dilationH > 0, 'dilation should be greater than 0, but got dilationH: %d dilationW: %d' % (dilationH, dilationW))
    _check(input.dim() in (3, 4), '3D or 4D input tensor expected but got: %d' % input.dim())
    batched = input.dim() == 4
    if not batched:                                    # deform_conv_cuda.cpp:176-183: a 3D input is one sample
        input, offset = input.unsqueeze(0), offset.unsqueeze(0)
    B, Cc, H, W = input.shape
    O, Cg = weight.size(0), weight.size(1)
    _check((Cg * group) % deformable_group == 0, 'input channels must divide deformable group size')
    Ho, Wo = _out_hw(H, W, kH, kW, dH, dW, padH, padW, dilationH, dilationW)
    _check(Ho >= 1 and Wo >= 1, 'Given input size: (%d x %d x %d). Calculated output size: (%d x %d x %d). Output size is too small'
           % (Cg * group, H, W, O, Ho, Wo))
    _check(Cc == Cg * group, 'invalid number of input planes, expected: %d, but got: %d' % (Cg * group, Cc))
    _check(H >= kH and W >= kW, 'input image is smaller than kernel')
    _check(offset.size(2) == Ho and offset.size(3) == Wo,
           'invalid spatial size of offset, expected height: %d width: %d, but got height: %d width: %d' % (Ho, Wo, offset.size(2), offset.size(3)))
    _check(offset.size(1) == deformable_group * 2 * kH * kW, 'invalid number of channels of offset')
    _check(offset.size(0) == B, 'invalid batch size of offset')                     # deform_conv_cuda.cpp:197
    _check(B % im2col_step == 0, 'im2col step must divide batchsize')               # deform_conv.py:88-89 / .cpp:186
    input = input.contiguous()                                                      # deform_conv_cuda.cpp:173-175
    res = ops.deform_conv_forward_nchw(input, weight, None, offset, None, (dH, dW), (padH, padW), (dilationH, dilationW), group,
                                       deformable_group,
                                       out=output if (batched and output.dtype == torch.float32 and output.is_contiguous()
                                                      and tuple(output.shape) == (B, O, Ho, Wo)) else None)
    if not batched:
        res = res[0]
    if res.data_ptr() != output.data_ptr():
        _write(output, res)
    return 1


def deform_conv_backward_input(*args, **kwargs):
    raise NotImplementedError('deform_conv_backward_input: training is out of scope of the MI355X inference path (SURVEY.md 8b)')


def deform_conv_backward_parameters(*args, **kwargs):
    raise NotImplementedError('deform_conv_backward_parameters: training is out of scope of the MI355X inference path (SURVEY.md 8b)')


def modulated_deform_conv_backward(*args, **kwargs):
    raise NotImplementedError('modulated_deform_conv_backward: training is out of scope of the MI355X inference path (SURVEY.md 8b)')
