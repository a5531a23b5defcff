"""GPU: deformable convolution (csrc/deform_conv.hip) vs golden outputs of the reference's own im2col code
(tests/golden/dcn_cases.npz) and vs the oracle for the module-level / NHWC engine paths."""
import numpy as np
import pytest
import torch

from oracle import dcn_ref
from oracle.make_golden_native import DCN_CASES, dcn_inputs
from tests.common import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', list(DCN_CASES))
def test_functional_matches_reference_golden(name):
    from visualdet3d_amd.networks.lib.ops.dcn import deform_conv, modulated_deform_conv
    g = load_golden('dcn_cases')
    x, off, mask, w, bias, kw = dcn_inputs(name)
    if mask is not None:
        out = modulated_deform_conv(x.cuda(), off.cuda(), mask.cuda(), w.cuda(), bias.cuda(), kw['stride'], kw['padding'], kw['dilation'],
                                    kw['groups'], kw['deformable_groups'])
    else:
        out = deform_conv(x.cuda(), off.cuda(), w.cuda(), kw['stride'], kw['padding'], kw['dilation'], kw['groups'], kw['deformable_groups'])
    want = g[name + '_out']
    err = np.abs(out.cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 1e-5, err


def test_cpu_tensors_are_rejected_like_the_reference():
    from visualdet3d_amd.networks.lib.ops.dcn import modulated_deform_conv
    x, off, mask, w, bias, kw = dcn_inputs('v2_3x3')
    with pytest.raises(NotImplementedError):
        modulated_deform_conv(x, off, mask, w, bias, 1, 1, 1, 1, 1)


def _pack_module(C, O, seed):
    from visualdet3d_amd.networks.lib.ops import ModulatedDeformConvPack
    m = ModulatedDeformConvPack(C, O, 3, stride=1, padding=1, dilation=1, deformable_groups=1)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.1)
        m.bias.copy_(torch.randn(O, generator=g) * 0.1)
        m.conv_offset.weight.copy_(torch.randn(m.conv_offset.weight.shape, generator=g) * 0.05)
        m.conv_offset.bias.copy_(torch.randn(27, generator=g) * 0.5)
    return m


def _ref_pack_forward(m, x, rnd=None):
    import torch.nn.functional as F
    out = F.conv2d(x, m.conv_offset.weight, m.conv_offset.bias, 1, 1)
    o1, o2, mask = torch.chunk(out, 3, dim=1)
    return dcn_ref.deform_conv_forward(x, torch.cat((o1, o2), 1), torch.sigmoid(mask), m.weight, m.bias, 1, 1, 1, 1, 1, rnd=rnd)


def test_pack_module_nchw_forward():
    m = _pack_module(24, 40, 0)
    x = torch.randn(2, 24, 10, 14, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = _ref_pack_forward(m, x)
        got = m.cuda()(x.cuda()).cpu()
    assert ((got - want).abs().max() / want.abs().max()).item() < 2e-5
    # legacy checkpoints (version < 2) name the offset conv "<prefix>_offset"
    sd = {('conv_offset.' + k[len('conv_offset.'):] if False else k): v for k, v in m.state_dict().items()}
    assert set(sd) == {'weight', 'bias', 'conv_offset.weight', 'conv_offset.bias'}


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 5e-5), (torch.bfloat16, 2e-2), (torch.float16, 2.5e-3)])
def test_pack_module_nhwc_engine_path_with_bn_relu(dtype, tol):
    C, O = 64, 64
    m = _pack_module(C, O, 2)
    bn = torch.nn.BatchNorm2d(O).eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(O, generator=g) + 0.5); bn.bias.copy_(torch.randn(O, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(O, generator=g) * 0.1); bn.running_var.copy_(torch.rand(O, generator=g) + 0.5)
    x = torch.randn(2, C, 9, 12, generator=g)
    rnd = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else None
    with torch.no_grad():
        xr = rnd(x) if rnd else x
        # offsets come from the (rounded-input, rounded-weight) conv in the engine; sampling positions are fp32
        import torch.nn.functional as F
        wo = rnd(m.conv_offset.weight) if rnd else m.conv_offset.weight
        out = F.conv2d(xr, wo, m.conv_offset.bias, 1, 1)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        y = dcn_ref.deform_conv_forward(xr, torch.cat((o1, o2), 1), torch.sigmoid(mask), m.weight, m.bias, 1, 1, 1, 1, 1, rnd=rnd)
        want = torch.relu(bn(y))
        m = m.cuda(); bn = bn.cuda()
        got = m.forward_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda().to(dtype), bn=bn, relu=True)
    got = got.float().cpu().permute(0, 3, 1, 2)
    assert ((got - want).abs().max() / want.abs().max()).item() < tol


@pytest.mark.parametrize('name', list(DCN_CASES))
def test_pybind_shaped_ext_surface_with_reference_argument_order(name):
    """``deform_conv_ext.{modulated_deform_conv_forward, deform_conv_forward}`` called exactly as the reference's autograd
    Functions call the pybind module (lib/ops/dcn/deform_conv.py:90-95,181-186): caller-allocated ``output`` written in place,
    two empty scratch tensors that stay empty, W-before-H order for DCNv1; results = the reference's own im2col goldens."""
    from visualdet3d_amd.networks.lib.ops.dcn import deform_conv_ext
    g = load_golden('dcn_cases')
    x, off, mask, w, bias, kw = dcn_inputs(name)
    x, off, w = x.cuda(), off.cuda(), w.cuda()
    st, pad, dil = kw['stride'], kw['padding'], kw['dilation']
    want = g[name + '_out']
    output = torch.full(want.shape, 123.0, device='cuda')
    bufs = [x.new_empty(0), x.new_empty(0)]
    ptr = output.data_ptr()
    if mask is not None:
        ret = deform_conv_ext.modulated_deform_conv_forward(x, w, bias.cuda(), bufs[0], off, mask.cuda(), output, bufs[1], w.shape[2], w.shape[3],
                                                            st, st, pad, pad, dil, dil, kw['groups'], kw['deformable_groups'], True)
        assert ret is None
    else:
        ret = deform_conv_ext.deform_conv_forward(x, w, off, output, bufs[0], bufs[1], w.size(3), w.size(2), st, st, pad, pad, dil, dil,
                                                  kw['groups'], kw['deformable_groups'], min(64, x.shape[0]))
        assert ret == 1
    assert output.data_ptr() == ptr and bufs[0].numel() == 0 and bufs[1].numel() == 0
    err = np.abs(output.cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 1e-5, err


def test_ext_surface_error_behaviour_and_asymmetric_geometry():
    """CPU tensors / bad shapes raise RuntimeError like the reference's AT_ERROR / TORCH_CHECK; backward names raise
    NotImplementedError; H != W strides / pads go to the right axes (W-before-H for v1, H-before-W for v2) -- checked against
    the oracle (oracle/dcn_ref.py) on an asymmetric case the symmetric goldens cannot distinguish."""
    from visualdet3d_amd.networks.lib.ops.dcn import deform_conv_ext
    x, off, mask, w, bias, kw = dcn_inputs('v2_3x3')
    out = torch.empty(2, 24, 9, 13)
    with pytest.raises(RuntimeError, match='not implemented on CPU'):
        deform_conv_ext.modulated_deform_conv_forward(x, w, bias, x.new_empty(0), off, mask, out, x.new_empty(0), 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, True)
    with pytest.raises(RuntimeError, match='not implemented on CPU'):
        deform_conv_ext.deform_conv_forward(x, w, off, out, x.new_empty(0), x.new_empty(0), 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 2)
    xc, wc, oc, mc, bc = x.cuda(), w.cuda(), off.cuda(), mask.cuda(), bias.cuda()
    outc = torch.empty(2, 24, 9, 13, device='cuda')
    with pytest.raises(RuntimeError, match='kernel shape wont match'):
        deform_conv_ext.modulated_deform_conv_forward(xc, wc, bc, xc.new_empty(0), oc, mc, outc, xc.new_empty(0), 5, 3, 1, 1, 1, 1, 1, 1, 1, 1, True)
    with pytest.raises(RuntimeError, match='has to be contiguous'):
        deform_conv_ext.modulated_deform_conv_forward(xc.transpose(2, 3), wc, bc, xc.new_empty(0), oc, mc, outc, xc.new_empty(0), 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, True)
    with pytest.raises(RuntimeError, match='invalid number of channels of offset'):
        deform_conv_ext.deform_conv_forward(xc, wc, oc[:, :10], outc, xc.new_empty(0), xc.new_empty(0), 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 2)
    for fn in (deform_conv_ext.deform_conv_backward_input, deform_conv_ext.deform_conv_backward_parameters,
               deform_conv_ext.modulated_deform_conv_backward):
        with pytest.raises(NotImplementedError):
            fn()
    # asymmetric geometry: stride (2, 1), pad (0, 2), dilation (1, 2), 3 x 3 kernel
    gg = torch.Generator().manual_seed(5)
    B, C_, H, W, O = 2, 8, 11, 9, 12
    sh, sw, ph, pw, dh, dw = 2, 1, 0, 2, 1, 2
    Ho, Wo = (H + 2 * ph - (dh * 2 + 1)) // sh + 1, (W + 2 * pw - (dw * 2 + 1)) // sw + 1
    x = torch.randn(B, C_, H, W, generator=gg)
    w = torch.randn(O, C_, 3, 3, generator=gg) * 0.2
    off = torch.randn(B, 18, Ho, Wo, generator=gg) * 1.5
    mask = torch.rand(B, 9, Ho, Wo, generator=gg)
    bias = torch.randn(O, generator=gg)
    want2 = dcn_ref.deform_conv_forward(x, off, mask, w, bias, (sh, sw), (ph, pw), (dh, dw), 1, 1)
    want1 = dcn_ref.deform_conv_forward(x, off, None, w, None, (sh, sw), (ph, pw), (dh, dw), 1, 1)
    o2 = torch.empty(B, O, Ho, Wo, device='cuda')
    deform_conv_ext.modulated_deform_conv_forward(x.cuda(), w.cuda(), bias.cuda(), x.new_empty(0).cuda(), off.cuda(), mask.cuda(), o2,
                                                  x.new_empty(0).cuda(), 3, 3, sh, sw, ph, pw, dh, dw, 1, 1, True)
    o1 = torch.empty(B, O, Ho, Wo, device='cuda')
    deform_conv_ext.deform_conv_forward(x.cuda(), w.cuda(), off.cuda(), o1, x.new_empty(0).cuda(), x.new_empty(0).cuda(), 3, 3, sw, sh, pw, ph, dw, dh,
                                        1, 1, 2)
    assert ((o2.cpu() - want2).abs().max() / want2.abs().max()).item() < 1e-5
    assert ((o1.cpu() - want1).abs().max() / want1.abs().max()).item() < 1e-5


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 5e-5), (torch.bfloat16, 2e-2), (torch.float16, 2.5e-3)])
def test_pack_module_nhwc_columns_route_for_many_output_channels(dtype, tol):
    """out_channels > 256 (the 2176 -> 2176 DCNv2 of the stereo base head): sampled columns written once (vd3d_deform_columns) +
    1x1 strip GEMM, against the oracle and against the fused kernel forced on the same module (same rounding points)."""
    C, O = 64, 320
    m = _pack_module(C, O, 5)
    bn = torch.nn.BatchNorm2d(O).eval()
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(O, generator=g) + 0.5); bn.bias.copy_(torch.randn(O, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(O, generator=g) * 0.1); bn.running_var.copy_(torch.rand(O, generator=g) + 0.5)
    x = torch.randn(2, C, 11, 13, generator=g)
    rnd = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else None
    with torch.no_grad():
        import torch.nn.functional as F
        xr = rnd(x) if rnd else x
        wo = rnd(m.conv_offset.weight) if rnd else m.conv_offset.weight
        out = F.conv2d(xr, wo, m.conv_offset.bias, 1, 1)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        y = dcn_ref.deform_conv_forward(xr, torch.cat((o1, o2), 1), torch.sigmoid(mask), m.weight, m.bias, 1, 1, 1, 1, 1, rnd=rnd)
        want = torch.relu(bn(y))
        m = m.cuda(); bn = bn.cuda()
        xin = x.permute(0, 2, 3, 1).contiguous().cuda().to(dtype)
        assert m.columns_above == 256
        got = m.forward_nhwc(xin, bn=bn, relu=True)
        m.columns_above = 10 ** 9                      # force the fused kernel
        fused_out = m.forward_nhwc(xin, bn=bn, relu=True)
    a = got.float().cpu().permute(0, 3, 1, 2)
    assert ((a - want).abs().max() / want.abs().max()).item() < tol
    assert ((got.float() - fused_out.float()).abs().max() / fused_out.float().abs().max()).item() < (1e-5 if dtype == torch.float32 else 2.0 ** -7)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_deform_columns_wave_kernel_matches_thread_per_vector_kernel(dtype):
    """vd3d_deform_columns has two kernels: wave-per-sample (inputs < 2 GiB; geometry once per (pixel, tap)) and the generic
    thread-per-vector one (VD3D_DCN_COLUMNS_GENERIC=1 forces it).  Same sampling arithmetic: fp32 equal to round-off, the 16-bit
    formats equal except where the folded modulation (sum (w m) x vs (sum w x) m) flips a rounding tie (<= 1 ulp, rare).
    Odd sizes: C = 72 channels (ragged last channel sweep), 7 x 9 pixels x 9 taps = 567 samples (ragged last wave)."""
    from visualdet3d_amd import _lib, hip_ops as ops
    g = torch.Generator().manual_seed(11)
    B, H, W, C = 2, 7, 9, 72
    x = torch.randn(B, H, W, C, generator=g).cuda().to(dtype)
    logits = (torch.randn(B, H, W, 32, generator=g) * 2.0).cuda()
    logits[0, 0, 0, :18] = 40.0                       # samples far outside the image -> zero columns
    cols = ops.deform_columns(x, logits[..., :18], logits[..., 18:27], (3, 3), (1, 1), (1, 1), (1, 1), mask_sigmoid=True)
    with _lib.test_switch('VD3D_DCN_COLUMNS_GENERIC'):
        ref = ops.deform_columns(x, logits[..., :18], logits[..., 18:27], (3, 3), (1, 1), (1, 1), (1, 1), mask_sigmoid=True)
    torch.cuda.synchronize()
    assert cols.shape == ref.shape and bool((cols[0, 0, 0] == 0).all())
    d = (cols.float() - ref.float()).abs()
    if dtype == torch.float32:
        assert (d.max() / ref.float().abs().max()).item() < 1e-6
    else:
        ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
        assert bool((d <= ref.float().abs() * ulp + 1e-6).all())
        assert (d > 0).float().mean().item() < 0.02


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,W,sigma', [(2, 21, 37, 0.5), (1, 64, 120, 0.5), (2, 16, 24, 4.0), (1, 40, 56, 1.7), (3, 8, 8, 0.3), (2, 33, 50, 40.0)])
def test_64_to_64_kernel_views_layouts_and_oracle(dtype, B, H, W, sigma):
    """The 64 -> 64 launch shape of KM3D's full-resolution DLA-Up nodes (dcn_nhwc_kernel<T, 64>; the alternative kernels of rounds 3 - 5 were
    removed from the library, see csrc/deform_conv.hip and profiles/r05_dcn_geo64_experiment.txt): sigma = spread of the learned offsets in pixels (0.3 ... 4.0, one sample far
    outside the image); ragged pixel counts, image borders, bias + folded BN + ReLU epilogue; a channel-slice INPUT view and a channel-slice
    OUTPUT view give the same bits as dense tensors; the logits as one packed [pixel][32] tensor (what the offset conv writes: staged through LDS)
    and as separate 18- / 9-channel tensors (the generic loader) give the same bits; and the result against the oracle (pinned to the reference's
    own im2col code)."""
    from visualdet3d_amd import _lib, hip_ops as ops
    g = torch.Generator().manual_seed(int(B * 1000 + H * 10 + sigma * 7))
    C = O = 64
    x = torch.randn(B, H, W, C, generator=g)
    wt = torch.randn(O, C, 3, 3, generator=g) * 0.05
    off = torch.randn(B, H, W, 18, generator=g) * sigma
    off[0, 0, 0, :] = 50.0                                         # far outside the image: the sample contributes nothing
    msk = torch.randn(B, H, W, 9, generator=g) * 2.0
    bias, scale, shift = torch.randn(O, generator=g) * 0.1, torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g) * 0.1
    xd = x.cuda().to(dtype)
    pd = ops.pack_dcn_weight(wt.cuda(), dtype)
    logits = torch.cat([off, msk, torch.zeros(B, H, W, 5)], dim=3).cuda()     # (o1 | o2 | mask | pad) like the offset conv writes it
    kw = dict(bias=bias.cuda(), scale=scale.cuda(), shift=shift.cuda(), stride=(1, 1), padding=(1, 1), dilation=(1, 1), mask_sigmoid=True, relu=True)

    def run(xin, out, o1=logits[..., :18], m1=logits[..., 18:27]):
        return ops.deform_conv_general(xin, pd, o1, m1, out, 'nhwc', **kw)

    b = run(xd, torch.empty((B, H, W, O), dtype=dtype, device='cuda'))
    buf = torch.full((B, H, W, 128), 3.0, dtype=dtype, device='cuda')
    a = run(xd, buf[..., 64:])                                                # channel-slice output view
    xwide = torch.full((B, H, W, 128), 9.0, dtype=dtype, device='cuda')
    xwide[..., 64:] = xd
    bs = run(xwide[..., 64:], torch.empty((B, H, W, O), dtype=dtype, device='cuda'))     # channel-slice input view (pixel stride 128 elements)
    bg = run(xd, torch.empty((B, H, W, O), dtype=dtype, device='cuda'), logits[..., :18].contiguous(), logits[..., 18:27].contiguous())
    with _lib.test_switch('VD3D_DCN_NO_LSTAGE'):
        bn = run(xd, torch.empty((B, H, W, O), dtype=dtype, device='cuda'))
    torch.cuda.synchronize()
    assert torch.equal(a, b) and bool((buf[..., :64] == 3.0).all()), 'channel-slice output view'
    assert torch.equal(bs, b), 'channel-slice input view'
    assert torch.equal(bg, b) and torch.equal(bn, b), 'logit layouts / loaders differ'
    rnd = lambda t: t.to(dtype).float()                             # noqa: E731
    y = dcn_ref.deform_conv_forward(rnd(x).permute(0, 3, 1, 2), off.permute(0, 3, 1, 2), torch.sigmoid(msk).permute(0, 3, 1, 2), wt, bias,
                                    1, 1, 1, 1, 1, rnd=rnd)
    want = torch.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    got = a.float().cpu().permute(0, 3, 1, 2)
    assert ((got - want).abs().max() / want.abs().max()).item() < (2e-2 if dtype == torch.bfloat16 else 2.5e-3)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('C,O,B,H,W', [(64, 64, 2, 21, 37), (128, 64, 1, 16, 24), (256, 256, 1, 9, 13), (64, 128, 2, 8, 8)])
def test_logit_staging_of_the_gather_kernel_is_bit_identical(dtype, C, O, B, H, W):
    """dcn_nhwc_kernel stages the tile's offset / mask logits through LDS with coalesced 16-byte loads when they come in the engine's own
    layout (32 fp32 per pixel: offsets | mask | pad); VD3D_DCN_NO_LSTAGE=1 reads them lane = pixel from global memory as before.  The same
    values either way: bit-identical outputs, ragged last tile and all three block widths."""
    from visualdet3d_amd import _lib, hip_ops as ops
    g = torch.Generator().manual_seed(C + O + H)
    x = torch.randn(B, H, W, C, generator=g).cuda().to(dtype)
    pd = ops.pack_dcn_weight((torch.randn(O, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda(), dtype)
    logits = (torch.randn(B, H, W, 32, generator=g) * 1.5).cuda()
    kw = dict(bias=(torch.randn(O, generator=g) * 0.1).cuda(), stride=(1, 1), padding=(1, 1), dilation=(1, 1), mask_sigmoid=True, relu=True)

    def run():
        return ops.deform_conv_general(x, pd, logits[..., :18], logits[..., 18:27], torch.empty((B, H, W, O), dtype=dtype, device='cuda'), 'nhwc', **kw)

    a = run()
    with _lib.test_switch('VD3D_DCN_NO_LSTAGE'):
        b = run()
    torch.cuda.synchronize()
    assert torch.equal(a, b), 'max diff %.3e' % (a.float() - b.float()).abs().max().item()
