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


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 5e-5), (torch.bfloat16, 2e-2)])
def test_pack_module_nhwc_engine_path_with_bn_relu(dtype, tol):
    C, O = 64, 64
    m = _pack_module(C, O, 2)
    bn = torch.nn.BatchNorm2d(O).eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(O, generator=g) + 0.5); bn.bias.copy_(torch.randn(O, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(O, generator=g) * 0.1); bn.running_var.copy_(torch.rand(O, generator=g) + 0.5)
    x = torch.randn(2, C, 9, 12, generator=g)
    rnd = (lambda t: t.to(torch.bfloat16).float()) if dtype == torch.bfloat16 else None
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
