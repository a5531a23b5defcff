"""GPU: stereo cost-volume kernels against the oracle (oracle/detector_oracle.py psm_cosine / cost_volume)."""
import pytest
import torch

from oracle import detector_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('C,H,W,D', [(64, 6, 80, 24), (128, 5, 40, 24), (64, 3, 70, 24), (256, 2, 33, 24), (24, 4, 20, 12)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_psm_cosine(C, H, W, D, dtype):
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(C + W)
    L = torch.randn(2, C, H, W, generator=g)
    R = torch.randn(2, C, H, W, generator=g)
    rnd = orc.bf16_round if dtype == torch.bfloat16 else orc.identity
    c = orc.Ctx({}, rnd)
    ref = orc.psm_cosine(c, rnd(L), rnd(R), D * 4, 4)
    buf = torch.full((2, H, W, D + 8), 3.0, dtype=dtype, device='cuda')
    out = ops.psm_cosine(L.permute(0, 2, 3, 1).contiguous().cuda().to(dtype), R.permute(0, 2, 3, 1).contiguous().cuda().to(dtype), D,
                         out=buf[..., 8:] if D % 8 == 0 else None)
    got = out.float().cpu().permute(0, 3, 1, 2)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert ((got - ref).abs().max() / ref.abs().max()).item() < tol
    # zero where x < d, exactly
    for d in range(1, D):
        assert bool((got[:, d, :, :d] == 0).all())
    if D % 8 == 0:
        assert bool((buf[..., :8] == 3.0).all())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_cost_volume_module(dtype):
    from visualdet3d_amd.networks.lib.PSM_cost_volume import CostVolume
    from visualdet3d_amd.utils import synthetic as syn
    m = CostVolume(downsample_scale=16, max_disp=192, input_features=256, PSM_features=8)
    sd = syn.seeded_state_dict({'x.' + k: v for k, v in m.state_dict().items()}, seed=5)
    sd = {k[2:]: v for k, v in sd.items()}
    m.load_state_dict(sd)
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(9)
    L = torch.randn(2, 256, 6, 20, generator=g)
    R = torch.randn(2, 256, 6, 20, generator=g)
    rnd = orc.bf16_round if dtype == torch.bfloat16 else orc.identity
    c = orc.Ctx({'cv.' + k: v for k, v in sd.items()}, rnd)
    ref = orc.cost_volume(c, 'cv', rnd(L), rnd(R), 192, 16)
    x = torch.cat([L, R], 0).permute(0, 2, 3, 1).contiguous().cuda().to(dtype)
    with torch.no_grad():
        out = m.forward_nhwc(x, 2)
    got = out.float().cpu().permute(0, 3, 1, 2)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert got.shape == ref.shape
    assert ((got - ref).abs().max() / ref.abs().max()).item() < tol


@pytest.mark.parametrize('B,H,W,D', [(2, 7, 45, 12), (1, 24, 80, 12), (3, 2, 40, 5), (1, 5, 83, 24)])
def test_cost_volume_fused_launch_equals_the_three_launch_path(B, H, W, D):
    """vd3d_cost_volume_fused (concat volume + 2 x Conv3d + BN3d + ReLU + reshape in one launch; the volume and the intermediate
    never reach HBM) against costvol_build + 2 x conv3d_3x3x3: same MFMA slices in the same order and the same bf16 rounding
    point for the intermediate, so the results are BIT-IDENTICAL -- ragged tiles (H, W not multiples of 2 x 40), a channel-slice
    output, and the oracle as the external reference."""
    from visualdet3d_amd import hip_ops as ops
    from visualdet3d_amd.networks.lib.PSM_cost_volume import CostVolume
    from visualdet3d_amd.utils import synthetic as syn
    m = CostVolume(downsample_scale=16, max_disp=16 * D, input_features=64, PSM_features=8)
    sd = syn.seeded_state_dict({'x.' + k: v for k, v in m.state_dict().items()}, seed=7)
    sd = {k[2:]: v for k, v in sd.items()}
    m.load_state_dict(sd)
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(B * 100 + W)
    x = torch.randn(2 * B, H, W, 64, generator=g).cuda().to(torch.bfloat16)
    buf = torch.full((B, H, W, 8 * D + 16), 5.0, dtype=torch.bfloat16, device='cuda')
    with torch.no_grad():
        m.fuse_volume = True
        a = m.forward_nhwc(x, B, out=buf[..., 8:8 + 8 * D])
        m.fuse_volume = False
        b = m.forward_nhwc(x, B)
    assert a.shape == b.shape == (B, H, W, 8 * D)
    assert torch.equal(a, b), (a.float() - b.float()).abs().max().item()
    assert bool((buf[..., :8] == 5.0).all()) and bool((buf[..., 8 + 8 * D:] == 5.0).all())
    xs = x.float().cpu().permute(0, 3, 1, 2)
    c = orc.Ctx({'cv.' + k: v for k, v in sd.items()}, orc.bf16_round)
    ref = orc.cost_volume(c, 'cv', xs[:B], xs[B:], 16 * D, 16)
    got = a.float().cpu().permute(0, 3, 1, 2)
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 2e-2


@pytest.mark.parametrize('B,C,H,W,D', [(2, 64, 5, 83, 24), (1, 128, 3, 160, 24), (2, 256, 2, 37, 24), (1, 64, 4, 16, 8), (1, 512, 2, 50, 32)])
def test_psm_cosine_mfma_kernel_vs_valu_kernel_and_oracle(B, C, H, W, D):
    """psm_cosine_mfma_kernel (banded product on v_mfma_f32_16x16x32, fragments straight from global memory; bf16, C % 32 == 0: every
    shape the detectors launch) against the VALU kernel it replaces (VD3D_PSM_VALU=1) and the oracle: same bf16 operands, fp32
    accumulation in a different order -> within ONE bf16 ulp (+ the fp32 summation noise); zeros for x < d exact; ragged rows
    (W not a multiple of 16), D = 8 / 24 / 32, a channel-slice output."""
    from visualdet3d_amd import _lib, hip_ops as ops
    g = torch.Generator().manual_seed(C + W)
    L = torch.randn(B, H, W, C, generator=g).cuda().to(torch.bfloat16)
    R = torch.randn(B, H, W, C, generator=g).cuda().to(torch.bfloat16)
    buf = torch.full((B, H, W, D + 16), 3.0, dtype=torch.bfloat16, device='cuda')
    a = ops.psm_cosine(L, R, D, out=buf[..., 8:8 + D])
    with _lib.test_switch('VD3D_PSM_VALU'):
        b = ops.psm_cosine(L, R, D)
    torch.cuda.synchronize()
    assert bool((buf[..., :8] == 3.0).all()) and bool((buf[..., 8 + D:] == 3.0).all())
    c = orc.Ctx({}, orc.bf16_round)
    ref = orc.psm_cosine(c, L.float().cpu().permute(0, 3, 1, 2), R.float().cpu().permute(0, 3, 1, 2), D * 4, 4).permute(0, 2, 3, 1)
    sc = ref.abs().max().item()
    for got, what in ((a.float().cpu(), 'mfma'), (b.float().cpu(), 'valu')):
        d = (got - ref).abs()
        assert (d / (ref.abs() * 2.0 ** -7 + 3e-5 * sc)).max().item() <= 1.0, what
        for dd in range(1, D):
            assert bool((got[:, :, :min(dd, W), dd] == 0).all()), what
