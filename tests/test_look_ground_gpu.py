"""GPU: the LookGround sampling kernel (`vd3d_look_ground_sample`, csrc/elementwise.hip) and the whole LookGround block directly against the
oracle (oracle/detector_oracle.py `look_ground`, restating lib/look_ground.py:24-71 with the reference's own `F.grid_sample`) -- until
round 5 the kernel was covered only through the mono end-to-end goldens.

  * the sampled tensor [x ; prior disparity] for seeded features, learned offsets spanning the whole tanh range, several calibrations
    (rows above / below the horizon cy, a camera whose cy lies outside the map), fp32 and both 16-bit formats;
  * the block (disparity conv -> sampling -> 1 x 1 extract -> alpha-scaled residual + ReLU) with alpha != 0 (the reference initialises it
    to 0, i.e. to a no-op: a test at 0 would prove nothing)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import detector_oracle as orc

pytestmark = pytest.mark.gpu


def _oracle_sample(x, disp_raw, P2, baseline=0.54, elevation=1.65):
    """lib/look_ground.py:31-69 from the raw disparity-conv output on: -> [B, 1 + C, H, W] (prior disparity first, like the reference's cat)"""
    P2 = P2.clone().float()
    P2[:, 0:2] /= 16.0
    disp = torch.tanh(disp_raw)
    disp = 0.1 * (0.05 * disp + 0.95 * disp)
    B, _, H, W = x.shape
    yy = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(1, H, W)
    fy, cy, Ty = P2[:, 1:2, 1:2], P2[:, 1:2, 2:3], P2[:, 1:2, 3:4]
    disparity = F.relu(fy * baseline * (yy - cy) / (torch.abs(fy * elevation + Ty) + 1e-10))
    x_base = torch.linspace(-1, 1, W).repeat(B, H, 1)
    y_base = torch.linspace(-1, 1, H).repeat(B, W, 1).transpose(1, 2)
    y_shifts = F.relu(1.535 * (yy - cy) / (2 * (elevation - 0.5 * 1.535))) / (H * 0.5) + disp[:, 0]
    flow = torch.stack((x_base, y_base + y_shifts), dim=3)
    return F.grid_sample(torch.cat([disparity.unsqueeze(1), x], dim=1), flow, mode='bilinear', padding_mode='border', align_corners=True)


def _calibs(B, W):
    from visualdet3d_amd.utils import synthetic as syn
    P2, _ = syn.kitti_calib(W * 16, batch=B)
    P2 = P2.clone()
    if B > 1:
        P2[1, 1, 2] *= 0.5           # horizon in the upper quarter: most rows carry a prior disparity
    if B > 2:
        P2[2, 1, 2] = -40.0          # cy above the image: every row below the horizon
        P2[2, 1, 3] = 30.0
    if B > 3:
        P2[3, 1, 1] *= 1.3
        P2[3, 1, 2] = 16.0 * 200     # cy far below the map: no prior anywhere, y shifts all zero
    return P2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(4, 24, 80, 64), (2, 18, 80, 1024), (1, 7, 13, 8)])
def test_sampling_kernel_matches_the_reference_grid_sample(dtype, shape):
    from visualdet3d_amd import _lib, hip_ops as ops
    B, H, W, C = shape
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn((B, C, H, W), generator=g)
    rnd = {torch.float32: orc.identity, torch.bfloat16: orc.bf16_round, torch.float16: orc.fp16_round}[dtype]
    x = rnd(x)
    disp_raw = 2.5 * torch.randn((B, 1, H, W), generator=g)           # tanh from -1 to 1: shifts of up to 0.1 of the map height either way
    P2 = _calibs(B, W)
    want = _oracle_sample(x, disp_raw, P2)                             # [B, 1 + C, H, W]
    ve = 8 if dtype != torch.float32 else 4
    cpad = (C + 1 + ve - 1) // ve * ve
    xg = x.permute(0, 2, 3, 1).contiguous().cuda().to(dtype)
    dg = disp_raw.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.full((B, H, W, cpad), 7.0, dtype=dtype, device='cuda')
    _lib.check(_lib.lib().vd3d_look_ground_sample(ops._p(xg), ops._p(dg), ops._p(P2.cuda().contiguous()), ops._p(out), B, H, W, C, xg.stride(2), out.stride(2),
                                                  0.54, 1.65, ops.dtype_code(dtype), ops._stream()), 'vd3d_look_ground_sample')
    got = out.float().cpu()
    feat = got[..., :C].permute(0, 3, 1, 2)
    prior = got[..., C]
    # the kernel holds the prior at channel C (the reference at channel 0), padding channels are written as zeros
    assert float(got[..., C + 1:].abs().max()) == 0.0 if cpad > C + 1 else True
    ulp = {torch.float32: 2.0 ** -22, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
    # a bilinear blend of two rows: <= 1 ulp of the format (the output rounding) + the fp32 evaluation of the sampling position (1e-5 of a row:
    # the weights move by that much, the blend by that much of the difference between the two rows)
    tol_f = want[:, 1:].abs() * ulp + 2e-5 * x.abs().max()
    assert bool(((feat - want[:, 1:]).abs() <= tol_f).all()), float(((feat - want[:, 1:]).abs() - tol_f).max())
    tol_p = want[:, 0].abs() * ulp + 2e-5 * max(float(want[:, 0].abs().max()), 1e-3)
    assert bool(((prior - want[:, 0]).abs() <= tol_p).all()), float(((prior - want[:, 0]).abs() - tol_p).max())
    assert float(want[:, 0].max()) > 0 and (B < 4 or float(want[3, 0].abs().max()) == 0.0)      # the cases really differ


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_look_ground_block_matches_oracle_with_nonzero_alpha(dtype):
    from visualdet3d_amd.networks.lib.look_ground import LookGround
    B, C, H, W = 3, 128, 24, 80
    g = torch.Generator().manual_seed(11)
    mod = LookGround(C)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.02))
        mod.alpha.fill_(0.7)
    sd = {'lg.' + k: v.detach().clone() for k, v in mod.state_dict().items()}
    rnd = orc.identity if dtype == torch.float32 else orc.bf16_round
    x = rnd(F.relu(torch.randn((B, C, H, W), generator=g)))
    P2 = _calibs(B, W)
    with torch.no_grad():
        want = orc.look_ground(orc.Ctx(sd, rnd), 'lg', x, P2)
        got = mod.cuda().forward_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda().to(dtype), P2.cuda()).float().cpu().permute(0, 3, 1, 2)
    sc = float(want.abs().max())
    err = float((got - want).abs().max()) / sc
    # fp32: summation order of the two convs; bf16: the sampled tensor and the block output are each rounded once (2 ulp + the order)
    assert err < (2e-5 if dtype == torch.float32 else 2.0 ** -6), err
    assert float((want - x).abs().max()) > 0.05 * sc, 'alpha != 0 must make the block do something'
