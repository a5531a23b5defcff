"""GPU: GroundAwareYolo3D (LookGround head) and Yolo3D (DCNv2 head) end to end on the HIP path: fp32 mode vs golden
outputs of the reference itself; bf16 mode vs the bf16-rounding oracle."""
import pytest
import torch

from oracle import detector_oracle as orc
from tests.common import assert_detections_close, load_golden, mono_case_from_golden, rel_err, subsample
from visualdet3d_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def _model(cfg, winit, dtype):
    from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT
    m = DETECTOR_DICT[cfg.name](cfg)
    sd = syn.seeded_state_dict(m.state_dict(), **winit)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = dtype
    return m, sd


# (the last case is config/Yolo3D_example:113-136 AS SHIPPED: ResNet-101, 288 x 1280, nms 0.5, post_optimization on)
@pytest.mark.parametrize('name', ['groundaware_r34_96x320', 'groundaware_r34_384x1280', 'yolo3d_dcn_r34_96x320', 'groundaware_r101_288x1280_postopt'])
def test_fp32_mode_matches_reference_golden(name):
    g = load_golden(name)
    cfg, (img, P2), winit = mono_case_from_golden(g, name)
    m, _ = _model(cfg, winit, torch.float32)
    outs = m.test_forward_batched(img.cuda(), P2.cuda())
    cls, reg = m._last_raw
    for f in range(img.shape[0]):
        assert rel_err(subsample(cls[f:f + 1].cpu()), g['f%d_cls_sub' % f]) < 1e-3
        assert rel_err(subsample(reg[f:f + 1].cpu()), g['f%d_reg_sub' % f]) < 1e-3
        s, b, l = [t.cpu() for t in outs[f]]
        # post_optimization on (the shipped config): the yaw is the result of a discrete hill climb (steps >= 0.0125 rad, 3.14-vs-pi wrap)
        loose = {10: (0.03, 2 * 3.141592653589793)} if cfg.head.test_cfg.post_optimization else None
        assert_detections_close((s, b, l), (g['f%d_scores' % f], g['f%d_boxes' % f], g['f%d_labels' % f]), rtol=1e-3,
                                what='%s frame %d' % (name, f), loose_fields=loose)
    s1, b1, l1 = m([img[:1].cuda(), P2[:1].cuda()])      # the reference's batch-1 entry point
    assert torch.equal(s1, outs[0][0]) and torch.equal(b1, outs[0][1])


@pytest.mark.parametrize('name', ['groundaware_r34_96x320', 'yolo3d_dcn_r34_96x320'])
def test_bf16_mode_close_to_bf16_oracle(name):
    g = load_golden(name)
    cfg, (img, P2), winit = mono_case_from_golden(g, name)
    m, sd = _model(cfg, winit, torch.bfloat16)
    m.test_forward_batched(img.cuda(), P2.cuda())
    cls, reg = m._last_raw
    with torch.no_grad():
        _, st = orc.mono3d_forward(sd, cfg, img, P2, rnd=orc.bf16_round, return_stages=True)
    assert rel_err(cls.cpu(), st['cls_preds']) < 4e-2
    assert rel_err(reg.cpu(), st['reg_preds']) < 4e-2


@pytest.mark.parametrize('name', ['groundaware_r34_96x320', 'yolo3d_dcn_r34_96x320'])
def test_post_optimization_enabled(name):
    """config/Yolo3D_example:136 turns the hill-climbing yaw refinement on: the detector output must be the oracle's
    _post_process of the un-refined output, and land next to what the reference produced from its own detections."""
    import numpy as np
    from oracle import post_opt_ref
    g, gp = load_golden(name), load_golden('post_opt_cases')
    cfg, (img, P2), winit = mono_case_from_golden(g, name)
    m, _ = _model(cfg, winit, torch.float32)
    raw = m.test_forward_batched(img.cuda(), P2.cuda())
    m.bbox_head.test_cfg.post_optimization = True
    opt = m.test_forward_batched(img.cuda(), P2.cuda())
    for f in range(img.shape[0]):
        s, b, l = [t.cpu().numpy() for t in raw[f]]
        want = post_opt_ref.post_process(s, b, l, P2[f].numpy())
        got = opt[f][1].cpu().numpy()
        assert np.array_equal(got[:, :10], b[:, :10]) and torch.equal(opt[f][0], raw[f][0])
        np.testing.assert_allclose(got[:, 10], want[:, 10], rtol=0, atol=2e-5)
        assert (got[:, 10] != b[:, 10]).any()
        ref = gp['%s_f%d_boxes' % (name, f)]
        if len(ref) == len(got):      # same detections (fp32 mode): yaw within two final hill-climbing steps of the reference's
            d = np.abs(np.sort(got[:, 10]) - np.sort(ref[:, 10]))
            assert (np.minimum(d, 2 * np.pi - d) < 0.03).all()
