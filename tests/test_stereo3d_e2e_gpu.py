"""GPU: end-to-end Stereo3D (YOLOStereo3D) on the HIP path.

  * fp32 validation mode vs golden outputs produced by the REFERENCE ITSELF (tests/golden, oracle/make_golden.py):
    the north-star bar, 1e-3 relative on scores/boxes, identical detections (matching tolerant to near-tied scores);
  * bf16 mode vs the oracle with identical bf16 rounding points, looser tolerance, stage taps;
  * batched == per-sample;  state_dict key parity with the reference's checkpoint layout."""
import numpy as np
import pytest
import torch

from oracle import detector_oracle as orc
from tests.common import assert_detections_close, load_golden, rel_err, stereo_case_from_golden, subsample
from visualdet3d_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def _model(cfg, winit, dtype):
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3D
    m = Stereo3D(cfg)
    sd = syn.seeded_state_dict(m.state_dict(), **winit)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = dtype
    return m, sd


@pytest.mark.parametrize('name', ['stereo3d_r34_96x320', 'stereo3d_r34_384x1280', 'stereo3d_r34_384x1280_thr06', 'stereo3d_r50_96x320'])
def test_fp32_mode_matches_reference_golden(name):
    g = load_golden(name)
    cfg, (L, R, P2, P3), winit = stereo_case_from_golden(g)
    m, _ = _model(cfg, winit, torch.float32)
    outs = m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    cls, reg = m._last_raw
    for f in range(L.shape[0]):
        assert rel_err(subsample(cls[f:f + 1].cpu()), g['f%d_cls_sub' % f]) < 1e-3
        assert rel_err(subsample(reg[f:f + 1].cpu()), g['f%d_reg_sub' % f]) < 1e-3
        s, b, l = [t.cpu() for t in outs[f]]
        assert_detections_close((s, b, l), (g['f%d_scores' % f], g['f%d_boxes' % f], g['f%d_labels' % f]),
                                rtol=1e-3, what='%s frame %d' % (name, f))
    # the reference's own batch-1 entry point gives the same result as the batched one
    s1, b1, l1 = m([L[:1].cuda(), R[:1].cuda(), P2[:1].cuda(), P3[:1].cuda()])
    assert torch.equal(s1, outs[0][0]) and torch.equal(b1, outs[0][1]) and torch.equal(l1, outs[0][2])
    assert l1.dtype == torch.int64 and b1.shape[1] == 11


def test_bf16_mode_matches_bf16_oracle():
    g = load_golden('stereo3d_r34_96x320')
    cfg, (L, R, P2, P3), winit = stereo_case_from_golden(g)
    m, sd = _model(cfg, winit, torch.bfloat16)
    outs = m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    cls, reg = m._last_raw
    with torch.no_grad():
        ref_outs, st = orc.stereo3d_forward(sd, cfg, L, R, P2, rnd=orc.bf16_round, return_stages=True)
    # logits: same rounding points, different fp32 summation order -> occasional 1-ulp bf16 flips upstream
    assert rel_err(cls.cpu(), st['cls_preds']) < 3e-2
    assert rel_err(reg.cpu(), st['reg_preds']) < 3e-2
    # and the bf16 path stays close to the fp32 reference in absolute terms
    for f in range(L.shape[0]):
        assert rel_err(subsample(cls[f:f + 1].cpu()), g['f%d_cls_sub' % f]) < 5e-2


def test_bf16_full_size_runs_and_is_deterministic():
    g = load_golden('stereo3d_r34_384x1280')
    cfg, (L, R, P2, P3), winit = stereo_case_from_golden(g)
    m, _ = _model(cfg, winit, torch.bfloat16)
    a = m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    b = m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    for x, y in zip(a, b):
        assert all(torch.equal(u, v) for u, v in zip(x, y))
    cls, _ = m._last_raw
    assert rel_err(subsample(cls[0:1].cpu()), g['f0_cls_sub']) < 5e-2


def test_state_dict_keys_match_reference_layout():
    g = load_golden('stereo3d_r34_96x320')
    cfg, _, winit = stereo_case_from_golden(g)
    m, sd = _model(cfg, winit, torch.float32)
    keys = list(m.state_dict().keys())
    assert len(keys) == 313   # Stereo3D-R34 checkpoint entries
    for k in ('core.backbone.layer1.0.conv1.weight', 'core.neck.cost_volume_2.down_sample.0.weight',
              'core.neck.depth_reasoning.four_to_eight.0.primary_conv.1.weight',
              'bbox_head.reg_feature_extraction.0.sequence.0.weight', 'bbox_head.balance_weights',
              'bbox_head.loss_cls.balance_weights', 'bbox_head.regression_weight',
              'core.neck.depth_reasoning.depth_output.8.bias'):
        assert k in keys, k


def test_config3_stereo_core_with_dcn_head_r50():
    """BASELINE config 3: YOLOStereo3D ResNet-50 core + the base (DCNv2) head, expressed exactly as SURVEY.md 0.8 says:
    override ``Stereo3D.build_head``.  Logits vs the oracle (stereo_core + dcn_head), fp32 mode."""
    import tempfile
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3D
    from visualdet3d_amd.networks.heads.detection_3d_head import AnchorBasedDetection3DHead

    class Stereo3DDCN(Stereo3D):
        def build_head(self, network_cfg):
            self.bbox_head = AnchorBasedDetection3DHead(**(network_cfg.head))

    tmp = tempfile.mkdtemp()
    cfg = syn.stereo3d_cfg(tmp, depth=50, score_thr=0.5)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    m = Stereo3DDCN(cfg)
    sd = syn.seeded_state_dict(m.state_dict(), seed=6, head_std=0.006)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = torch.float32
    L, R = syn.stereo_pair(2, 96, 320, seed=9)
    P2, P3 = syn.kitti_calib(320, batch=2)
    outs = m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    cls, reg = m._last_raw
    with torch.no_grad():
        c = orc.Ctx(sd)
        feats, _ = orc.stereo_core(c, L, R, 50)
        want_cls, want_reg = orc.dcn_head(c, feats, 3)
    assert rel_err(cls.cpu(), want_cls) < 1e-3
    assert rel_err(reg.cpu(), want_reg) < 1e-3
    assert len(outs) == 2
