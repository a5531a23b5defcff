"""CPU: mono detectors' oracle restatement (LookGround, DCN head) against golden outputs of the reference itself."""
import pytest
import torch

from oracle import detector_oracle as orc
from tests.common import assert_detections_close, load_golden, mono_case_from_golden, rel_err, subsample
from visualdet3d_amd.utils import synthetic as syn


# (the last case is config/Yolo3D_example:113-136 AS SHIPPED: ResNet-101, 288 x 1280, nms 0.5, post_optimization on)
@pytest.mark.parametrize('name', ['groundaware_r34_96x320', 'groundaware_r34_384x1280', 'yolo3d_dcn_r34_96x320', 'groundaware_r101_288x1280_postopt'])
def test_mono_oracle_matches_reference_golden(name):
    from visualdet3d_amd.networks.detectors import GroundAwareYolo3D, Yolo3D
    g = load_golden(name)
    cfg, (img, P2), winit = mono_case_from_golden(g, name)
    m = (GroundAwareYolo3D if cfg.name == 'GroundAwareYolo3D' else Yolo3D)(cfg)
    sd = syn.seeded_state_dict(m.state_dict(), **winit)
    with torch.no_grad():
        outs, st = orc.mono3d_forward(sd, cfg, img, P2, return_stages=True)
    for f in range(img.shape[0]):
        assert rel_err(subsample(st['features'][f:f + 1]), g['f%d_features_sub' % f]) < 1e-4
        assert rel_err(subsample(st['cls_preds'][f:f + 1]), g['f%d_cls_sub' % f]) < 1e-4
        assert rel_err(subsample(st['reg_preds'][f:f + 1]), g['f%d_reg_sub' % f]) < 1e-4
        s, b, l, _ = outs[f]
        loose = None
        if cfg.head.test_cfg.post_optimization:
            # detection_3d_head.py:396-398 -> _post_process: the hill climb (fp64, steps >= 0.0125 rad) restated in oracle/post_opt_ref.py
            import numpy as np
            from oracle import post_opt_ref
            b = torch.from_numpy(post_opt_ref.post_process(s.numpy(), b.numpy(), l.numpy(), P2[f].numpy()))
            loose = {10: (0.03, 2 * np.pi)}
        assert_detections_close((s, b, l), (g['f%d_scores' % f], g['f%d_boxes' % f], g['f%d_labels' % f]), rtol=1e-4,
                                what='%s frame %d' % (name, f), loose_fields=loose)
