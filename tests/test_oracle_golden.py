"""CPU: the oracle restatement (oracle/detector_oracle.py) against golden outputs produced by the reference
itself (oracle/make_golden.py).  This is what pins the oracle (the reference ships no tests of its own)."""
import numpy as np
import pytest
import torch

from oracle import detector_oracle as orc
from tests.common import assert_detections_close, load_golden, rel_err, stereo_case_from_golden, subsample
from visualdet3d_amd.utils import synthetic as syn


def _state_dict_for(cfg, winit):
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3D
    m = Stereo3D(cfg)
    return syn.seeded_state_dict(m.state_dict(), **winit)


# (stereo3d_r34_288x1280: config/Stereo3D_example:114-122 at its shipped crop size)
@pytest.mark.parametrize('name', ['stereo3d_r34_96x320', 'stereo3d_r34_384x1280', 'stereo3d_r34_384x1280_thr06', 'stereo3d_r50_96x320', 'stereo3d_r34_288x1280'])
def test_oracle_matches_reference_golden(name):
    g = load_golden(name)
    cfg, (L, R, P2, P3), winit = stereo_case_from_golden(g)
    sd = _state_dict_for(cfg, winit)
    with torch.no_grad():
        outs, st = orc.stereo3d_forward(sd, cfg, L, R, P2, return_stages=True)
    for f in range(L.shape[0]):
        assert rel_err(subsample(st['features'][f:f + 1]), g['f%d_features_sub' % f]) < 1e-4
        assert rel_err(subsample(st['cls_preds'][f:f + 1]), g['f%d_cls_sub' % f]) < 1e-4
        assert rel_err(subsample(st['reg_preds'][f:f + 1]), g['f%d_reg_sub' % f]) < 1e-4
        assert int(st['mask'][f].sum()) == int(g['f%d_mask_sum' % f])
        s, b, l, _ = outs[f]
        assert_detections_close((s, b, l), (g['f%d_scores' % f], g['f%d_boxes' % f], g['f%d_labels' % f]),
                                rtol=1e-4, what='%s frame %d' % (name, f))
