"""CPU: KM3D (DLA-34 + DLA-Up + keypoint head + decode) oracle restatement against golden outputs of the reference."""
import numpy as np
import pytest
import torch

from oracle import detector_oracle as orc
from tests.common import assert_detections_close, load_golden, rel_err, subsample
from visualdet3d_amd.utils import synthetic as syn


def km3d_case_from_golden(g):
    _, H, W, frames, wseed, iseed = [int(v) for v in g['meta']]
    depth = int(g['meta'][0])
    if depth == 34:
        cfg = syn.km3d_cfg(score_thr=float(g['score_thr']), output_w=W // 4)
    else:                                                     # ResNet + ConvTranspose core (config/KM3D_example)
        cfg = syn.km3d_resnet_cfg(score_thr=float(g['score_thr']), output_w=W // 4, depth=depth)
    img = syn.mono_image(frames, H, W, seed=iseed)
    P2, _ = syn.kitti_calib(W, batch=frames)
    return cfg, (img, P2), dict(seed=wseed)


def km3d_state_dict(model, g, winit):
    sd = syn.seeded_state_dict(model.state_dict(), **winit)
    if 'head_gain' in g and float(g['head_gain']) != 1.0:
        syn.scale_km3d_head(sd, float(g['head_gain']))
    return sd


@pytest.mark.parametrize('name', ['km3d_dla34_96x320', 'km3d_dla34_192x640'])
def test_km3d_oracle_matches_reference_golden(name):
    from visualdet3d_amd.networks.detectors import KM3D
    g = load_golden(name)
    cfg, (img, P2), winit = km3d_case_from_golden(g)
    m = KM3D(cfg)
    sd = syn.seeded_state_dict(m.state_dict(), **winit)
    assert len(sd) == 424 and sum(v.numel() for v in sd.values()) == 20572615   # the reference's KM3D(DLA-34) checkpoint layout
    with torch.no_grad():
        dets, st = orc.km3d_forward(sd, cfg, img, P2, return_stages=True)
    for f in range(img.shape[0]):
        assert rel_err(subsample(st['features'][f:f + 1]), g['f%d_features_sub' % f]) < 1e-4
        for h in orc.KM3D_HEADS:
            assert rel_err(subsample(st[h][f:f + 1]), g['f%d_%s_sub' % (f, h)]) < 2e-4, h
        s, b, l = dets[f]
        assert l.shape[1:] == (1,)
        assert_detections_close((s, b, l), (g['f%d_scores' % f], g['f%d_boxes' % f], g['f%d_labels' % f]), rtol=1e-3,
                                what='%s frame %d' % (name, f))


def test_km3d_resnet18_core_oracle_matches_reference_golden():
    """config/KM3D_example's core: ResNet-18 + three ConvTranspose2d(4x4, s2) + BN + ReLU (KM3D_core.py:34-47), head 256 -> 64."""
    from visualdet3d_amd.networks.detectors import KM3D
    name = 'km3d_res18_192x640'
    g = load_golden(name)
    cfg, (img, P2), winit = km3d_case_from_golden(g)
    m = KM3D(cfg)
    sd = km3d_state_dict(m, g, winit)
    assert len(sd) == 176 and sum(v.numel() for v in sd.values()) == 16714375   # the reference's KM3D(ResNet-18) checkpoint layout
    with torch.no_grad():
        dets, st = orc.km3d_forward(sd, cfg, img, P2, return_stages=True)
    for f in range(img.shape[0]):
        assert rel_err(subsample(st['features'][f:f + 1]), g['f%d_features_sub' % f]) < 1e-4
        for h in orc.KM3D_HEADS:
            assert rel_err(subsample(st[h][f:f + 1]), g['f%d_%s_sub' % (f, h)]) < 2e-4, h
        assert_detections_close(dets[f], (g['f%d_scores' % f], g['f%d_boxes' % f], g['f%d_labels' % f]), rtol=1e-3,
                                what='%s frame %d' % (name, f))


def test_deconv4x4s2_as_conv3x3_is_the_transposed_convolution():
    """Host logic of the ResNet core: the 3x3 / 4*Cout re-expression + pixel shuffle == F.conv_transpose2d(4x4, s2, p1)."""
    import torch.nn.functional as F
    from visualdet3d_amd.networks.detectors.KM3D_core import deconv4x4s2_as_conv3x3
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 5, 7, generator=gen)
    w = torch.randn(8, 6, 4, 4, generator=gen)
    want = F.conv_transpose2d(x, w, None, stride=2, padding=1)
    y = F.conv2d(x, deconv4x4s2_as_conv3x3(w), None, padding=1)
    B, _, H, W = y.shape
    got = y.view(B, 2, 2, 6, H, W).permute(0, 3, 4, 1, 5, 2).reshape(B, 6, 2 * H, 2 * W)
    assert (got - want).abs().max().item() < 1e-4
