"""GPU: vd3d_preprocess_image (uint8 frame -> network input, one launch) against the reference's augmentation classes
(golden) and the oracle at the KITTI frame size; the packed NHWC4 output feeds the fused stem unchanged."""
import numpy as np
import pytest
import torch

from tests.common import load_golden

pytestmark = pytest.mark.gpu
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def test_matches_reference_pipeline_golden():
    from visualdet3d_amd import hip_ops as ops
    g = load_golden('preprocess_cases')
    for i in range(3):
        Hs, Ws, crop_top, H, W = [int(v) for v in g['c%d_cfg' % i]]
        frames = [torch.from_numpy(g['c%d_%s_u8' % (i, s)]).cuda() for s in ('left', 'right')]
        got = ops.preprocess_images(frames, crop_top, (H, W), MEAN, STD).cpu().numpy()
        np.testing.assert_allclose(got[0], g['c%d_left' % i], rtol=0, atol=2e-6)
        np.testing.assert_allclose(got[1], g['c%d_right' % i], rtol=0, atol=2e-6)


def test_kitti_size_against_oracle_and_packed_layout():
    from oracle import preprocess_ref
    from visualdet3d_amd import hip_ops as ops
    rng = np.random.default_rng(21)
    u8 = rng.integers(0, 256, (375, 1242, 3), dtype=np.uint8)
    want = preprocess_ref.preprocess(u8, 100, (288, 1280), MEAN, STD)
    f = torch.from_numpy(u8).cuda()
    got = ops.preprocess_images([f], 100, (288, 1280), MEAN, STD)
    np.testing.assert_allclose(got[0].cpu().numpy(), want, rtol=0, atol=2e-6)
    packed = ops.preprocess_images([f], 100, (288, 1280), MEAN, STD, packed=True)
    ref_pack, _, _, _ = ops._pack_stem_images(got, torch.bfloat16)       # the path the detectors use from fp32 NCHW
    assert torch.equal(packed, ref_pack)
