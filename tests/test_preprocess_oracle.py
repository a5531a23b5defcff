"""CPU: oracle/preprocess_ref.py and the host calibration helper against the reference's own augmentation classes
(tests/golden/preprocess_cases.npz; the cv2.resize inside them is served by the oracle's INTER_LINEAR restatement -- the
interpolation itself is unpinned, everything around it is the reference's code)."""
import numpy as np

from oracle import preprocess_ref
from tests.common import load_golden

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def test_pipeline_matches_reference_classes():
    g = load_golden('preprocess_cases')
    for i in range(3):
        Hs, Ws, crop_top, H, W = [int(v) for v in g['c%d_cfg' % i]]
        for side in ('left', 'right'):
            got = preprocess_ref.preprocess(g['c%d_%s_u8' % (i, side)], crop_top, (H, W), MEAN, STD)
            np.testing.assert_allclose(got, g['c%d_%s' % (i, side)], rtol=0, atol=1e-6)


def test_identity_scale_and_padding_value():
    g = load_golden('preprocess_cases')
    u8 = g['c2_left_u8']                       # scale 1: output = normalised crop
    got = preprocess_ref.preprocess(u8, 4, (32, 128), MEAN, STD)
    want = ((u8[4:].astype(np.float32) / 255.0 - np.float32(MEAN)) / np.float32(STD)).transpose(2, 0, 1)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    pad = preprocess_ref.preprocess(g['c1_left_u8'], 10, (32, 128), MEAN, STD)     # zero pad BEFORE Normalize -> -mean/std
    np.testing.assert_allclose(pad[:, :, -1], np.broadcast_to((-np.float32(MEAN) / np.float32(STD))[:, None], (3, 32)), atol=1e-6)


def test_resize_coordinates_clamp_like_opencv():
    s0, w = preprocess_ref._coords(8, 4)       # 2x upsample: first / last destination pixels clamp to the border sources
    assert s0.tolist() == [0, 0, 0, 1, 1, 2, 2, 3] and w[0] == 0 and w[-1] == 0 and abs(w[1] - 0.25) < 1e-7 and abs(w[2] - 0.75) < 1e-7
    s0, w = preprocess_ref._coords(2, 4)       # 2x downsample: centres between source pairs
    assert s0.tolist() == [0, 2] and np.allclose(w, 0.5)


def test_calibration_update_matches_reference():
    from visualdet3d_amd.hip_ops import adjust_calib, resized_shape
    g = load_golden('preprocess_cases')
    for i in range(3):
        Hs, Ws, crop_top, H, W = [int(v) for v in g['c%d_cfg' % i]]
        Hr, Wr, scale = resized_shape(Hs, Ws, crop_top, (H, W))
        assert Hr == H
        np.testing.assert_allclose(adjust_calib(g['c%d_P2_in' % i], crop_top, scale), g['c%d_P2' % i], rtol=1e-14)
