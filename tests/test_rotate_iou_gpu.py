"""GPU: vd3d_rotate_iou_eval against the reference's own device functions (golden) and the oracle on a larger random set."""
import numpy as np
import pytest

from tests.common import load_golden

pytestmark = pytest.mark.gpu


def test_matches_reference_golden():
    from visualdet3d_amd.evaluator.kitti.rotate_iou import rotate_iou_gpu_eval
    g = load_golden('rotate_iou_cases')
    for crit in (-1, 0, 1, 2):
        want = g['iou_crit%d' % crit]
        got = rotate_iou_gpu_eval(g['boxes'], g['query'], crit)
        ok = ~np.isnan(want)
        # exact duplicates are degenerate in the reference (every inside / crossing test is a float tie; it reports 0 or an
        # arbitrary fraction, rotate_iou.py:32-68,160-202): which side of the ties cosf / sinf land on differs per libm
        ok &= ~(g['boxes'][:, None, :] == g['query'][None, :, :]).all(axis=2)
        np.testing.assert_allclose(got[ok], want[ok], rtol=2e-5, atol=2e-6)
        assert ((got > 0) == (want > 0))[ok].all()


def test_random_set_matches_oracle_and_edge_sizes():
    from oracle import rotate_iou_ref
    from visualdet3d_amd.evaluator.kitti.rotate_iou import rotate_iou_gpu_eval
    rng = np.random.default_rng(4)
    N, K = 70, 90
    boxes = np.stack([rng.uniform(-6, 6, N), rng.uniform(0, 12, N), rng.uniform(1, 3, N), rng.uniform(2, 5, N), rng.uniform(-3.2, 3.2, N)], 1).astype(np.float32)
    query = np.stack([rng.uniform(-6, 6, K), rng.uniform(0, 12, K), rng.uniform(1, 3, K), rng.uniform(2, 5, K), rng.uniform(-3.2, 3.2, K)], 1).astype(np.float32)
    for crit in (-1, 2):
        want = rotate_iou_ref.rotate_iou_eval(boxes, query, crit)
        got = rotate_iou_gpu_eval(boxes, query, crit)
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    assert rotate_iou_gpu_eval(boxes[:0], query, -1).shape == (0, K)
    # float32 whatever the caller passes, like the reference (its `boxes` is re-bound to the float32 copy before astype, :307-328)
    assert rotate_iou_gpu_eval(boxes.astype(np.float64), query[:0], -1).dtype == np.float32
    assert rotate_iou_gpu_eval(boxes.astype(np.float64), query.astype(np.float64), -1).dtype == np.float32
