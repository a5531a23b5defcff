"""CPU: oracle/rotate_iou_ref.py against the reference's own numba device functions executed as plain Python
(tests/golden/rotate_iou_cases.npz)."""
import numpy as np

from oracle import rotate_iou_ref
from tests.common import load_golden


def test_matches_reference_device_functions():
    g = load_golden('rotate_iou_cases')
    for crit in (-1, 0, 1, 2):
        want = g['iou_crit%d' % crit]
        got = rotate_iou_ref.rotate_iou_eval(g['boxes'], g['query'], crit)
        ok = ~np.isnan(want)
        # the stub run evaluates cos / sin and a few literals in fp64 where numba's typing (restated here) stays fp32
        np.testing.assert_allclose(got[ok], want[ok], rtol=2e-5, atol=2e-6)
        assert ((got > 0) == (want > 0))[ok].all()
    # exact duplicates: the reference's vertex ordering degenerates and it reports 0 (rotate_iou.py:32-68); reproduced
    dup = rotate_iou_ref.rotate_iou_eval(g['boxes'][:5], g['boxes'][:5], -1)
    assert np.array_equal(np.diag(dup) > 0, np.diag(g['iou_crit-1'][:5, :5]) > 0)


def test_simple_geometry():
    a = np.array([[0, 0, 4, 2, 0.0]], np.float32)
    b = np.array([[1, 0, 4, 2, 0.0], [10, 10, 1, 1, 0.3], [0, 0, 2, 4, np.pi / 2]], np.float32)
    iou = rotate_iou_ref.rotate_iou_eval(a, b, -1)
    assert abs(iou[0, 0] - 6.0 / 10.0) < 1e-5 and iou[0, 1] == 0
    inter = rotate_iou_ref.rotate_iou_eval(a, b, 2)
    assert abs(inter[0, 0] - 6.0) < 1e-5
