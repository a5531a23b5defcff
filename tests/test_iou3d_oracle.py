"""CPU: the iou3d restatement (oracle/iou3d_ref.py) against golden values produced by the reference's own device
functions compiled for the host (oracle/build_ref.sh, oracle/make_golden_native.py)."""
import numpy as np

from oracle import iou3d_ref
from tests.common import load_golden


def test_iou3d_restatement_matches_reference_native_golden():
    g = load_golden('iou3d_cases')
    a, b = g['boxes_a'], g['boxes_b']
    assert np.allclose(iou3d_ref.pairwise(iou3d_ref.box_overlap, a, b), g['overlap'], rtol=1e-4, atol=1e-5)
    assert np.allclose(iou3d_ref.pairwise(iou3d_ref.iou_bev, a, b), g['iou_bev'], rtol=1e-4, atol=1e-6)
    assert np.allclose(iou3d_ref.pairwise(iou3d_ref.iou_normal, a, b), g['iou_normal'], rtol=1e-6, atol=1e-7)


def test_iou3d_nms_restatement_matches_golden():
    g = load_golden('iou3d_cases')
    nb = g['nms_boxes']
    assert np.array_equal(iou3d_ref.nms(nb, 0.3), g['nms_keep_rot_03'])
    assert np.array_equal(iou3d_ref.nms(nb, 0.3, normal=True), g['nms_keep_norm_03'])
    assert np.array_equal(iou3d_ref.nms(nb, 0.1), g['nms_keep_rot_01'])


def test_iou3d_edge_cases():
    z = np.zeros((0, 5), dtype=np.float32)
    assert iou3d_ref.pairwise(iou3d_ref.box_overlap, z, z).shape == (0, 0)
    assert len(iou3d_ref.nms(z, 0.5)) == 0
    sq = np.array([0, 0, 2, 2, 0.7], dtype=np.float32)
    assert abs(float(iou3d_ref.iou_bev(sq, sq)) - 1.0) < 1e-4
    far = np.array([50, 50, 52, 52, 0.1], dtype=np.float32)
    assert float(iou3d_ref.box_overlap(sq, far)) == 0.0
