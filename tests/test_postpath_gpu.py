"""GPU: vd3d_kitti_postpath (padded batch, one launch) against the reference's own rows (golden) and the oracle."""
import numpy as np
import pytest
import torch

from tests.common import load_golden

pytestmark = pytest.mark.gpu


def test_postprocess_batch_matches_reference_rows():
    from visualdet3d_amd.networks.pipelines.evaluators import postprocess_batch
    from visualdet3d_amd.data.kitti.utils import format_result
    g = load_golden('postpath_cases')
    cap = 64
    boxes = torch.full((4, cap, 11), 3.0)
    scores = torch.zeros(4, cap)
    counts = torch.zeros(4, dtype=torch.int32)
    for c in range(4):
        b = torch.from_numpy(g['c%d_bbox' % c])
        boxes[c, :len(b)] = b
        scores[c, :len(b)] = torch.from_numpy(g['c%d_scores' % c])
        counts[c] = len(b)
    P2s = np.stack([g['c%d_P2' % c] for c in range(4)])
    outs = postprocess_batch(scores.cuda(), boxes.cuda(), counts.cuda(), P2s, [g['c%d_origP' % c] for c in range(4)])
    names = ['Car', 'Pedestrian', 'Cyclist']
    for c, (s, rows) in enumerate(outs):
        want = g['c%d_rows' % c].reshape(-1, 12)
        assert rows.shape == want.shape
        # fp32, same operation order; atan2f (device libm) vs torch CPU may differ in the last ulp
        np.testing.assert_allclose(rows, want, rtol=2e-6, atol=2e-6)
        text = format_result(s, rows[:, 0:4], rows[:, 4:11], rows[:, 11], [names[i] for i in g['c%d_labels' % c]], bottom_center_done=True)
        ref = bytes(g['c%d_text' % c]).decode()
        assert len(text.splitlines()) == len(ref.splitlines())
        for a, b in zip(text.split(), ref.split()):
            assert a == b or abs(float(a) - float(b)) < 2e-5, (a, b)


def test_padding_rows_zero_and_empty_batch():
    from visualdet3d_amd import hip_ops
    boxes = torch.rand(2, 8, 11).cuda() * 40 + 2
    P2 = torch.tensor([[[700.0, 0, 600, 45], [0, 700.0, 180, 0.2], [0, 0, 1, 0.003]]]).repeat(2, 1, 1).cuda()
    xf = torch.tensor([[1.0, 2.0, 1.1, 0.9]]).repeat(2, 1).cuda()
    out = hip_ops.kitti_postpath(boxes, torch.tensor([3, 0], dtype=torch.int32).cuda(), P2, xf).cpu()
    assert bool((out[0, 3:] == 0).all()) and bool((out[1] == 0).all()) and bool((out[0, :3, 6] == boxes[0, :3, 6].cpu()).all())
    hip_ops.kitti_postpath(torch.empty(0, 0, 11).cuda(), None, torch.empty(0, 3, 4).cuda(), torch.empty(0, 4).cuda())
