"""CPU: host-side edges of the evaluation pipeline that need no GPU -- frames with zero detections, overflow-marked
detection counts in the multi-GPU gather, evaluator dtype conventions."""
import os
import tempfile

import numpy as np
import pytest
import torch


class _Dataset:
    def __getitem__(self, i):
        P = np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.21], [0, 0, 1, 0.0027]])
        return dict(calib=P, original_P=P, image=np.zeros((8, 8, 3), np.float32))

    @staticmethod
    def collate_fn(batch):
        return batch


def test_test_one_writes_an_empty_file_for_a_frame_without_detections():
    """bbox is [0, 11] on such a frame; the reference writes an empty result file and carries on
    (networks/pipelines/evaluators.py:101-129, data/kitti/utils.py:186)."""
    from visualdet3d_amd.networks.pipelines.evaluators import test_one

    def test_func(collated, model, writer, cfg=None):
        return torch.zeros(0), torch.zeros(0, 11), []

    out = tempfile.mkdtemp()
    test_one(None, 7, _Dataset(), None, test_func, result_path=out)
    path = os.path.join(out, '000007.txt')
    assert os.path.exists(path) and open(path).read() == ''


def test_result_text_prints_scores_like_a_zero_dim_tensor():
    """The reference formats ``scores[i]`` (a 0-d tensor) with '{}' -> python-float repr of the fp32 value, not numpy's
    shortest float32 repr (data/kitti/utils.py:196-200)."""
    from visualdet3d_amd.data.kitti.utils import format_result
    s = np.array([0.9], dtype=np.float32)
    text = format_result(s, np.zeros((1, 4), np.float32), np.zeros((1, 7), np.float32), np.zeros(1, np.float32), ['Car'])
    assert text.split()[-1] == '{}'.format(torch.tensor(0.9)) == repr(float(np.float32(0.9)))


def test_gather_helpers_raise_on_overflow_marked_counts():
    """head_postprocess marks an overflowed frame with a negative count; slicing ``pack[b, :-1]`` would silently
    return padding rows as detections."""
    from visualdet3d_amd import distributed as vdist
    pack = torch.zeros(3, 4, 13)
    with pytest.raises(RuntimeError, match='max_candidates'):
        vdist.unpack_detections(pack, torch.tensor([2, -1, 0], dtype=torch.int32))
    g = vdist.DetectionGather(2, 4, 'cpu', world=1)
    g.fill(torch.zeros(2, 4), torch.zeros(2, 4, 11), torch.zeros(2, 4, dtype=torch.int32), torch.tensor([3, -2], dtype=torch.int32))
    g.out[0].copy_(g.pack)
    with pytest.raises(RuntimeError, match='max_candidates'):
        g.detections()
    assert g.tensor_collective is False           # no process group / gloo: the list form; decided once, no try/except per step


def test_padding_rows_are_zeroed_by_selection_not_by_multiplication():
    """the padding of the device result buffers is uninitialised memory (``torch.empty``): NaN / Inf there must not
    survive into the gathered record (NaN * 0 = NaN)."""
    from visualdet3d_amd import distributed as vdist
    B, K, k = 2, 6, 4
    scores = torch.full((B, K), float('nan'))
    boxes = torch.full((B, K, 11), float('inf'))
    labels = torch.full((B, K), 7, dtype=torch.int32)
    count = torch.tensor([2, 0], dtype=torch.int32)
    scores[0, :2] = torch.tensor([0.9, 0.8])
    boxes[0, :2] = 1.5
    pack, c = vdist.pack_detections(scores, boxes, labels, count, k)
    assert bool(torch.isfinite(pack).all()) and bool((pack[0, 2:] == 0).all()) and bool((pack[1] == 0).all())
    assert pack[0, 0, 0] == 0.9 and pack[0, 1, 1] == 1.5 and pack[0, 1, 12] == 7 and c.tolist() == [2, 0]
    g = vdist.DetectionGather(B, k, 'cpu', world=1)
    g.fill(scores, boxes, labels, count)
    assert bool(torch.isfinite(g.pack).all()) and bool((g.pack[0, 2:k] == 0).all()) and g.pack[0, k, 0] == 2
    # a record wider than the detector's own capacity (KM3D decodes K = 100 rows, the bench gathers 128): rows K .. k-1 are zero
    g2 = vdist.DetectionGather(B, 8, 'cpu', world=1)
    g2.fill(scores, boxes, labels, count)
    assert bool(torch.isfinite(g2.pack).all()) and bool((g2.pack[:, K:8] == 0).all())
