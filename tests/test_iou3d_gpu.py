"""GPU: iou3d HIP kernels vs (a) golden values from the reference's own device functions, (b) the oracle."""
import numpy as np
import pytest
import torch

from oracle import iou3d_ref
from tests.common import load_golden

pytestmark = pytest.mark.gpu


def test_pairwise_matches_reference_golden():
    from visualdet3d_amd.networks.lib.ops.iou3d import boxes_iou_bev, iou3d_hip
    g = load_golden('iou3d_cases')
    a, b = torch.from_numpy(g['boxes_a']).cuda(), torch.from_numpy(g['boxes_b']).cuda()
    ov = torch.zeros(a.shape[0], b.shape[0], device='cuda')
    iou3d_hip.boxes_overlap_bev_gpu(a, b, ov)
    assert np.allclose(ov.cpu().numpy(), g['overlap'], rtol=1e-4, atol=1e-5)
    assert np.allclose(boxes_iou_bev(a, b).cpu().numpy(), g['iou_bev'], rtol=1e-4, atol=1e-6)


def test_nms_matches_reference_golden():
    from visualdet3d_amd.networks.lib.ops.iou3d import iou3d_hip, nms_gpu, nms_normal_gpu
    g = load_golden('iou3d_cases')
    nb = torch.from_numpy(g['nms_boxes']).cuda()
    for thr, normal, key in ((0.3, False, 'nms_keep_rot_03'), (0.3, True, 'nms_keep_norm_03'), (0.1, False, 'nms_keep_rot_01')):
        keep = torch.zeros(nb.shape[0], dtype=torch.long)   # CPU keep, like the reference extension
        k = (iou3d_hip.nms_normal_gpu if normal else iou3d_hip.nms_gpu)(nb, keep, thr)
        assert np.array_equal(keep[:k].numpy(), g[key]), key
    # python-level wrappers: sort by score first
    scores = torch.linspace(1, 0, nb.shape[0]).cuda()   # already sorted -> same keep
    assert np.array_equal(nms_gpu(nb, scores, 0.3).cpu().numpy(), g['nms_keep_rot_03'])
    perm = torch.randperm(nb.shape[0], generator=torch.Generator().manual_seed(0)).cuda()
    kept = nms_normal_gpu(nb[perm], scores[perm], 0.3)
    assert np.array_equal(perm[kept].cpu().numpy(), g['nms_keep_norm_03'])


def test_boxes_iou3d_and_large_nms():
    from visualdet3d_amd.networks.lib.ops.iou3d import boxes_iou3d_gpu, iou3d_hip
    rng = np.random.default_rng(5)
    def b7(n):
        return np.concatenate([rng.uniform(-10, 10, (n, 1)), rng.uniform(0, 2, (n, 1)), rng.uniform(5, 40, (n, 1)),
                               rng.uniform(1.3, 1.8, (n, 1)), rng.uniform(1.4, 1.9, (n, 1)), rng.uniform(3, 5, (n, 1)),
                               rng.uniform(-3.14, 3.14, (n, 1))], axis=1).astype(np.float32)
    a, b = b7(20), b7(13)
    b[:5] = a[:5] + 0.1
    got = boxes_iou3d_gpu(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    assert np.allclose(got, iou3d_ref.boxes_iou3d(a, b), rtol=1e-4, atol=1e-6)
    assert got.max() > 0.3
    # multi-block NMS (n > 64 words of mask per row irrelevant; n = 700 -> 11 column blocks) vs the oracle scan with normal IoU
    n = 700
    c = rng.uniform(0, 60, (n, 2)); wh = rng.uniform(2, 8, (n, 2))
    nb = np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(-3, 3, (n, 1))], 1).astype(np.float32)
    keep = torch.zeros(n, dtype=torch.long, device='cuda')
    k = iou3d_hip.nms_normal_gpu(torch.from_numpy(nb).cuda(), keep, 0.2)
    assert np.array_equal(keep[:k].cpu().numpy(), iou3d_ref.nms(nb, 0.2, normal=True))
    # empty input
    e = torch.zeros(0, 5, device='cuda')
    assert iou3d_hip.nms_gpu(e, torch.zeros(0, dtype=torch.long), 0.5) == 0
