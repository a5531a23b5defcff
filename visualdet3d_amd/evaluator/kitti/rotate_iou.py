"""``rotate_iou_gpu_eval`` with the reference's signature (``visualDet3D/evaluator/kitti/rotate_iou.py:294-328``): numpy in,
numpy out, the pairwise rotated IoU computed by ``vd3d_rotate_iou_eval`` on the GPU instead of a numba.cuda kernel."""
import numpy as np
import torch

from ... import _lib
from ...hip_ops import _p, _stream, check


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1):
    # the reference returns float32 whatever the caller passed: `boxes` is re-bound to its float32 copy before
    # `iou.astype(boxes.dtype)` (rotate_iou.py:307-328)
    dtype = np.float32
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32)).cuda()
    q = torch.from_numpy(np.ascontiguousarray(query_boxes, dtype=np.float32)).cuda()
    N, K = b.shape[0], q.shape[0]
    iou = torch.zeros((N, K), dtype=torch.float32, device='cuda')
    if N == 0 or K == 0:
        return iou.cpu().numpy().astype(dtype)
    check(_lib.lib().vd3d_rotate_iou_eval(_p(b), _p(q), N, K, int(criterion), _p(iou), _stream()), 'vd3d_rotate_iou_eval')
    return iou.cpu().numpy().astype(dtype)
