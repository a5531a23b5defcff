"""iou3d operator surface of the reference (lib/ops/iou3d/iou3d.py) on the HIP kernels of csrc/iou3d.hip.

Two layers, like the reference:
  * ``iou3d_hip``: the extension-level functions with the pybind signatures of ``iou3d_cuda``
    (src/iou3d.cpp:174-179): ``boxes_overlap_bev_gpu``, ``boxes_iou_bev_gpu``, ``nms_gpu``, ``nms_normal_gpu``
    (``keep`` may be a CPU or GPU LongTensor; the scan itself runs on the device);
  * the Python helpers ``boxes3d_to_bev_torch``, ``boxes_iou_bev``, ``boxes_iou3d_gpu``, ``nms_gpu``, ``nms_normal_gpu``.
    In the reference the last two recurse into themselves (the ``def`` shadows the imported extension symbol,
    iou3d.py:5,72-103); here they do what they were written to do."""
import ctypes as C

import torch

from ..... import _lib
from .....hip_ops import _p, _require_cuda, _stream


class iou3d_hip:
    """Namespace mirroring the ``iou3d_cuda`` extension module."""

    @staticmethod
    def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
        _require_cuda(boxes_a, boxes_b, ans_overlap)
        assert boxes_a.is_contiguous() and boxes_b.is_contiguous() and ans_overlap.is_contiguous()
        assert boxes_a.dtype == boxes_b.dtype == ans_overlap.dtype == torch.float32
        _lib.check(_lib.lib().vd3d_boxes_overlap_bev(_p(boxes_a), boxes_a.shape[0], _p(boxes_b), boxes_b.shape[0],
                                                     _p(ans_overlap), _stream()), 'vd3d_boxes_overlap_bev')
        return 1

    @staticmethod
    def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
        _require_cuda(boxes_a, boxes_b, ans_iou)
        assert boxes_a.is_contiguous() and boxes_b.is_contiguous() and ans_iou.is_contiguous()
        assert boxes_a.dtype == boxes_b.dtype == ans_iou.dtype == torch.float32
        _lib.check(_lib.lib().vd3d_boxes_iou_bev(_p(boxes_a), boxes_a.shape[0], _p(boxes_b), boxes_b.shape[0],
                                                 _p(ans_iou), _stream()), 'vd3d_boxes_iou_bev')
        return 1

    @staticmethod
    def _nms(boxes, keep, thresh, normal):
        _require_cuda(boxes)
        assert boxes.is_contiguous() and boxes.dtype == torch.float32 and boxes.shape[1] == 5
        n = boxes.shape[0]
        ws = torch.empty(_lib.lib().vd3d_nms_bev_workspace_bytes(n), dtype=torch.uint8, device=boxes.device)
        keep_dev = torch.empty(max(n, 1), dtype=torch.int32, device=boxes.device)
        count = torch.zeros(1, dtype=torch.int32, device=boxes.device)
        _lib.check(_lib.lib().vd3d_nms_bev(_p(boxes), n, float(thresh), int(normal), _p(keep_dev), _p(count), _p(ws), _stream()),
                   'vd3d_nms_bev')
        k = int(count.item())
        keep[:k] = keep_dev[:k].to(device=keep.device, dtype=keep.dtype)
        return k

    @staticmethod
    def nms_gpu(boxes, keep, nms_overlap_thresh):
        return iou3d_hip._nms(boxes, keep, nms_overlap_thresh, False)

    @staticmethod
    def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
        return iou3d_hip._nms(boxes, keep, nms_overlap_thresh, True)


def boxes3d_to_bev_torch(boxes3d):
    """(N,7) [x, y, z, h, w, l, ry] -> (N,5) [x1, y1, x2, y2, ry]   (iou3d.py:8-21)"""
    bev = boxes3d.new_empty((boxes3d.shape[0], 5))
    cu, cv = boxes3d[:, 0], boxes3d[:, 2]
    half_l, half_w = boxes3d[:, 5] / 2, boxes3d[:, 4] / 2
    bev[:, 0], bev[:, 1] = cu - half_l, cv - half_w
    bev[:, 2], bev[:, 3] = cu + half_l, cv + half_w
    bev[:, 4] = boxes3d[:, 6]
    return bev


def boxes_iou_bev(boxes_a, boxes_b):
    ans = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_hip.boxes_iou_bev_gpu(boxes_a.float().contiguous(), boxes_b.float().contiguous(), ans)
    return ans


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7) x (M,7) [x, y, z, h, w, l, ry] -> 3D IoU (N,M)   (iou3d.py:37-69)"""
    a_bev, b_bev = boxes3d_to_bev_torch(boxes_a.float()), boxes3d_to_bev_torch(boxes_b.float())
    ov = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_hip.boxes_overlap_bev_gpu(a_bev.contiguous(), b_bev.contiguous(), ov)
    a_hmin, a_hmax = (boxes_a[:, 1] - boxes_a[:, 3]).view(-1, 1), boxes_a[:, 1].view(-1, 1)
    b_hmin, b_hmax = (boxes_b[:, 1] - boxes_b[:, 3]).view(1, -1), boxes_b[:, 1].view(1, -1)
    oh = torch.clamp(torch.min(a_hmax, b_hmax) - torch.max(a_hmin, b_hmin), min=0)
    o3 = ov * oh
    va = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vb = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return o3 / torch.clamp(va + vb - o3, min=1e-7)


def _nms_sorted(boxes, scores, thresh, normal):
    order = scores.sort(0, descending=True)[1]
    boxes = boxes[order].float().contiguous()
    keep = torch.empty(boxes.size(0), dtype=torch.long, device=boxes.device)
    num_out = iou3d_hip._nms(boxes, keep, thresh, normal)
    return order[keep[:num_out]].contiguous()


def nms_gpu(boxes, scores, thresh):
    """boxes (N,5) [x1,y1,x2,y2,ry], scores (N) -> kept indices (rotated IoU)."""
    return _nms_sorted(boxes, scores, thresh, False)


def nms_normal_gpu(boxes, scores, thresh):
    return _nms_sorted(boxes, scores, thresh, True)
