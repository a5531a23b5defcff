"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of ``torchvision.ops.nms`` -- the third-party op the reference calls at
``visualDet3D/networks/heads/detection_3d_head.py:386`` and ``heads/km3d_head.py:303``.
torchvision is an unpinned dependency (``requirement.txt:3``) and is NOT installed in this image, so
its published semantics are restated here:

  * boxes are ``(x1, y1, x2, y2)``; ``area = (x2 - x1) * (y2 - y1)``;
  * candidates are visited in order of decreasing score (stable: equal scores keep index order);
  * a visited, not-yet-suppressed box ``i`` suppresses every later box ``j`` with
    ``inter / (area_i + area_j - inter) > iou_threshold`` where
    ``inter = max(0, min(x2) - max(x1)) * max(0, min(y2) - max(y1))``;
  * the result is the int64 indices of the kept boxes, in decreasing-score order.

All arithmetic is IEEE fp32 in exactly this operation order; the HIP kernel reproduces the same order so
the keep-indices are bit-exact.  parity unpinned: the reference holds no test vector for this op.
"""
import numpy as np


def nms_numpy(boxes: np.ndarray, scores: np.ndarray, iou_threshold: float) -> np.ndarray:
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = (boxes[:, i] for i in range(4))
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    thr = np.float32(iou_threshold)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), (xx2 - xx1).astype(np.float32))
        h = np.maximum(np.float32(0), (yy2 - yy1).astype(np.float32))
        inter = (w * h).astype(np.float32)
        iou = inter / ((areas[i] + areas[rest]).astype(np.float32) - inter).astype(np.float32)
        suppressed[rest[iou > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def nms_torch(boxes, scores, iou_threshold):
    """torch-tensor front end with the torchvision signature (used by the reference shim)."""
    import torch

    keep = nms_numpy(boxes.detach().cpu().float().numpy(), scores.detach().cpu().float().numpy(),
                     float(iou_threshold))
    return torch.from_numpy(keep).to(boxes.device)
