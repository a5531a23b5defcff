"""Yaw post-optimisation (reference: ``visualDet3D/networks/lib/fast_utils/hill_climbing.py``).

The reference climbs one box at a time on the host (numba) after a device->host copy per box; here the whole set of
detections is refined by one launch of ``vd3d_post_opt`` (csrc/post_opt.hip, fp64 per lane).  ``post_opt`` keeps the
reference signature (hill_climbing.py:7) for single boxes; ``post_opt_batch`` is what the heads call."""
import numpy as np
import torch

from .... import hip_ops

# the reference clamps projected corners to a hard-coded 1280 x 288 crop (hill_climbing.py:111-113)
REFERENCE_CLAMP_WH = (1280.0, 288.0)


def post_opt_batch(bboxes, labels, P2s, counts=None, clamp_wh=REFERENCE_CLAMP_WH):
    """bboxes [B,K,11] (or [K,11]) fp32 cuda, labels int, P2s [B,3,4]: refines alpha in place for label-0 boxes deeper
    than 3 m (detection_3d_head.py:303-305) and returns ``bboxes``."""
    squeeze = bboxes.dim() == 2
    b3 = bboxes.unsqueeze(0) if squeeze else bboxes
    l2 = (labels.unsqueeze(0) if squeeze else labels).to(torch.int32).contiguous()
    P2s = P2s.reshape(-1, 3, 4)
    if not b3.is_contiguous():
        tmp = b3.contiguous()
        hip_ops.post_opt_batched(tmp, l2, counts, P2s, clamp_wh)
        b3.copy_(tmp)
    else:
        hip_ops.post_opt_batched(b3, l2, counts, P2s, clamp_wh)
    return bboxes


def post_opt(bbox_2d, bbox3d_state_3d, P2, cx, cy):
    """Reference signature (hill_climbing.py:7-23): one box; ``bbox3d_state_3d`` = [x3d, y3d, z, w, h, l, alpha]
    (x3d / y3d are unused -- the centre is re-derived from (cx, cy, z) as in the reference).  Returns a tensor
    [cx, cy, z, w, h, l, alpha'] like ``bbox3d_state_3d``."""
    dev = bbox3d_state_3d.device
    if dev.type != 'cuda':
        raise NotImplementedError('post_opt runs on the GPU (vd3d_post_opt); got a %s tensor' % dev.type)
    st = bbox3d_state_3d.float()
    box = torch.cat([bbox_2d.float().reshape(4), torch.tensor([cx, cy], dtype=torch.float32, device=dev), st[2:7]])
    box = box.reshape(1, 1, 11).contiguous()
    P2 = torch.as_tensor(np.asarray(P2), dtype=torch.float32, device=dev).reshape(1, 3, 4)
    hip_ops.post_opt_batched(box, torch.zeros((1, 1), dtype=torch.int32, device=dev), None, P2,
                             REFERENCE_CLAMP_WH, min_depth=float('-inf'))
    return box[0, 0, 4:].to(bbox3d_state_3d.dtype)
