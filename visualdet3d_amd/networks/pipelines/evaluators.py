"""What happens to a frame's detections between the detector and the KITTI result file (reference:
``networks/pipelines/evaluators.py:101-146`` test_one).  The geometry runs as one device launch for a whole padded batch
(``vd3d_kitti_postpath``) followed by one device->host copy; the reference does ~15 tensor ops per frame and formats the
text from GPU tensors element by element."""
import numpy as np
import torch

from ... import hip_ops as ops
from ...data.kitti.utils import write_result_to_file


def box_transform(P2, original_P):
    """(shift_left, shift_top, scale_x, scale_y) of evaluators.py:118-122, in float64 like the reference's numpy scalars."""
    P2 = np.asarray(P2, dtype=np.float64)
    original_P = np.asarray(original_P, dtype=np.float64)
    scale_x = original_P[0, 0] / P2[0, 0]
    scale_y = original_P[1, 1] / P2[1, 1]
    return np.array([original_P[0, 2] / scale_x - P2[0, 2], original_P[1, 2] / scale_y - P2[1, 2], scale_x, scale_y])


def postprocess_batch(scores, boxes, counts, P2s, original_Ps):
    """Padded device results of ``forward_device`` / ``get_bboxes_batched`` -> per-frame numpy (scores[k], rows[k,12]).
    rows = x1,y1,x2,y2 (original image), x3d, y_bottom, z, w, h, l, alpha, theta."""
    B = boxes.shape[0]
    P2_host = P2s.detach().cpu().numpy() if torch.is_tensor(P2s) else np.asarray(P2s)
    xf = np.stack([box_transform(P2_host[b], original_Ps[b]) for b in range(B)]).astype(np.float32)
    rows = ops.kitti_postpath(boxes.float().contiguous(), counts, torch.as_tensor(P2_host, dtype=torch.float32, device=boxes.device),
                              torch.from_numpy(xf).to(boxes.device))
    rows_h, scores_h = rows.cpu().numpy(), scores.detach().cpu().numpy()
    ks = counts.cpu().tolist() if counts is not None else [boxes.shape[1]] * B
    return [(scores_h[b, :k], rows_h[b, :k]) for b, k in enumerate(ks)]


def test_one(cfg, index, dataset, model, test_func, backprojector=None, projector=None, result_path='.'):
    """Reference signature (evaluators.py:101); ``backprojector`` / ``projector`` are accepted and unused (their arithmetic is
    inside vd3d_kitti_postpath).  3D detectors only (the 2D-only branch of the reference, :131-146, is not on the path)."""
    data = dataset[index]
    P2 = data['calib'][0] if isinstance(data['calib'], list) else data['calib']
    collated = dataset.collate_fn([data])
    scores, bbox, obj_names = test_func(collated, model, None, cfg=cfg)
    n = len(scores)
    if n == 0:
        # no detection in this frame: the reference still writes an (empty) result file and carries on (evaluators.py:112-129
        # run on empty tensors; data/kitti/utils.py:186 `if len(scores) > 0`)
        write_result_to_file(result_path, index, [], np.zeros((0, 4), np.float32), np.zeros((0, 7), np.float32), np.zeros((0,), np.float32),
                             obj_names, bottom_center_done=True)
        return
    counts = torch.tensor([n], dtype=torch.int32, device=bbox.device)
    (s, rows), = postprocess_batch(scores.reshape(1, n), bbox[:, :11].reshape(1, n, 11).contiguous(), counts,
                                   np.asarray(P2)[None], [data['original_P']])
    write_result_to_file(result_path, index, s, rows[:, 0:4], rows[:, 4:11], rows[:, 11], obj_names, bottom_center_done=True)
