"""ORACLE (test infrastructure only): restatement of the post-path geometry and result text of
``visualDet3D/networks/pipelines/evaluators.py:112-129`` (test_one), ``networks/utils/utils.py:262-278`` (BackProjection),
``utils/utils.py:47-62`` (alpha2theta_3d) and ``data/kitti/utils.py:162-201`` (write_result_to_file), fp32 torch on the host in the
reference's operation order.  Pinned by tests/golden/postpath_cases.npz: rows and result text produced by the reference's own
BackProjection / BBox3dProjector / write_result_to_file (oracle/make_golden_postpath.py)."""
import numpy as np
import torch


def postpath_rows(bbox, P2, original_P):
    """bbox [N,11] float32 tensor, P2 / original_P numpy [3,4] -> rows [N,12] float32 (x1,y1,x2,y2 in the original image, x3d,
    y_bottom, z, w, h, l, alpha, theta)."""
    bbox = torch.as_tensor(bbox, dtype=torch.float32).clone()
    P2 = np.asarray(P2, dtype=np.float64)
    b2 = bbox[:, 0:4]
    st = bbox[:, 4:]
    fx, fy, cx, cy, tx, ty = (float(P2[0, 0]), float(P2[1, 1]), float(P2[0, 2]), float(P2[1, 2]), float(P2[0, 3]), float(P2[1, 3]))
    z = st[:, 2:3]
    x3d = (st[:, 0:1] * z - cx * z - tx) / fx
    y3d = (st[:, 1:2] * z - cy * z - ty) / fy
    s3 = torch.cat([x3d, y3d, st[:, 2:]], dim=1)
    tp2 = torch.tensor(P2, dtype=torch.float32)
    theta = s3[:, 6] + torch.atan2(s3[:, 0] + tp2[0, 3] / tp2[0, 0], s3[:, 2])
    original_P = np.asarray(original_P, dtype=np.float64)
    scale_x = original_P[0, 0] / P2[0, 0]
    scale_y = original_P[1, 1] / P2[1, 1]
    shift_left = original_P[0, 2] / scale_x - P2[0, 2]
    shift_top = original_P[1, 2] / scale_y - P2[1, 2]
    b2[:, 0:4:2] += float(shift_left)
    b2[:, 1:4:2] += float(shift_top)
    b2[:, 0:4:2] *= float(scale_x)
    b2[:, 1:4:2] *= float(scale_y)
    s3[:, 1] = s3[:, 1] + 0.5 * s3[:, 4]
    return torch.cat([b2, s3[:, :7], theta[:, None]], dim=1).numpy()


def result_text(scores, rows, obj_types, threshold=0.4):
    text = ''
    for i in range(len(rows)):
        if scores[i] < threshold:
            continue
        r = rows[i]
        text += ('{} -1 -1 {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {} \n').format(
            obj_types[i], r[10], r[0], r[1], r[2], r[3], r[8], r[7], r[9], r[4], r[5], r[6], r[11], float(scores[i]))
    return text
