"""KITTI result writer (reference: ``visualDet3D/data/kitti/utils.py:162-201``) -- host I/O at the end of the path."""
import os

import numpy as np


def format_result(scores, bbox_2d, bbox_3d_state_3d=None, thetas=None, obj_types=('Car', 'Pedestrian', 'Cyclist'), threshold=0.4,
                  bottom_center_done=False):
    """The text of one frame's result file.  Same columns / precision as the reference writer; ``bbox_3d_state_3d`` rows are
    [x, y_center, z, w, h, l, alpha] (the writer moves y to the bottom centre like data/kitti/utils.py:180-182) unless
    ``bottom_center_done`` (rows produced by ``vd3d_kitti_postpath`` already carry the bottom centre)."""
    n = len(bbox_2d)
    if bbox_3d_state_3d is None:
        st = np.ones([n, 7], dtype=int)
        st[:, 3:6] = -1
        st[:, 0:3] = -1000
        st[:, 6] = -10
    else:
        st = np.array(bbox_3d_state_3d, dtype=np.float32, copy=True).reshape(n, 7)
        if not bottom_center_done:
            st[:, 1] = st[:, 1] + np.float32(0.5) * st[:, 4]
    if thetas is None:
        thetas = np.ones(n) * -10
    text = ''
    if len(scores) > 0:
        for i in range(n):
            if scores[i] < threshold:
                continue
            b = bbox_2d[i]
            text += ('{} -1 -1 {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f} {} \n').format(
                obj_types[i], st[i][-1], b[0], b[1], b[2], b[3], st[i][4], st[i][3], st[i][5], st[i][0], st[i][1], st[i][2],
                thetas[i], float(scores[i]))      # the reference formats a 0-d tensor: python-float repr of the fp32 value
    return text


def write_result_to_file(base_result_path, index, scores, bbox_2d, bbox_3d_state_3d=None, thetas=None,
                         obj_types=('Car', 'Pedestrian', 'Cyclist'), threshold=0.4, bottom_center_done=False):
    """Reference signature (data/kitti/utils.py:162) plus ``bottom_center_done``."""
    with open(os.path.join(base_result_path, '%06d.txt' % index), 'w') as f:
        f.write(format_result(scores, bbox_2d, bbox_3d_state_3d, thetas, obj_types, threshold, bottom_center_done))
