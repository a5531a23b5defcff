"""Generates tests/golden/postpath_cases.npz with the REFERENCE's own code (imported from /root/reference through
oracle/ref_shim.py): for seeded detections and KITTI-like calibrations, the body of test_one
(networks/pipelines/evaluators.py:112-129: BackProjection, BBox3dProjector thetas, 2D shift/rescale) followed by
data/kitti/utils.py write_result_to_file.  Run here: python -m oracle.make_golden_postpath"""
import os
import tempfile

import numpy as np
import torch

from oracle import ref_shim

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
    ref_shim.load()
    from visualDet3D.networks.utils import BackProjection, BBox3dProjector
    from visualDet3D.data.kitti.utils import write_result_to_file
    rng = np.random.default_rng(3)
    out = {}
    names = ['Car', 'Pedestrian', 'Cyclist']
    for case in range(4):
        n = [0, 1, 17, 60][case]
        # original KITTI calibration, then CropTop(100) + Resize(288 / 275) as the test pipeline does (stereo_augmentator.py)
        original_P = np.array([[721.5377, 0, 609.5593 + case, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]])
        P2 = original_P.copy()
        P2[1, 2] -= 100
        P2[1, 3] -= 100 * P2[2, 3]
        sc = 288.0 / 275.0
        P2[0] *= sc
        P2[1] *= sc
        cx, cy = rng.uniform(0, 1280, n), rng.uniform(100, 288, n)
        z = rng.uniform(3, 60, n)
        bw, bh = rng.uniform(10, 200, n), rng.uniform(10, 120, n)
        bbox = np.stack([cx - bw, cy - bh, cx + bw, cy + bh, cx, cy, z, rng.uniform(1.4, 2, n), rng.uniform(1.3, 1.9, n),
                         rng.uniform(3, 5, n), rng.uniform(-np.pi, np.pi, n)], 1).astype(np.float32).reshape(n, 11)
        scores = rng.uniform(0.3, 1.0, n).astype(np.float32)
        labels = rng.integers(0, 3, n)
        obj_names = [names[i] for i in labels]
        tb = torch.from_numpy(bbox.copy())
        bbox_2d = tb[:, 0:4]
        st3 = BackProjection()(tb[:, 4:], P2)
        _, _, thetas = BBox3dProjector()(st3, st3.new(P2))
        scale_x = original_P[0, 0] / P2[0, 0]
        scale_y = original_P[1, 1] / P2[1, 1]
        shift_left = original_P[0, 2] / scale_x - P2[0, 2]
        shift_top = original_P[1, 2] / scale_y - P2[1, 2]
        bbox_2d[:, 0:4:2] += shift_left
        bbox_2d[:, 1:4:2] += shift_top
        bbox_2d[:, 0:4:2] *= scale_x
        bbox_2d[:, 1:4:2] *= scale_y
        tmp = tempfile.mkdtemp()
        write_result_to_file(tmp, case, torch.from_numpy(scores), bbox_2d, st3, thetas, obj_names)
        text = open(os.path.join(tmp, '%06d.txt' % case)).read()
        rows = torch.cat([bbox_2d, st3, thetas[:, None]], dim=1).numpy()     # st3[:, 1] already moved to the bottom centre by the writer
        out['c%d_bbox' % case], out['c%d_scores' % case], out['c%d_labels' % case] = bbox, scores, labels
        out['c%d_P2' % case], out['c%d_origP' % case] = P2, original_P
        out['c%d_rows' % case] = rows
        out['c%d_text' % case] = np.frombuffer(text.encode(), dtype=np.uint8)
        print('case', case, 'n', n, 'lines', text.count('\n'))
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'postpath_cases.npz'), **out)


if __name__ == '__main__':
    main()
