"""Generates tests/golden/rotate_iou_cases.npz by executing the REFERENCE's own numba.cuda device functions
(evaluator/kitti/rotate_iou.py devRotateIoUEval and everything below it, imported from /root/reference) as plain Python through
the numba stand-in of oracle/stubs (cuda.local.array -> numpy float32 arrays).  Pairs for which the reference overflows its
intersection buffer (IndexError here, undefined behaviour on a GPU) are marked NaN and excluded by the tests.
Run here: python -m oracle.make_golden_rotate_iou"""
import os

import numpy as np

from oracle import ref_shim

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
    ref_shim.load()
    from visualDet3D.evaluator.kitti import rotate_iou as ri
    rng = np.random.default_rng(17)
    N, K = 40, 30
    boxes = np.stack([rng.uniform(-10, 10, N), rng.uniform(0, 40, N), rng.uniform(1.4, 2.2, N), rng.uniform(3, 5, N), rng.uniform(-np.pi, np.pi, N)], 1).astype(np.float32)
    q = boxes[rng.integers(0, N, K)].copy()
    q[:, :2] += rng.normal(0, 1.0, (K, 2)).astype(np.float32)          # overlapping neighbours
    q[:, 4] += rng.normal(0, 0.3, K).astype(np.float32)
    q[:5] = boxes[:5]                                                   # exact duplicates
    q[5:8, 4] = 0.0                                                     # axis aligned
    out = {'boxes': boxes, 'query': q}
    for crit in (-1, 0, 1, 2):
        iou = np.zeros((N, K), np.float32)
        bad = 0
        with np.errstate(all='ignore'):
            for n in range(N):
                for k in range(K):
                    try:
                        iou[n, k] = ri.devRotateIoUEval(q[k], boxes[n], crit)
                    except IndexError:
                        iou[n, k] = np.nan
                        bad += 1
        out['iou_crit%d' % crit] = iou
        print('criterion', crit, 'nonzero', int((iou > 0).sum()), 'overflow pairs', bad, 'max', float(np.nanmax(iou)))
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'rotate_iou_cases.npz'), **out)


if __name__ == '__main__':
    main()
