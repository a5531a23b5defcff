"""Generates tests/golden/post_opt_cases.npz by running the REFERENCE's own hill-climbing code (imported from
/root/reference through oracle/ref_shim.py; the numba stub makes the @jit functions plain Python):
  * literal_theta : the worked example at the bottom of lib/fast_utils/hill_climbing.py:125-141;
  * inputs/outputs: 40 seeded single boxes through post_opt (hill_climbing.py:7-23);
  * <case>_f<k>_boxes: AnchorBasedDetection3DHead._post_process (heads/detection_3d_head.py:294-308) applied to the
    detections stored in the mono golden files (label 0, depth > 3 m boxes get a refined alpha).
Run here (needs /root/reference): python -m oracle.make_golden_postopt"""
import os
import types

import numpy as np
import torch

from oracle import ref_shim
from visualdet3d_amd.utils import synthetic as syn

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
HEAD_CASES = ['groundaware_r34_96x320', 'groundaware_r34_384x1280', 'yolo3d_dcn_r34_96x320']


def main():
    ref_shim.load()
    from visualDet3D.networks.lib.fast_utils import hill_climbing as hc
    from visualDet3D.networks.heads.detection_3d_head import AnchorBasedDetection3DHead
    from visualDet3D.networks.utils import BackProjection
    out = {}

    p2 = np.array([[5.02790613e+02, 0, 4.29568996e+02, 3.25392427e+01], [0, 5.02790613e+02, 5.72491378e+01, -5.99834524e-01],
                   [0, 0, 1, 4.98101600e-03], [0, 0, 0, 1]])
    box = np.array([490.3174, 64.63407, 568.4109, 105.2571])
    args = (528.2042846679688, 82.82894134521484, 20.556593, 1.5336921, 1.4364641, 3.3523552, 1.6921594)
    out['literal_theta'] = np.float64(hc.post_optimization(p2, np.linalg.inv(p2), box, *args, step_r_init=0.4, r_lim=0.01)[0])

    rng = np.random.default_rng(0)
    P2n = syn.kitti_calib(1280, batch=1)[0][0].numpy()
    ins, outs = [], []
    for _ in range(40):
        cx, cy, z = np.float32(rng.uniform(100, 1180)), np.float32(rng.uniform(100, 250)), rng.uniform(5, 50)
        w, h, l, alpha = rng.uniform(1.4, 1.9), rng.uniform(1.3, 1.8), rng.uniform(3, 4.8), rng.uniform(-3.1, 3.1)
        bw, bh = rng.uniform(30, 200) * 20 / z, rng.uniform(20, 100) * 20 / z
        b2 = torch.tensor([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], dtype=torch.float32)
        st = torch.tensor([0.0, 0.0, z, w, h, l, alpha], dtype=torch.float32)       # x3d / y3d are not read by post_opt
        outs.append(hc.post_opt(b2, st, P2n, float(cx), float(cy)).numpy())
        ins.append(np.concatenate([b2.numpy(), [cx, cy], st.numpy()[2:]]))
    out['inputs'], out['outputs'], out['P2'] = np.array(ins, np.float32), np.array(outs, np.float32), P2n

    fake_head = types.SimpleNamespace(backprojector=BackProjection())
    for name in HEAD_CASES:
        g = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        depth, H, W, frames, wseed, iseed = [int(v) for v in g['meta']]
        P2, _ = syn.kitti_calib(W, batch=frames)
        for f in range(frames):
            s, b, l = (torch.from_numpy(g['f%d_%s' % (f, k)]) for k in ('scores', 'boxes', 'labels'))
            _, nb, _ = AnchorBasedDetection3DHead._post_process(fake_head, s, b.clone(), l, P2[f:f + 1])
            out['%s_f%d_boxes' % (name, f)] = nb.numpy()
            print(name, f, 'boxes', len(b), 'changed', int((nb[:, 10] != b[:, 10]).sum()),
                  'max |dalpha|', float((nb[:, 10] - b[:, 10]).abs().max()) if len(b) else 0.0)
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'post_opt_cases.npz'), **out)


if __name__ == '__main__':
    main()
