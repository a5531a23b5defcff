"""What does the work on the side streams cost the headline step?  Each side part (neck s4, neck s8, the cls tower) is run TWICE (idempotent: the second run
rewrites the same values) and the step timed: a part that is hidden under the main stream's chip-filling kernels adds nothing, a part that competes with
them adds its own duration.   python tools/side_work_cost.py  (prints ms per step for every variant, round-robin, same process)"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402


def main():
    args = types.SimpleNamespace(dtype='bf16', batch=8, height=384, width=1280)
    dev = torch.device('cuda', 0)
    from visualdet3d_amd.utils import synthetic as syn
    model, cfg, sd = bench.build_model(args, dev)
    L, R = syn.stereo_pair(8, 384, 1280, seed=100)
    P2, _ = syn.kitti_calib(1280, batch=8)
    inputs = (L.to(dev), R.to(dev), P2.to(dev))
    neck, head = model.core.neck, model.bbox_head
    o4, o8, ocls = neck.part_s4, neck.part_s8, head._cls_forward_nhwc
    variants = {
        'base': {},
        's4 x2': {'s4': 2}, 's8 x2': {'s8': 2}, 'cls x2': {'cls': 2}, 'cls x3': {'cls': 3}, 'all x2': {'s4': 2, 's8': 2, 'cls': 2},
    }
    steppers = {}
    for name, v in variants.items():
        neck.part_s4 = (lambda x, st, n=v.get('s4', 1): [o4(x, st) for _ in range(n)][-1])
        neck.part_s8 = (lambda x, st, n=v.get('s8', 1): [o8(x, st) for _ in range(n)][-1])
        head._cls_forward_nhwc = (lambda feat, n=v.get('cls', 1): [ocls(feat) for _ in range(n)][-1])
        steppers[name] = bench.Stepper(model, inputs, 8, dev)
    res = {k: [] for k in variants}
    for rnd in range(4):
        for name, st in steppers.items():
            st.run(5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st.run(20)
            torch.cuda.synchronize()
            if rnd:
                res[name].append((time.perf_counter() - t0) / 20 * 1e3)
    for name, v in res.items():
        print('%-8s %s   median %.3f ms' % (name, ' '.join('%.3f' % t for t in v), sorted(v)[len(v) // 2]), flush=True)


if __name__ == '__main__':
    main()
