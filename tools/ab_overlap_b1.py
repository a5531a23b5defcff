"""MI355X: the head towers / stereo neck on side streams or on the main stream, wall clock per `test_forward_batched` call (hipGraph replay), small to large batches.
A forked launch inside a hipGraph costs ~13.6 us (profiles/r06_b1_l2_warm_experiment.txt): an overlap only pays for a long branch.
    python tools/ab_overlap_b1.py [mono|stereo] [batches...]"""
import sys
import time

import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import bench_b1 as bb  # noqa: E402

kinds = [a for a in sys.argv[1:] if a in ('mono', 'stereo')] or ['mono', 'stereo']
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4]
for kind in kinds:
    for B in batches:
        for towers in (True, False):
            for neck in ((True, False) if kind == 'stereo' else (True,)):
                m, x = bb.build(kind)
                m.bbox_head.overlap_towers = towers
                if hasattr(m.core, 'overlap_neck'):
                    m.core.overlap_neck = neck
                xs = [t.repeat(B, *([1] * (t.dim() - 1))).contiguous() for t in x]
                with torch.no_grad():
                    for _ in range(5):
                        m.test_forward_batched(*xs)
                    torch.cuda.synchronize()
                    ts = []
                    for r in range(3):
                        t0 = time.perf_counter()
                        for _ in range(50):
                            m.test_forward_batched(*xs)
                        torch.cuda.synchronize()
                        ts.append((time.perf_counter() - t0) / 50 * 1e3)
                print('%s B=%d towers=%s neck=%s: %.4f ms per call (%s)' % (kind, B, towers, neck, sorted(ts)[1], ' '.join('%.4f' % t for t in ts)), flush=True)
                del m
                torch.cuda.empty_cache()
