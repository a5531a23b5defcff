"""The reference's call pattern -- ONE frame per `module([...])` call (networks/pipelines/testers.py:15-42) -- timed wall-clock on one
MI355X, including the result sync: BASELINE config 1 (GroundAwareYolo3D R34, 384x1280, batch 1) and the stereo pair of config 2 at batch 1.
    python tools/bench_b1.py [--layers] [--eager] [--calls N] [mono|stereo]
--layers: per-launch HIP-event table of the MFMA families (serial launches), --eager: VD3D_NO_GRAPH behaviour for comparison."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT  # noqa: E402
import visualdet3d_amd.networks.detectors  # noqa: E402,F401
from visualdet3d_amd.utils import synthetic as syn  # noqa: E402

GF = dict(mono=82.10, stereo=473.82)


def build(kind):
    tmp = tempfile.mkdtemp()
    if kind == 'mono':
        cfg = syn.mono3d_cfg(tmp, depth=34, score_thr=0.75, name='GroundAwareYolo3D')
        syn.write_synthetic_priors(tmp, cfg.obj_types, 2)
    else:
        cfg = syn.stereo3d_cfg(tmp, depth=34, score_thr=0.75, nms_iou_thr=0.4)
        syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    m = DETECTOR_DICT[cfg.name](cfg)
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), seed=1, head_std=0.00042 if kind == 'stereo' else 0.015))      # (mono: 25 detections on this frame, as bench.py's C1)
    m = m.cuda().eval()
    m.compute_dtype = torch.bfloat16
    P2, P3 = syn.kitti_calib(1280, batch=1)
    if kind == 'stereo':
        L, R = syn.stereo_pair(1, 384, 1280, seed=3)
        return m, [L.cuda(), R.cuda(), P2.cuda(), P3.cuda()]
    return m, [syn.mono_image(1, 384, 1280, seed=3).cuda(), P2.cuda()]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    calls = int(sys.argv[sys.argv.index('--calls') + 1]) if '--calls' in sys.argv else 200
    args = [a for a in args if not a.isdigit()]
    for kind in ('mono', 'stereo'):
        if args and kind not in args:
            continue
        m, x = build(kind)
        if '--eager' in sys.argv:
            m.use_graph = False
        if '--no-overlap' in sys.argv:           # one stream: kernel durations of a rocprofv3 trace are additive
            m.bbox_head.overlap_towers = False
            if hasattr(m.core, 'overlap_neck'):
                m.core.overlap_neck = False
        with torch.no_grad():
            for _ in range(5):
                out = m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(calls):
                out = m(x)                       # returns after the host has read the detection count: the sync is inside
            dt = (time.perf_counter() - t0) / calls
            # device time of one replay alone (no input copies, no sync per call): back-to-back replays
            dev = None
            if m.use_graph:
                ent = next(iter(m._graph_state()['entries'].values()))
                if ent.graph is not None:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(calls):
                        ent.graph.replay()
                    torch.cuda.synchronize()
                    dev = (time.perf_counter() - t0) / calls
        print('%-7s B=1 384x1280 bf16 %s: %.3f ms per module([...]) call = %.1f img/s, %.1f %% of 2.5 PF whole path; %d detections%s  %s'
              % (kind, 'eager' if not m.use_graph else 'hipGraph cache', dt * 1e3, 1 / dt, GF[kind] / dt / 25e3, out[0].numel(),
                 '; replay alone %.3f ms' % (dev * 1e3) if dev else '', m.graph_stats), flush=True)
        if '--breakdown' in sys.argv and m.use_graph:
            # where the host's time of one call goes: stamps around the graph cache (_graphed) and the host sync (read_counts)
            import visualdet3d_amd.networks.lib.graphed as G
            import visualdet3d_amd.networks.heads.detection_3d_head as H  # noqa: F401
            st = []
            g0, r0 = m._graphed, G.read_counts

            def g1(*a):
                st.append(time.perf_counter()); o = g0(*a); st.append(time.perf_counter()); return o

            def r1(c):
                st.append(time.perf_counter()); o = r0(c); st.append(time.perf_counter()); return o
            m._graphed, G.read_counts = g1, r1
            acc = [0.0] * 5
            with torch.no_grad():
                for _ in range(calls):
                    del st[:]
                    ta = time.perf_counter()
                    m(x)
                    tb = time.perf_counter()
                    if len(st) == 4:
                        for i, d in enumerate((st[0] - ta, st[1] - st[0], st[2] - st[1], st[3] - st[2], tb - st[3])):
                            acc[i] += d
            m._graphed, G.read_counts = g0, r0
            print('  host breakdown per call (us): before the graph cache %.1f | key + input copies + replay enqueue %.1f | result copies enqueue %.1f | '
                  'wait for the counts (GPU time + wake-up) %.1f | slicing + return %.1f' % tuple(a / calls * 1e6 for a in acc), flush=True)
        if '--layers' in sys.argv:
            import bench
            os.environ['VD3D_BENCH_LAYERS'] = '1'
            fam, dom, tot = bench.profile_ops(m, tuple(x[:3]) if kind == 'stereo' else tuple(x), reps=3)
            for k, v in fam.items():
                print('  family %-16s %3d launches %8.1f us %7.1f TF/s' % (k, round(v['launches']), v['secs'] * 1e6, v['flops'] / max(v['secs'], 1e-12) / 1e12), flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
