"""EXPERIMENT (became bench.py --in-flight 2): K headline steps in flight -- two replicas of the detector (same weights), each with its own hipGraph, static buffers and scratch, replayed
alternately on two streams: step i + 1 starts while step i's tail (select / NMS on 8 CUs, pack, D2H) and its partially-filled rounds still run.
    python tools/two_in_flight.py [K]    (ms per step: one in flight | K in flight, round-robin, same process)"""
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
    L, R = syn.stereo_pair(8, 384, 1280, seed=100)
    P2, _ = syn.kitti_calib(1280, batch=8)
    reps = []
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for r in range(K):
        model, cfg, sd = bench.build_model(args, dev)
        if K >= 2:                      # K in flight: one-stream graphs (bench.py main(): the intra-step forks cost there)
            model.core.overlap_neck = False
            model.bbox_head.overlap_towers = False
        inputs = (L.to(dev).clone(), R.to(dev).clone(), P2.to(dev).clone())
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            st = bench.Stepper(model, inputs, 8, dev)
        torch.cuda.synchronize()
        reps.append((st, s))

    def run_one(n):         # (replica 0 alone; with K >= 2 it has no intra-step forks: NOT bench.py's one_in_flight)
        return reps[0][0].run(n)

    def run_two(n):
        return bench.run_in_flight(reps, n)

    res = {'one': [], 'two': []}
    for rnd in range(4):
        for name, fn in (('one', run_one), ('two', run_two)):
            fn(6)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c = fn(40)
            torch.cuda.synchronize()
            if rnd:
                res[name].append((time.perf_counter() - t0) / 40 * 1e3)
        print('detections in the last step:', int(c.sum()), flush=True)
    for k, v in res.items():
        print('%s in flight: %s ms per step' % (k, ' '.join('%.3f' % t for t in v)), flush=True)


if __name__ == '__main__':
    main()
