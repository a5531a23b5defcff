"""EXPERIMENT: two headline steps in flight -- two replicas of the detector (same weights), each with its own hipGraph, static buffers and scratch, replayed
alternately on two streams: step i + 1 starts while step i's tail (select / NMS on 8 CUs, pack, D2H) and its partially-filled rounds still run.
    python tools/two_in_flight.py        (ms per step: one in flight | two in flight, round-robin, same process)"""
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
    for r in range(2):
        model, cfg, sd = bench.build_model(args, dev)
        inputs = (L.to(dev).clone(), R.to(dev).clone(), P2.to(dev).clone())
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            st = bench.Stepper(model, inputs, 8, dev)
        torch.cuda.synchronize()
        reps.append((st, s))

    def run_one(n):
        return reps[0][0].run(n)

    def run_two(n):
        counts = None
        for i in range(n):
            st, s = reps[i & 1]
            with torch.cuda.stream(s):
                st.forward_step()
                st.pinned_ring[0][0].copy_(st.pack_static, non_blocking=True)
                st.copied[0].record(s)
            if i >= 1:
                pst = reps[(i - 1) & 1][0]
                pst.copied[0].synchronize()
                counts = pst.check(pst.pinned_ring[0])
        if n >= 1:
            pst = reps[(n - 1) & 1][0]
            pst.copied[0].synchronize()
            counts = pst.check(pst.pinned_ring[0])
        return counts

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
