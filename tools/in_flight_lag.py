"""EXPERIMENT: two one-stream replicas in flight (bench.py's default), the host reading the record 1, 2 or 3 steps behind the step it enqueues.
    python tools/in_flight_lag.py     (measured: 3.565 / 3.556 / 3.589 | 3.530 / 3.566 / 3.545 | 3.533 / 3.541 / 3.559 ms per step: no difference worth a deeper ring)"""
import sys, time, types, torch
sys.path.insert(0, '.')
import bench
from visualdet3d_amd.networks.pipelines.in_flight import InFlight
args = types.SimpleNamespace(dtype='bf16', batch=8, height=384, width=1280)
dev = torch.device('cuda', 0)
from visualdet3d_amd.utils import synthetic as syn
L, R = syn.stereo_pair(8, 384, 1280, seed=100)
P2, _ = syn.kitti_calib(1280, batch=8)
inputs = (L.to(dev), R.to(dev), P2.to(dev))
steps = []
for r in range(2):
    m = bench.build_model(args, dev)[0]
    m.core.overlap_neck = False; m.bbox_head.overlap_towers = False
    steps.append(bench.Stepper(m, inputs, 8, dev))
pipe = InFlight(steps)
def run(n, lag):
    first = pipe.submitted
    for j in range(n):
        t = pipe.submit()
        if j >= lag: pipe.counts(t - lag)
    for j in range(max(0, n - lag), n): c = pipe.counts(first + j)
    return c
res = {1: [], 2: [], 3: []}
for rnd in range(4):
    for lag in (1, 2, 3):
        run(6, lag); torch.cuda.synchronize(); t0 = time.perf_counter(); run(40, lag); torch.cuda.synchronize()
        if rnd: res[lag].append((time.perf_counter() - t0) / 40 * 1e3)
for k, v in res.items(): print('lag %d: %s' % (k, ' '.join('%.3f' % t for t in v)))
