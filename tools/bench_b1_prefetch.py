"""MI355X experiment: does warming the NEXT layer's weights in every XCD's L2 while the current layer computes shorten a batch-1 forward?

    python tools/bench_b1_prefetch.py [mono|stereo] [blocks per XCD]

The model's `forward_device` is captured into a hipGraph twice: as it is, and with a side-stream launch of tools/l2_touch.hip (reads layer i + 1's packed
weights -- and register image -- through all eight L2s) forked right before layer i's convolution and joined at the end.  Reports the median replay time of both
graphs, alternating, and the bytes touched per forward.  An UPPER bound of what a prefetch issued from inside the kernels' tails could save: the touch kernel runs
beside the layer, costs no issue slots of the layer's own waves, and warms every XCD."""
import ctypes
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops  # noqa: E402
from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT  # noqa: E402
import visualdet3d_amd.networks.detectors  # noqa: E402,F401
from visualdet3d_amd.utils import synthetic as syn  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'mono'
per_xcd = int(sys.argv[2]) if len(sys.argv) > 2 else 4
empty = len(sys.argv) > 3 and sys.argv[3] == 'empty'      # the same fork / launch / join structure, the touch kernel reads nothing: the structure's own cost
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'l2_touch.so'))
lib.l2_touch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

tmp = tempfile.mkdtemp()
if kind == 'mono':
    cfg = syn.mono3d_cfg(tmp, depth=34, score_thr=0.75, name='GroundAwareYolo3D')
    syn.write_synthetic_priors(tmp, cfg.obj_types, 2)
else:
    cfg = syn.stereo3d_cfg(tmp, depth=34, score_thr=0.75, nms_iou_thr=0.4)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
m = DETECTOR_DICT[cfg.name](cfg)
m.load_state_dict(syn.seeded_state_dict(m.state_dict(), seed=1, head_std=0.015 if kind == 'mono' else 0.00042))
m = m.cuda().eval()
m.compute_dtype = torch.bfloat16
P2, P3 = syn.kitti_calib(1280, batch=1)
if kind == 'mono':
    inputs = (syn.mono_image(1, 384, 1280, seed=3).cuda(), P2.cuda())
else:
    L, R = syn.stereo_pair(1, 384, 1280, seed=3)
    inputs = (L.cuda(), R.cuda(), P2.cuda())

orig = ops.conv2d
seq = []                      # packed convs in launch order
sink = torch.zeros(4, dtype=torch.int32, device='cuda')


def recording(x, pc, **kw):
    seq.append(pc)
    return orig(x, pc, **kw)


with torch.no_grad():
    for _ in range(3):
        m.forward_device(*inputs)
    ops.conv2d = recording
    m.forward_device(*inputs)
    ops.conv2d = orig
torch.cuda.synchronize()
order = list(seq)
touched = [0]


def capture(warm):
    idx = [0]
    side = torch.cuda.Stream()

    def conv(x, pc, **kw):
        i = idx[0]
        idx[0] += 1
        if warm and i + 1 < len(order):
            nxt = order[i + 1]
            main = torch.cuda.current_stream()
            side.wait_stream(main)                       # runs beside layer i, not earlier
            with torch.cuda.stream(side):
                for t in (nxt.w, nxt.w_frag):
                    if t is not None and t.numel() * t.element_size() <= (8 << 20):      # (a 36 MB head panel does not fit a 4 MB L2: skipped)
                        nb = t.numel() * t.element_size()
                        lib.l2_touch(t.data_ptr(), 0 if empty else nb, per_xcd, sink.data_ptr(), side.cuda_stream)
                        touched[0] += nb
        return orig(x, pc, **kw)

    ops.conv2d = conv
    try:
        with torch.no_grad():
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                idx[0] = 0
                m.forward_device(*inputs)
                s.wait_stream(side)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            touched[0] = 0
            with torch.cuda.graph(g):
                idx[0] = 0
                out = m.forward_device(*inputs)
                torch.cuda.current_stream().wait_stream(side)
    finally:
        ops.conv2d = orig
    return g, out


# side streams of the model itself off for the mono / stereo comparison?  No: the product configuration as it is
g0, out0 = capture(False)
g1, out1 = capture(True)
nb = touched[0]


def timeit(g, n=200):
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {0: [], 1: []}
for r in range(5):
    res[0].append(timeit(g0))
    res[1].append(timeit(g1))
med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
same = all(torch.equal(a, b) for a, b in zip(out0, out1))
print('%s batch 1, %d conv launches per forward, %.1f MB of next-layer weights touched per forward through all 8 L2s (%d blocks per XCD)' % (kind, len(order), nb / 1e6, per_xcd))
print('replay, plain graph:        %s  median %.4f ms' % (' '.join('%.4f' % v for v in res[0]), med(res[0])))
print('replay, with %s:    %s  median %.4f ms' % ('EMPTY side launches' if empty else 'L2 warmers', ' '.join('%.4f' % v for v in res[1]), med(res[1])))
print('results identical: %s' % same)
