"""bench.py -- images/sec of the YOLOStereo3D forward path (BASELINE.json configs[1]: Stereo3D ResNet-34,
384x1280 stereo pairs, bf16, batch 8 per MI355X), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one test_forward_batched-equivalent pass over one batch of synthetic pairs already resident in HBM:
stem -> ResNet-34 (L and R stacked) -> cost volumes -> ghost pyramid -> head towers -> device-side decode/NMS ->
(N > 1: RCCL all_gather of the padded detections, the only collective) -> one host sync for the counts.
Two steps are in flight per GPU by default (--in-flight 2: two replicas of the detector with the same weights, step i on replica i & 1, each on its own stream;
every step is still one forward over `batch` pairs whose record is read and checked inside the timed region; `one_in_flight` in the line = the steps one at a time).
Prints ONE JSON line (rank 0) with the contract fields plus `roofline` and `cpu_baseline`.

Environment knobs (A/B and diagnostics; none is needed for the contract run):
  VD3D_BENCH_FORCE_DIST=1      the RCCL path at world 1 (process group, collective, host-ordered loop)
  VD3D_BENCH_COMM_STREAM=1     two in flight with the GPU-ordered comm-stream loop of rounds 1 - 5 (the default orders the collective from the host)
  VD3D_BENCH_SIDE_STREAMS=1    two in flight WITH the intra-step side streams (default: the replicas are one-stream graphs)
  VD3D_BENCH_REPLICAS=k        k replicas for configs 3 / 5 (default 2)
  VD3D_BENCH_SYNC_EACH_STEP=1  one in flight: host sync and check BEFORE the next replay is launched (rounds 1 - 5)
  VD3D_BENCH_NONECK / NOTOWER / NOSELECT=1   single intra-step forks off;  VD3D_BENCH_NOSIDE=1  no side pass before the capture
  VD3D_BENCH_NO_LEGACY=1       skip the round-5 workload of configs 3 / 5;  VD3D_BENCH_SEED=n  input seed
  VD3D_BENCH_DUMP=path         (tests) the last step's host record next to forward_device's own results
  VD3D_BENCH_LAYERS / VD3D_BENCH_LAYERS_OTHER / VD3D_BENCH_DEBUG=1   per-launch tables / progress on stderr
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import tempfile
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from visualdet3d_amd.networks.pipelines.in_flight import CapturedStep, InFlight  # noqa: E402

GFLOP_PER_PAIR = 473.82          # conv/GEMM 2*MAC per 384x1280 pair (Stereo3D R34)
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0           # MI355X HBM3E peak (MI355X_MICROARCH.md)
ACHIEVABLE_HBM_GBPS = 6300.0     # what a streaming kernel reaches on this part (same guide: ~6.3 TB/s copy / fill)
C_float3 = ctypes.c_float * 3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None,
                    help='ranks (one process per GPU).  Without a torchrun environment (WORLD_SIZE unset) and N > 1 this process re-executes itself under '
                         '`python -m torch.distributed.run --nproc-per-node N`; under torchrun N must equal WORLD_SIZE (a mismatch is an error, never a '
                         'silent one-GPU line).  Default: WORLD_SIZE, or 1')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='stereo pairs per GPU per step')
    ap.add_argument('--height', type=int, default=384)
    ap.add_argument('--width', type=int, default=1280)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--regions', type=int, default=3,
                    help='the timed region (EXACTLY --steps steps between barrier + synchronize on both sides) is run this many times back to '
                         'back; value / ms_per_step are the MEDIAN region, `spread` carries all of them (box-to-box and run-to-run noise is +-3 %%: '
                         'a single sample cannot carry a 1 %% claim)')
    ap.add_argument('--no-graph', action='store_true', help='launch kernels eagerly instead of replaying a hipGraph')
    ap.add_argument('--in-flight', type=int, default=2, choices=[1, 2],
                    help='steps in flight per GPU.  2 (default): two replicas of the detector (same weights; each its own hipGraph, static buffers and scratch) '
                         'are replayed alternately on two streams, so that step i + 1 starts under the tail of step i (select / NMS on 8 CUs, pack, D2H) and its '
                         'partially-filled rounds.  Every step is still one forward over `batch` pairs, read back and checked inside the timed region.  1: one '
                         'replica, steps back to back (rounds 1 - 5; also reported in the line as `one_in_flight` for the resident feed).  --no-graph / --no-overlap run with 1.')
    ap.add_argument('--no-overlap', action='store_true',
                    help='no side streams (cls tower / stereo neck run serially on the main stream): the configuration whose '
                         'rocprofv3 --kernel-trace durations are additive (profiles/*_serial_*), NOT the headline configuration')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true',
                    help='no GPU work: time the CPU baseline(s) on this host and print them (used in the build container, where '
                         'the reference tree exists, to calibrate the "port" against the reference itself)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--feed', default='resident', choices=['resident', 'host'],
                    help="resident (default, the headline): synthetic network inputs already in HBM.  host: every step uploads 2 x batch "
                         "uint8 camera frames of the network's size (384 x 1280 x 3) from pinned host memory into a double-buffered device ring on a "
                         "copy stream and runs vd3d_preprocess_image (crop / resize / normalise) inside the captured step -- the "
                         "PCIe-inclusive rate, the one thing that differs between 1 and 8 ranks of a node")
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the extra timed runs of BASELINE configs 3 and 5 (attached to the line as other_configs at N = 1)')
    return ap.parse_args()


def resolve_world(gpus, env, argv, executable=sys.executable):
    """What `--gpus N` means given the environment -> ('run', world) | ('exec', command list).
      * WORLD_SIZE set (torchrun started us): world = WORLD_SIZE; `--gpus` given and different -> ValueError;
      * WORLD_SIZE unset, N > 1: the command that re-runs this script under torch.distributed.run with N ranks on this node;
      * otherwise one rank."""
    ws = env.get('WORLD_SIZE')
    if ws is not None:
        world = int(ws)
        if gpus is not None and gpus != world:
            raise ValueError('bench.py --gpus %d under WORLD_SIZE=%d: the launcher and the flag disagree (launch with --nproc-per-node %d)' % (gpus, world, gpus))
        return 'run', world
    if gpus is None or gpus == 1:
        return 'run', 1
    if gpus < 1:
        raise ValueError('--gpus %d' % gpus)
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return 'exec', [executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus), '--master-addr', '127.0.0.1',
                    '--master-port', str(port)] + list(argv)


def build_model(args, device):
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3D
    from visualdet3d_amd.utils import synthetic as syn
    tmp = tempfile.mkdtemp()
    cfg = syn.stereo3d_cfg(tmp, depth=34, score_thr=0.75, nms_iou_thr=0.4)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    model = Stereo3D(cfg)
    sd = syn.seeded_state_dict(model.state_dict(), seed=1, head_std=0.00042)
    model.load_state_dict(sd)
    model = model.to(device).eval()
    model.compute_dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    return model, cfg, sd


def dcn_sampling_stats(offset, layout, kernel, stride, padding, dilation, in_hw):
    """What a DCN launch samples: offsets [B,Ho,Wo,2K] ('nhwc') or [B,2K,Ho,Wo] ('nchw'), (dy, dx) per tap (the reference's layout,
    lib/ops/dcn/src/deform_conv_cuda_kernel.cu: offset channel 2k = y, 2k + 1 = x).  -> (sum of squared offsets, offsets counted, bilinear corners that
    lie inside the input, corners counted).  A corner outside the image contributes zero and costs no memory traffic: a workload whose
    corners are mostly outside under-exercises the gather."""
    off = offset.detach().float()
    if layout == 'nchw':
        off = off.permute(0, 2, 3, 1)
    B, Ho, Wo, K2 = off.shape
    kh, kw = kernel
    K = kh * kw
    off = off[..., :2 * K].reshape(B, Ho, Wo, K, 2)
    dev = off.device
    ys = (torch.arange(Ho, device=dev, dtype=torch.float32) * stride[0] - padding[0]).view(1, Ho, 1, 1)
    xs = (torch.arange(Wo, device=dev, dtype=torch.float32) * stride[1] - padding[1]).view(1, 1, Wo, 1)
    ky = (torch.arange(K, device=dev) // kw).float().view(1, 1, 1, K) * dilation[0]
    kx = (torch.arange(K, device=dev) % kw).float().view(1, 1, 1, K) * dilation[1]
    y0 = torch.floor(ys + ky + off[..., 0])
    x0 = torch.floor(xs + kx + off[..., 1])
    H, W = in_hw
    yin = [(y0 >= 0) & (y0 <= H - 1), (y0 + 1 >= 0) & (y0 + 1 <= H - 1)]
    xin = [(x0 >= 0) & (x0 <= W - 1), (x0 + 1 >= 0) & (x0 + 1 <= W - 1)]
    inside = sum(int((a & b).sum()) for a in yin for b in xin)
    return float(off.double().pow(2).sum()), off.numel(), inside, 4 * y0.numel()


def profile_ops(model, inputs, reps=3, verbose_env='VD3D_BENCH_LAYERS'):
    """Per-launch HIP-event timing of the MFMA kernel families on the launch stream, side streams off (durations must not
    overlap): every launch through hip_ops.conv2d (the implicit-GEMM family), deform_conv_general (fused DCN), deform_columns
    (DCN sampling half), km3d_head_fused and conv2d_pair (DLA level0 + level1, counted in the conv family).  Returns a dict: conv family (flops, seconds, launches, algorithmic bytes, dominant
    layer shape) + per-family totals of the other entries."""
    from visualdet3d_amd import hip_ops as ops
    records = []     # (family, flops, start, end, bytes, label)

    def ev():
        return torch.cuda.Event(enable_timing=True)

    orig = dict(conv2d=ops.conv2d, dcn=ops.deform_conv_general, cols=ops.deform_columns, head=ops.km3d_head_fused, pair=ops.conv2d_pair,
                bottleneck=ops.conv2d_bottleneck)

    def conv2d(x, pc, out=None, residual=None, relu=False, out_f32=False):
        s, e = ev(), ev()
        s.record()
        o = orig['conv2d'](x, pc, out=out, residual=residual, relu=relu, out_f32=out_f32)
        e.record()
        B, Ho, Wo, Co = o.shape
        nbytes = (x.shape[0] * x.shape[1] * x.shape[2] * pc.Cin * x.element_size() + pc.Cout * pc.kh * pc.kw * pc.Cin * x.element_size()
                  + o.numel() * o.element_size() + (residual.numel() * residual.element_size() if residual is not None else 0))
        records.append(('conv', 2.0 * B * Ho * Wo * Co * pc.kh * pc.kw * pc.Cin, s, e, nbytes,
                        '%dx%d s%d %4d->%4d @ %dx%dx%d' % (pc.kh, pc.kw, pc.stride, pc.Cin, pc.Cout, B, Ho, Wo)))
        return o

    dcn_stats = [0.0, 0, 0, 0, 0, 0]      # first pass only: sum off^2, n offsets, corners inside, corners, (corners inside, corners) at ZERO offsets
    dcn_layer_rms = []

    def add_stats(offset, layout, kernel, kw, in_hw):
        if len(records) < first_pass_len[0]:
            geo = (kernel, kw.get('stride', (1, 1)), kw.get('padding', (0, 0)), kw.get('dilation', (1, 1)), in_hw)
            st = dcn_sampling_stats(offset, layout, *geo)
            st0 = dcn_sampling_stats(torch.zeros_like(offset), layout, *geo)     # the zero-padding border alone (what the reference's zero-initialised offset convs sample)
            for i in range(4):
                dcn_stats[i] += st[i]
            dcn_stats[4] += st0[2]
            dcn_stats[5] += st0[3]
            dcn_layer_rms.append(round((st[0] / st[1]) ** 0.5, 3))

    first_pass_len = [1 << 30]

    def dcn(x, pd, offset, mask, out, layout, **kw):
        s, e = ev(), ev()
        s.record()
        o = orig['dcn'](x, pd, offset, mask, out, layout, **kw)
        e.record()
        add_stats(offset, layout, (pd.kh, pd.kw), kw, (x.shape[1], x.shape[2]) if layout == 'nhwc' else (x.shape[2], x.shape[3]))
        if layout == 'nhwc':
            B, Ho, Wo, Co = o.shape
        else:
            B, Co, Ho, Wo = o.shape
        nb = (x.numel() * x.element_size() + offset.numel() * 4 + (mask.numel() * 4 if mask is not None else 0)
              + Co * pd.kh * pd.kw * pd.Cg * x.element_size() + o.numel() * o.element_size())
        records.append(('dcn', 2.0 * B * Ho * Wo * Co * pd.kh * pd.kw * pd.Cg, s, e, nb,
                        'dcn %dx%d %4d->%4d @ %dx%dx%d' % (pd.kh, pd.kw, pd.Cg, Co, B, Ho, Wo)))
        return o

    def cols(x, *a, **kw):
        s, e = ev(), ev()
        s.record()
        o = orig['cols'](x, *a, **kw)
        e.record()
        names = ('stride', 'padding', 'dilation')
        kw_all = dict(zip(names, a[3:6]), **{k: v for k, v in kw.items() if k in names})
        add_stats(a[0], kw.get('offset_layout', 'nhwc'), a[2], kw_all, (x.shape[1], x.shape[2]))
        records.append(('dcn_columns', 0.0, s, e, x.numel() * x.element_size() + o.numel() * o.element_size(), 'dcn columns %s' % (tuple(o.shape),)))
        return o

    def head(x, pc_first, w2_packed, b2, n_out):
        s, e = ev(), ev()
        s.record()
        o = orig['head'](x, pc_first, w2_packed, b2, n_out)
        e.record()
        B, H, W, _ = x.shape
        fl = 2.0 * B * H * W * (pc_first.Cout * 9 * pc_first.Cin + 256 * sum(int(n) for n in n_out))
        records.append(('km3d_head_fused', fl, s, e, 0, 'fused head %d->%d x %d @ %dx%dx%d' % (pc_first.Cin, 256, len(n_out), B, H, W)))
        return o

    def pair(x, pc_a, pc_b, relu_a=True, relu_b=True):
        s, e = ev(), ev()
        s.record()
        o = orig['pair'](x, pc_a, pc_b, relu_a=relu_a, relu_b=relu_b)
        e.record()
        B, H, W, _ = x.shape
        _, Ho, Wo, Co = o.shape
        fl = 2.0 * B * 9 * (H * W * pc_a.Cout * pc_a.Cin + Ho * Wo * Co * pc_b.Cin)
        nbytes = x.numel() * x.element_size() + o.numel() * o.element_size()
        records.append(('conv', fl, s, e, nbytes, '3x3 s1 %d->%d + 3x3 s2 %d->%d (one launch) @ %dx%dx%d' % (pc_a.Cin, pc_a.Cout, pc_b.Cin, Co, B, H, W)))
        return o

    def bottleneck(x, pc1, pc2, pc3, pc_ds=None, out=None):
        s, e = ev(), ev()
        s.record()
        o = orig['bottleneck'](x, pc1, pc2, pc3, pc_ds, out=out)
        e.record()
        B, H, W, Cx = x.shape
        fl = 2.0 * B * H * W * (Cx * 64 + 576 * 64 + 64 * 256 + (Cx * 256 if pc_ds is not None else 0))
        records.append(('conv', fl, s, e, x.numel() * x.element_size() + o.numel() * o.element_size(),
                        'bottleneck %d->64->64->256%s (one launch) @ %dx%dx%d' % (Cx, ' + ds' if pc_ds is not None else '', B, H, W)))
        return o

    # the HBM-bound stages north_star names (stem, cosine volumes, fused cost volume, ghost depth-wise convs, head selection + NMS): same
    # per-launch events, algorithmic bytes = every operand read once + every result written once
    def nbytes(*ts):
        return float(sum(t.numel() * t.element_size() for t in ts if t is not None))

    def hbm_wrap(name, label, traffic):
        fn = getattr(ops, name)
        orig[name] = fn

        def wrapped(*a, **kw):
            s, e = ev(), ev()
            s.record()
            o = fn(*a, **kw)
            e.record()
            hbm_records.append((label(a, kw, o), traffic(a, kw, o), s, e))
            return o
        setattr(ops, name, wrapped)

    hbm_records = []
    imgs_of = lambda a: list(a[0]) if isinstance(a[0], (list, tuple)) else [a[0]]                                      # noqa: E731
    hbm_wrap('stem_conv_pool', lambda a, kw, o: 'stem_pool_kernel (7x7/s2 conv + BN + ReLU + maxpool, fp32 NCHW in) @ %s' % (tuple(o.shape),),
             lambda a, kw, o: nbytes(*imgs_of(a)) + nbytes(o))
    hbm_wrap('psm_cosine', lambda a, kw, o: 'psm_cosine_mfma_kernel C=%d D=%d @ %dx%dx%d' % (a[0].shape[3], a[2], a[0].shape[0], a[0].shape[1], a[0].shape[2]),
             lambda a, kw, o: nbytes(a[0], a[1]) + float(o.shape[0] * o.shape[1] * o.shape[2] * a[2] * o.element_size()))
    hbm_wrap('cost_volume_fused', lambda a, kw, o: 'cost_volume_fused_kernel F=%d D=%d @ %dx%dx%d' % (a[0].shape[3], a[4], a[0].shape[0], a[0].shape[1], a[0].shape[2]),
             lambda a, kw, o: 2.0 * a[0].shape[0] * a[0].shape[1] * a[0].shape[2] * a[0].shape[3] * a[0].element_size()
             + float(o.shape[0] * o.shape[1] * o.shape[2] * a[0].shape[3] * a[4] * o.element_size()))
    hbm_wrap('dwconv3x3', lambda a, kw, o: 'dwconv3x3_kernel C=%d @ %dx%dx%d' % (a[0].shape[3], a[0].shape[0], a[0].shape[1], a[0].shape[2]),
             lambda a, kw, o: 2.0 * a[0].shape[0] * a[0].shape[1] * a[0].shape[2] * a[0].shape[3] * a[0].element_size())
    hbm_wrap('head_postprocess', lambda a, kw, o: 'head_select_kernel + head_nms_kernel, %d anchors x %d frames' % (a[0].shape[1], a[0].shape[0]),
             lambda a, kw, o: nbytes(a[0], a[2]))          # class logits + anchor table once; the regression rows are read for candidates only
    hbm_wrap('dwconv_transpose', lambda a, kw, o: 'dwconvT_phase_kernel f=%d C=%d -> %s' % (a[2], a[0].shape[3], tuple(o.shape)),
             lambda a, kw, o: nbytes(a[0], o) + (nbytes(kw.get('add')) if kw.get('add') is not None else (nbytes(a[3]) if len(a) > 3 else 0.0)))
    hbm_wrap('image_conv', lambda a, kw, o: 'image_conv7_kernel (7x7/s1 on the fp32 image) -> %s' % (tuple(o.shape),), lambda a, kw, o: nbytes(a[0], o))
    ops.conv2d, ops.deform_conv_general, ops.deform_columns, ops.km3d_head_fused, ops.conv2d_pair = conv2d, dcn, cols, head, pair
    ops.conv2d_bottleneck = bottleneck
    switches = []
    for mod, attr in ((getattr(model, 'bbox_head', None), 'overlap_towers'), (getattr(model, 'core', None), 'overlap_neck')):
        if mod is not None and hasattr(mod, attr):
            switches.append((mod, attr, getattr(mod, attr)))
            setattr(mod, attr, False)           # serial launches: per-kernel event times must not overlap
    try:
        with torch.no_grad():
            for _ in range(reps):
                # a ~1 ms spin kernel ahead of every pass: the host gets ahead of the device, so that an event pair brackets the launch's DEVICE time
                # and not the host's dispatch latency into an idle queue (the first launches of a pass read 2x too long otherwise)
                torch.cuda._sleep(2_000_000)
                model.forward_device(*inputs)
                first_pass_len[0] = 0                   # DCN sampling statistics: the first pass only (they sync the device)
        torch.cuda.synchronize()
    finally:
        ops.conv2d, ops.deform_conv_general, ops.deform_columns, ops.km3d_head_fused = orig['conv2d'], orig['dcn'], orig['cols'], orig['head']
        ops.conv2d_pair = orig['pair']
        ops.conv2d_bottleneck = orig['bottleneck']
        for name in ('stem_conv_pool', 'psm_cosine', 'cost_volume_fused', 'dwconv3x3', 'head_postprocess', 'dwconv_transpose', 'image_conv'):
            setattr(ops, name, orig[name])
        for mod, attr, v in switches:
            setattr(mod, attr, v)
    # No event-overhead correction: per-launch HIP events were compared with the kernel durations of a rocprofv3 --kernel-trace
    # of this very command (`bench.py --no-overlap`, tools/profile_round.sh -> profiles/*_serial_roofline_check.txt): the raw
    # event sums read the family 0.1-4 % LONGER than the profiler (an empty event pair reads 5-6 us, of which 0-3 us end up inside
    # a reading; subtracting a calibrated overhead over-corrected by 3.5-8 %), so the line stays on the conservative side.
    # One reading per launch: the MEDIAN of its `reps` passes (the passes launch the same sequence, so launch i of pass r is record i + r * per).
    # A mean lets ONE stalled pass -- seen under rocprofv3: the host falls behind, an event pair then brackets dispatch latency -- move the whole
    # family by 20 % (profiles/r05: a serial profiled run read 836 instead of ~1 040 TF/s); the three timed regions of the line use the median too.
    def med(vals):
        return sorted(vals)[len(vals) // 2]
    per = len(records) // reps
    launch_s = [med([max(records[i + r * per][2].elapsed_time(records[i + r * per][3]), 0.0) for r in range(reps)]) * 1e-3 for i in range(per)]
    if os.environ.get(verbose_env):
        for i in range(per):
            fl, t = records[i][1], launch_s[i]
            print('  %-16s %2d  %-40s %8.2f GF  %8.1f us  %7.1f TF/s' % (records[i][0], i, records[i][5], fl / 1e9, t * 1e6, fl / t / 1e12), file=sys.stderr)
    fam = {}
    for i in range(per):
        r = records[i]
        d = fam.setdefault(r[0], dict(flops=0.0, secs=0.0, launches=0, bytes=0.0))
        d['flops'] += r[1]
        d['secs'] += launch_s[i]
        d['launches'] += 1.0
        d['bytes'] += r[4]
    # the single most expensive layer shape over all families (its launches all run the same kernel)
    by_shape = {}
    for i in range(per):
        r = records[i]
        d = by_shape.setdefault((r[0], r[5]), [0.0, 0.0, 0])
        d[0] += r[1]
        d[1] += launch_s[i]
        d[2] += 1
    total_secs = sum(d['secs'] for d in fam.values())
    dominant = {}
    for family in fam:
        (f_, name), (fl, tt, n) = max(((k, v) for k, v in by_shape.items() if k[0] == family), key=lambda kv: kv[1][1])
        dominant[family] = dict(layer=name, launches_per_step=n, share_of_family_time=round(tt / fam[family]['secs'], 4),
                                avg_launch_us=round(tt / n * 1e6, 1), achieved=round(fl / tt / 1e12, 1) if tt > 0 else 0.0)
    # HBM-bound kernels: per distinct launch shape, algorithmic bytes / duration (median over the passes per launch) against the HBM roof
    hbm = {}
    hper = len(hbm_records) // reps
    for i in range(hper):
        label, nb = hbm_records[i][0], hbm_records[i][1]
        d = hbm.setdefault(label, [0.0, 0.0, 0])
        d[0] += nb
        d[1] += med([max(hbm_records[i + r * hper][2].elapsed_time(hbm_records[i + r * hper][3]), 0.0) for r in range(reps)]) * 1e-3
        d[2] += 1
    for i in range(per):                   # the DCN launches with few channels are gather / HBM-side kernels too (64 -> 64 at full resolution)
        r = records[i]
        if r[0] == 'dcn' and r[4] and ' 64->  64 ' in r[5]:
            d = hbm.setdefault('dcn_nhwc_kernel ' + r[5], [0.0, 0.0, 0])
            d[0] += r[4]
            d[1] += launch_s[i]
            d[2] += 1
    hbm_kernels = [dict(kernel=k, launches_per_step=v[2], algorithmic_bytes_per_launch=int(v[0] / v[2]), avg_launch_us=round(v[1] / v[2] * 1e6, 1),
                        achieved_gbps=round(v[0] / v[1] / 1e9, 1) if v[1] > 0 else None,
                        frac_of_peak=round(v[0] / v[1] / 1e9 / PEAK_HBM_GBPS, 4) if v[1] > 0 else None,
                        frac_of_achievable=round(v[0] / v[1] / 1e9 / ACHIEVABLE_HBM_GBPS, 4) if v[1] > 0 else None)
                   for k, v in sorted(hbm.items(), key=lambda kv: -kv[1][1])]
    profile_ops.hbm_kernels = hbm_kernels       # (function attribute: the three-value return is used by tools/)
    profile_ops.dcn_sampling = (dict(dcn_offset_rms_px=round((dcn_stats[0] / dcn_stats[1]) ** 0.5, 3), dcn_offset_rms_px_per_layer=dcn_layer_rms,
                                     dcn_corners_in_image_frac=round(dcn_stats[2] / dcn_stats[3], 4),
                                     dcn_corners_in_image_frac_at_zero_offsets=round(dcn_stats[4] / dcn_stats[5], 4))
                                if dcn_stats[1] else None)
    return fam, dominant, total_secs


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(cfg, sd, args):
    """The reference's CPU path timed on this box's host cores, on a bounded sample (a reported baseline, not the target).
    kind "reference": the reference's own Stereo3D.test_forward, imported through oracle/ref_shim.py -- only possible where
    /root/reference exists (the build container; never on the GPU box).  kind "port": the oracle (oracle/detector_oracle.py,
    the builder's CPU restatement of that path, pinned against the reference by tests/golden) -- what the GPU box can time."""
    from oracle import detector_oracle as orc
    from visualdet3d_amd.utils import synthetic as syn
    torch.set_num_threads(min(32, os.cpu_count() or 1))   # more threads than this is slower on the 256-thread host
    L, R = syn.stereo_pair(1, args.height, args.width, seed=0)
    P2, P3 = syn.kitti_calib(args.width, batch=1)
    sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
    kind, what, run = 'port', 'oracle/detector_oracle.py (CPU restatement of the reference path)', None
    if os.path.isdir('/root/reference/visualDet3D') and not os.environ.get('VD3D_BENCH_CPU_PORT'):
        try:
            from oracle import ref_shim          # NB: patches Tensor.cuda() to a no-op -- this leg runs after all GPU work
            ref = ref_shim.detector_dict()['Stereo3D'](cfg)
            ref.load_state_dict(sd_cpu)
            ref.eval()
            run = lambda: ref([L, R, P2, P3])                              # noqa: E731
            kind, what = 'reference', "the reference's own Stereo3D.test_forward (/root/reference via oracle/ref_shim.py)"
        except Exception as e:                                              # shim not importable here: fall back to the port
            print('[bench] reference not importable (%s); timing the port' % e, file=sys.stderr)
    if run is None:
        run = lambda: orc.stereo3d_forward(sd_cpu, cfg, L, R, P2)          # noqa: E731
    with torch.no_grad():
        run()  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            run()
            n += 1
            el = time.perf_counter() - t0
            if el > args.cpu_seconds or n >= 20:
                break
    return dict(value=n / el, unit='img/s', cores=torch.get_num_threads(), kind=kind, cpu_model=_cpu_model(), nproc=os.cpu_count(),
                sample='%d fp32 batch-1 %dx%d stereo pairs through %s in %.1f s' % (n, args.height, args.width, what, el))


KDET = 128                                   # detections per frame that travel (device -> host, rank -> ranks)

# BASELINE.json configs 3 and 5 AS STATED, timed after the headline with the same rules (inputs resident in HBM, hipGraph replay,
# results packed + copied to the host + checked inside the timed region).  gf = conv / GEMM / DCN 2*MAC per unit.
OTHER_CONFIGS = [
    # `offset_scale`, `offset_target_rms`: the DCN offset convs (`conv_offset`, zero-initialised by the reference, lib/ops/dcn/deform_conv.py:453-457) are
    # seeded randomly, scaled, and then brought to 1 px rms PER LAYER on the timed input (prepare_other_config) -- the regime of a trained network (smooth,
    # small offsets).  Unscaled they are 8 px (C3) / up to 49 px (C5's first up-path layer) rms with many bilinear corners OUTSIDE the image, which cost
    # no traffic: the r5 workload, kept as `legacy_workload` (timed once beside the new one).
    # `head_bias` / `hm_bias`: the class-logit bias that sets the detection count to a KITTI-like 5-50 per frame (r5: 222 / exactly K = 100 per frame);
    # calibrated with tools/calibrate_bench_workloads.py on the GPU, the counts are reported in every entry.
    dict(key='C3', workload='YOLOStereo3D ResNet-50 + DCNv2 (base) head, 288x1280 stereo pairs, batch=32', kind='stereo', depth=50,
         H=288, W=1280, B=32, gf=472.0, dcn_head=True, dtype='bf16', wseed=6, head_std=0.006, score_thr=0.5,   # weights / threshold of the golden case stereo3d_r50_dcn_288x1280
         offset_scale=1.0 / 8, offset_target_rms=1.0, head_bias=-1.4),
    dict(key='C5', workload='KM3D_example (DLA-34 CenterNet mono3D, keypoint head), 512x1760 images, batch=16', kind='km3d',
         H=512, W=1760, B=16, gf=326.18, dtype='fp16', offset_scale=1.0 / 32, offset_target_rms=1.0, hm_bias=-2.19, hm_hp_bias=-3.5, head_gain=0.086),
]


def other_config_state_dict(c, model, legacy=False):
    """The seeded weights of one OTHER_CONFIGS entry.  legacy = the round-5 workload (unscaled random DCN offsets, uncalibrated head biases)."""
    from visualdet3d_amd.utils import synthetic as syn
    sd = syn.seeded_state_dict(model.state_dict(), seed=c.get('wseed', 1), head_std=c.get('head_std', 0.0005),
                               head_bias=-1.0 if legacy else c.get('head_bias', -1.0))
    if legacy:
        return sd
    for k in sd:
        if '.conv_offset.' in k:
            sd[k] = sd[k] * c.get('offset_scale', 1.0)
        if c['kind'] == 'km3d' and k.endswith('head_layers.hm.2.bias'):
            sd[k] = torch.full_like(sd[k], c.get('hm_bias', -2.19))
        if c['kind'] == 'km3d' and k.endswith('head_layers.hm_hp.2.bias'):
            sd[k] = torch.full_like(sd[k], c.get('hm_hp_bias', -2.19))
        if c['kind'] == 'km3d' and k.startswith('bbox_head.head_layers.') and k.endswith('.2.weight'):
            sd[k] = sd[k] * c.get('head_gain', 1.0)
    return sd


def dcn_layer_offsets(model, inputs):
    """One eager forward; -> [(ModulatedDeformConvPack module, rms of the offsets it sampled with, px)] in call order."""
    from visualdet3d_amd import hip_ops as ops
    from visualdet3d_amd.networks.lib.ops.dcn.deform_conv import ModulatedDeformConvPack as M
    mods, rms = [], []
    orig_fwd, orig_dcn, orig_cols = M.forward_nhwc, ops.deform_conv_general, ops.deform_columns

    def fwd(self, *a, **kw):
        mods.append(self)
        return orig_fwd(self, *a, **kw)

    def dcn(x, pd, offset, *a, **kw):
        rms.append(float(offset.float().pow(2).mean().sqrt()))
        return orig_dcn(x, pd, offset, *a, **kw)

    def cols(x, offset, *a, **kw):
        rms.append(float(offset.float().pow(2).mean().sqrt()))
        return orig_cols(x, offset, *a, **kw)

    M.forward_nhwc, ops.deform_conv_general, ops.deform_columns = fwd, dcn, cols
    try:
        with torch.no_grad():
            model.forward_device(*inputs)
        torch.cuda.synchronize()
    finally:
        M.forward_nhwc, ops.deform_conv_general, ops.deform_columns = orig_fwd, orig_dcn, orig_cols
    assert len(mods) == len(rms), (len(mods), len(rms))
    return list(zip(mods, rms))


def prepare_other_config(c, model, inputs):
    """Load the entry's weights into `model` (on the device) and bring EVERY DCN layer's offsets to `offset_target_rms` pixels rms: the seeded offset convs
    give 0.4 ... 49 px depending on the layer (fan-in 64 ... 512 inputs of very different scale), a trained network's are small and smooth everywhere.
    Two rounds of `measure each layer on this very input, rescale its conv_offset weight and bias` (layers feed each other; the second round lands
    within a few per cent).  Workload construction, outside every timed region; deterministic (seeded weights, seeded input)."""
    model.load_state_dict(other_config_state_dict(c, model))
    target = c.get('offset_target_rms')
    if target:
        for _ in range(2):
            for mod, rms in dcn_layer_offsets(model, inputs):
                if rms > 0:
                    with torch.no_grad():
                        # (offsets AND mask logits scale: the conv's 27 outputs are one tensor; a mask nearer sigmoid(0) = 0.5 is the reference's initial state)
                        mod.conv_offset.weight.mul_(target / rms)
                        mod.conv_offset.bias.mul_(target / rms)


def pin_to_gpu_numa_node(local_rank, n_local=None):
    """Bind this rank's host threads to ITS share of the CPUs of the NUMA node its GPU hangs off (visualdet3d_amd.distributed.rank_cpu_sets: the
    node's CPUs divided among the local ranks whose GPUs share the node; matters for --feed host at 8 ranks).  Best effort: returns a description,
    never raises."""
    try:
        from visualdet3d_amd.distributed import rank_cpu_sets
        n_local = n_local or int(os.environ.get('LOCAL_WORLD_SIZE', '0')) or torch.cuda.device_count()
        bdfs = []
        for i in range(n_local):
            props = torch.cuda.get_device_properties(i)
            bdfs.append('%04x:%02x:%02x.0' % (getattr(props, 'pci_domain_id', 0), props.pci_bus_id, props.pci_device_id))
        cpus = rank_cpu_sets(bdfs, allowed=os.sched_getaffinity(0))[local_rank]
        if not cpus:
            return 'gpu %s: no NUMA node reported' % bdfs[local_rank]
        os.sched_setaffinity(0, cpus)
        return 'gpu %s -> %d cpus of its NUMA node (%d local ranks)' % (bdfs[local_rank], len(cpus), n_local)
    except Exception as e:      # noqa: BLE001  (sysfs layout / permissions differ between hosts)
        return 'not pinned (%s)' % e


class Stepper(CapturedStep):
    """One configuration's timed step = `visualdet3d_amd.networks.pipelines.in_flight.CapturedStep` (two warm-up passes, hipGraph capture of `forward_device` +
    pack; per step: replay, copy the packed [B, KDET + 1, 13] record to a pinned slot, ONE host sync -- taken after the next step is enqueued -- and a check of
    the counts) with this file's record size and A/B switches: VD3D_BENCH_SYNC_EACH_STEP=1 = the round 1 - 5 loop (sync and check before the next replay is
    launched), VD3D_BENCH_NOSIDE=1 = no pass off the default stream before the capture."""

    def __init__(self, model, inputs, B, device, use_graph=True, pre=None):
        super().__init__(model, inputs, B, device, k=KDET, use_graph=use_graph, pre=pre, side_pass=not os.environ.get('VD3D_BENCH_NOSIDE'),
                         sync_each_step=bool(os.environ.get('VD3D_BENCH_SYNC_EACH_STEP')))


class HostFeed:
    """--feed host: 2 x B uint8 camera frames per step travel from pinned host memory into a double-buffered device ring on a
    copy stream; the step itself (captured in the hipGraph) starts with vd3d_preprocess_image on a STATIC device frame buffer
    (crop / cv2-style resize / pad / normalise -> the fp32 NCHW network input), which the main stream refreshes from ring slot
    (i & 1) right before the replay.  Upload i + 1 overlaps step i."""
    def __init__(self, B, H, W, device, L, R, seed=0, replicas=1):
        from visualdet3d_amd import hip_ops
        # The frames are the resident workload's own images as uint8 camera frames of the NETWORK's size (de-normalised, rounded to bytes): the
        # preprocessed input then equals the resident one up to the byte rounding (0.02 sigma) and the timed decode / NMS see the same ~100
        # detections.  (Until round 5: uniform random bytes at KITTI's 375 x 1242 -- NO detections; the resident images resampled to 375 x 1242 and
        # back lose their high frequencies: 3 detections.  384 x 1280 frames are 5 % more bytes per upload than a KITTI frame; the resize still
        # runs, at scale 1.)  The second ring slot holds the same frames in another batch order.
        self.HS, self.WS = H, W
        self.host = [f.pin_memory() for f in self.slot_frames(L, R)]
        self.ring = [torch.empty((2 * B, self.HS, self.WS, 3), dtype=torch.uint8, device=device) for _ in range(2)]
        # the graphs' static sources: one per replica of the detector (two steps in flight: step i's frames are copied in while step i - 1 still reads its own)
        self.frames_all = [torch.empty((2 * B, self.HS, self.WS, 3), dtype=torch.uint8, device=device) for _ in range(max(1, replicas))]
        self.frames = self.frames_all[0]
        self.copy_stream = torch.cuda.Stream()
        self.uploaded = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        self.B, self.size, self.L, self.R = B, (H, W), L, R
        self.mean, self.std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
        self.bytes_per_step = self.host[0].numel()
        self.hip_ops = hip_ops
        for f in self.frames_all:
            f.copy_(self.host[0])

    @staticmethod
    def slot_frames(L, R):
        """the two ring slots' uint8 frames [2 B, H, W, 3] (left frames, then right frames) of the normalised images L, R [B, 3, H, W]"""
        mean_t = torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)
        std_t = torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)

        def frames_of(x):
            img = x.detach().float().cpu() * std_t + mean_t
            return (img.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        return [torch.cat([frames_of(L), frames_of(R)], 0), torch.cat([frames_of(L).roll(1, 0), frames_of(R).roll(1, 0)], 0)]

    preprocess = None               # set by main(): the 2 x B vd3d_preprocess_image launches that open the captured step

    def upload(self, i):
        """enqueue the H2D copy of step i's frames into ring slot (i & 1) on the copy stream"""
        s = i & 1
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.consumed[s])          # slot free: the step that used it has copied it out
            self.ring[s].copy_(self.host[s], non_blocking=True)
            self.uploaded[s].record(self.copy_stream)

    def before_step(self, i, r=0):
        """current stream (replica r's), right before step i's replay: wait for step i's frames, refresh that replica's static source from ring slot
        (i & 1), then start step i + 1's upload so that it travels while step i computes."""
        main = torch.cuda.current_stream()
        s = i & 1
        if i == 0:
            self.upload(0)                                          # first step of a run: nothing was prefetched for it
        main.wait_event(self.uploaded[s])
        self.frames_all[r].copy_(self.ring[s], non_blocking=True)   # D2D, 22 MB at HBM speed
        self.consumed[s].record(main)
        self.upload(i + 1)


def time_other_config(c, device, steps, warmup, in_flight=2):
    """-> the `other_configs` entry of one BASELINE configuration, or {'error': ...}.  in_flight = 2: timed like the headline, two replicas of the detector
    replayed alternately on two streams (`one_in_flight`: the same steps on one replica, one region)."""
    from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT
    import visualdet3d_amd.networks.detectors  # noqa: F401
    from visualdet3d_amd.utils import synthetic as syn
    tmp = tempfile.mkdtemp()

    def make():
        if c['kind'] == 'stereo':
            cfg = syn.stereo3d_cfg(tmp, depth=c['depth'], score_thr=c.get('score_thr', 0.75), nms_iou_thr=0.4)
            syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
            if c.get('dcn_head'):
                from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3DBaseHead
                mm = Stereo3DBaseHead(cfg)
            else:
                mm = DETECTOR_DICT[cfg.name](cfg)
        else:
            cfg = syn.km3d_cfg(output_w=c['W'] // 4)
            mm = DETECTOR_DICT[cfg.name](cfg)
        mm = mm.to(device).eval()
        mm.compute_dtype = torch.float16 if c['dtype'] == 'fp16' else torch.bfloat16
        return mm
    m = make()

    def forks(mm, on):
        if hasattr(getattr(mm, 'core', None), 'overlap_neck'):
            mm.core.overlap_neck = on
        if hasattr(getattr(mm, 'bbox_head', None), 'overlap_towers'):
            mm.bbox_head.overlap_towers = on
    B, H, W = c['B'], c['H'], c['W']
    P2, _ = syn.kitti_calib(W, batch=B)
    if c['kind'] == 'stereo':
        L, R = syn.stereo_pair(B, H, W, seed=3)
        inputs = (L.to(device), R.to(device), P2.to(device))
    else:
        inputs = (syn.mono_image(B, H, W, seed=3).to(device), P2.to(device))
    # the round-5 workload once, for the record (random 8 / 49 px DCN offsets, 222 / 100 detections per frame): one short timed region
    legacy = None
    if not os.environ.get('VD3D_BENCH_NO_LEGACY'):
        m.load_state_dict(other_config_state_dict(c, m, legacy=True))
        st = Stepper(m, inputs, B, device)
        n_leg = max(3, steps // 2)
        el, _, counts = st.timed(n_leg, 2, regions=1)
        legacy = dict(what='round-5 workload: unscaled random conv_offset weights, uncalibrated head bias', steps=n_leg,
                      ms_per_step=round(el / n_leg * 1e3, 3), value=round(B * n_leg / el, 2), detections_per_frame_mean=round(float(counts.sum()) / B, 1))
        del st
        torch.cuda.empty_cache()
    prepare_other_config(c, m, inputs)
    st = Stepper(m, inputs, B, device)           # (intra-step side streams on: the one-in-flight loop)
    one = None
    if in_flight == 2:
        side = bool(os.environ.get('VD3D_BENCH_SIDE_STREAMS'))
        ms_ = [make() for _ in range(int(os.environ.get('VD3D_BENCH_REPLICAS', '2')))]      # (more than two: measured no better, tools/two_in_flight.py)
        for mm in ms_:
            mm.load_state_dict(m.state_dict())       # the calibrated workload (weights are the only thing the replicas have in common)
            forks(mm, side)                          # two in flight: every step a one-stream graph (see main())
        reps = [(Stepper(mm, inputs, B, device), torch.cuda.Stream()) for mm in ms_]
        torch.cuda.synchronize()
        run_in_flight(reps, warmup)
        all_s = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            counts = run_in_flight(reps, steps)
            torch.cuda.synchronize()
            all_s.append(time.perf_counter() - t0)
        elapsed = sorted(all_s)[1]
        el1, _, _ = st.timed(steps, 2, regions=1)
        one = dict(ms_per_step=round(el1 / steps * 1e3, 3), value=round(B * steps / el1, 2))
        del reps, ms_
        torch.cuda.empty_cache()
    else:
        elapsed, all_s, counts = st.timed(steps, warmup)
    fam, dominant, _ = profile_ops(m, inputs, reps=3, verbose_env='VD3D_BENCH_LAYERS_OTHER')
    value = B * steps / elapsed
    # the single (family, layer shape) that costs the step most
    dom_family = max(dominant, key=lambda k: dominant[k]['avg_launch_us'] * dominant[k]['launches_per_step'])
    per_frame = [int(v) for v in counts.reshape(-1).tolist()]
    entry = dict(config=c['key'], workload=c['workload'], dtype=c['dtype'], steps=steps, warmup=warmup,
                 ms_per_step=round(elapsed / steps * 1e3, 3), value=round(value, 2), unit='img/s',
                 spread=dict(timed_regions=len(all_s), ms_per_step=[round(t / steps * 1e3, 3) for t in all_s],
                             value_min=round(B * steps / max(all_s), 2), value_max=round(B * steps / min(all_s), 2)),
                 whole_path_frac=round(value * c['gf'] / 1e3 / PEAK_BF16_TFLOPS, 4), gflop_per_unit=c['gf'],
                 detections_last_step=int(counts.sum()), detections_per_frame=per_frame,
                 dcn_sampling=dict(profile_ops.dcn_sampling or {}, conv_offset_scale=c.get('offset_scale', 1.0)),
                 legacy_workload=legacy, in_flight=in_flight, one_in_flight=one,
                 families={k: dict(launches=int(round(v['launches'])), ms=round(v['secs'] * 1e3, 3),
                                   achieved_tflops=round(v['flops'] / v['secs'] / 1e12, 1) if v['flops'] else None,
                                   frac=round(v['flops'] / v['secs'] / 1e12 / PEAK_BF16_TFLOPS, 4) if v['flops'] else None)
                           for k, v in fam.items()},
                 dominant_kernel=dict(dominant[dom_family], family=dom_family, unit='TFLOP/s'),
                 frac=round(dominant[dom_family]['achieved'] / PEAK_BF16_TFLOPS, 4), hbm_kernels=profile_ops.hbm_kernels)
    if entry['detections_last_step'] < 1:
        entry['error'] = 'the timed workload produced no detections: decode / NMS ran on empty candidate lists'
    del st, m, inputs
    torch.cuda.empty_cache()
    return entry


# The reference's own call pattern: ONE frame per `module([...])` call (networks/pipelines/testers.py:15-42; the detectors assert
# batch 1, yolostereo3d_detector.py:78, yolomono3d_detector.py:111), wall clock around the call INCLUDING the host's read of the detection
# count -- what a drop-in user of scripts/eval.py gets per frame.  gf = conv / GEMM 2*MAC per frame.
API_CONFIGS = [
    dict(key='C1', kind='mono', name='GroundAwareYolo3D', gf=82.10,
         workload='Yolo3D_example (GroundAware Mono3D, ResNet-34) single 384x1280 image, batch=1, through module([image, P2])'),
    dict(key='C2_B1_api', kind='stereo', name='Stereo3D', gf=GFLOP_PER_PAIR,
         workload='Stereo3D_example (YOLOStereo3D, ResNet-34) one 384x1280 stereo pair, batch=1, through module([left, right, P2, P3])'),
]


def time_api_config(c, device, calls=100, warm=10):
    from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT
    import visualdet3d_amd.networks.detectors  # noqa: F401
    from visualdet3d_amd.utils import synthetic as syn
    tmp = tempfile.mkdtemp()
    if c['kind'] == 'mono':
        cfg = syn.mono3d_cfg(tmp, depth=34, score_thr=0.75, name=c['name'])
        syn.write_synthetic_priors(tmp, cfg.obj_types, 2)
    else:
        cfg = syn.stereo3d_cfg(tmp, depth=34, score_thr=0.75, nms_iou_thr=0.4)
        syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    m = DETECTOR_DICT[cfg.name](cfg)
    # mono: head scale 0.015 -> 31 candidates / 26 detections on this frame in fp32 (the CPU oracle), a KITTI-like count.  With round 4's 0.0005 no
    # anchor passed score_thr 0.75 and the timed decode / NMS ran on an empty candidate list; the golden cases' 0.02 gives 151.
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), seed=1, head_std=0.00042 if c['kind'] == 'stereo' else c.get('head_std', 0.015)))
    m = m.to(device).eval()
    m.compute_dtype = torch.bfloat16
    P2, P3 = syn.kitti_calib(1280, batch=1)
    if c['kind'] == 'stereo':
        L, R = syn.stereo_pair(1, 384, 1280, seed=3)
        x = [L.to(device), R.to(device), P2.to(device), P3.to(device)]
    else:
        x = [syn.mono_image(1, 384, 1280, seed=3).to(device), P2.to(device)]
    with torch.no_grad():
        for _ in range(warm):                    # first call: eager passes + hipGraph capture (lib/graphed.py)
            out = m(x)
        per = []
        for _ in range(3):                       # three regions of `calls` calls: the median is the entry, all three are its spread
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(calls):
                out = m(x)                       # returns after the host has read the detection count
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / calls)
        dt = sorted(per)[1]
        stats = m.graph_stats
        m.use_graph = False                      # the same kernels launched eagerly, for the record
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m(x)
        torch.cuda.synchronize()
        dt_eager = (time.perf_counter() - t0) / 20
    entry = dict(config=c['key'], workload=c['workload'], dtype='bf16', calls=calls, ms_per_call=round(dt * 1e3, 4), value=round(1.0 / dt, 1), unit='img/s',
                 spread=dict(timed_regions=3, ms_per_call=[round(t * 1e3, 4) for t in per]),
                 timing='wall clock around module([...]) incl. input copies into the graph, replay, result sync and result clones',
                 whole_path_frac=round(c['gf'] / dt / 1e3 / PEAK_BF16_TFLOPS, 4), gflop_per_unit=c['gf'], hip_graph=stats,
                 ms_per_call_eager_launches=round(dt_eager * 1e3, 4), detections=int(out[0].numel()))
    if entry['detections'] < 1:
        entry['error'] = 'the timed workload produced no detections: decode / NMS ran on an empty candidate list'
    del m, x
    torch.cuda.empty_cache()
    return entry


def run_in_flight(reps, n, before=None):
    """n steps over the replicas `reps` = [(Stepper, stream), ...], step i on replica i % len(reps): `InFlight.run` -- replay, copy the packed record to the replica's
    pinned slot, record; the host waits for (and checks) step i - 1's record after step i is enqueued.  -> the counts of the last step."""
    return InFlight([st for st, _ in reps], [s for _, s in reps]).run(n, before=before)


def main():
    args = parse()
    try:
        how, what = resolve_world(args.gpus, os.environ, [os.path.abspath(__file__)] + sys.argv[1:])
    except ValueError as e:
        sys.exit('[bench] %s' % e)
    if how == 'exec':                               # `python bench.py --gpus 8` with no launcher: become the 8-rank job (rank 0 prints the line)
        print('[bench] --gpus %d without a launcher: %s' % (args.gpus, ' '.join(what)), file=sys.stderr, flush=True)
        os.execvpe(what[0], what, dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')))
    # stdout carries the ONE JSON line and nothing else: keep a private handle on the real stdout and point fd 1 at stderr, so that whatever else
    # writes to stdout -- RCCL prints its version banner through C stdio, flushed at exit, i.e. AFTER a Python print -- lands on stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    world = what
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = world > 1 or bool(os.environ.get('VD3D_BENCH_FORCE_DIST'))   # the env: exercise the RCCL path on one GPU
    if args.cpu_baseline_only:
        _, cfg, sd = build_model(args, torch.device('cpu'))
        os.environ['VD3D_BENCH_CPU_PORT'] = '1'
        port = cpu_baseline(cfg, sd, args)
        del os.environ['VD3D_BENCH_CPU_PORT']
        ref = cpu_baseline(cfg, sd, args)
        json_out.write(json.dumps({'cpu_baseline_port': port, 'cpu_baseline': ref}) + '\n')
        json_out.flush()
        return
    assert torch.cuda.is_available(), 'bench.py measures the MI355X HIP path; no GPU visible'
    torch.cuda.set_device(local_rank)              # before the process group: RCCL binds to the current device
    device = torch.device('cuda', local_rank)
    numa = pin_to_gpu_numa_node(local_rank) if (world > 1 or args.feed == 'host') else 'not pinned (single rank, resident inputs)'
    if dist:
        import torch.distributed as td
        td.init_process_group(backend='nccl', init_method='env://')

    from visualdet3d_amd.utils import synthetic as syn
    model, cfg, sd = build_model(args, device)
    B = args.batch
    L, R = syn.stereo_pair(B, args.height, args.width, seed=int(os.environ.get('VD3D_BENCH_SEED', '100')) + rank)
    P2, P3 = syn.kitti_calib(args.width, batch=B)
    L, R, P2 = L.to(device), R.to(device), P2.to(device)   # inputs resident in HBM before the timed region
    inputs = (L, R, P2)

    # Intra-step side streams (neck s4 / s8 under backbone layer2 / layer3, cls tower next to the reg tower) fill a lone step's gaps; with TWO steps in flight the
    # other step does that, and the forks only cost (same box, two in flight: 3.467 / 3.453 ms with them, 3.364 / 3.393 / 3.396 without the neck fork, 3.352 / 3.355 /
    # 3.385 / 3.391 without any; one in flight: 3.66 with, 3.74 - 3.76 without).  So: two in flight = every step a ONE-stream graph (VD3D_BENCH_SIDE_STREAMS=1
    # keeps the forks, for the A/B); one in flight = forks on (--no-overlap: off).
    n_fl_wanted = 1 if (args.no_graph or args.no_overlap) else args.in_flight
    no_forks = args.no_overlap or (n_fl_wanted == 2 and not os.environ.get('VD3D_BENCH_SIDE_STREAMS'))

    def apply_knobs(m, no_forks=no_forks):
        if os.environ.get('VD3D_BENCH_NONECK') or no_forks:
            m.core.overlap_neck = False
        if os.environ.get('VD3D_BENCH_NOTOWER') or no_forks:
            m.bbox_head.overlap_towers = False
        if os.environ.get('VD3D_BENCH_NOSELECT'):
            m.bbox_head.overlap_select = False      # A/B: candidate selection after the towers instead of on the cls tower's stream
    apply_knobs(model)
    from visualdet3d_amd import hip_ops

    feed = None
    inputs_r = [inputs, inputs]                 # per replica: resident feed -> the same (read-only) tensors
    pre_r = [None, None]
    if args.feed == 'host':
        feed = HostFeed(B, args.height, args.width, device, L, R, seed=rank, replicas=n_fl_wanted)
        mean_c = (C_float3)(*feed.mean)
        std_c = (C_float3)(*feed.std)
        Hr, Wr, _ = hip_ops.resized_shape(feed.HS, feed.WS, 0, feed.size)
        inputs_r = [inputs, (L.clone(), R.clone(), P2)]      # the preprocessing WRITES the network inputs: each replica its own

        def make_preprocess(r):
            Lr, Rr = inputs_r[r][0], inputs_r[r][1]
            src = feed.frames_all[min(r, len(feed.frames_all) - 1)]

            def preprocess():       # 2 x B launches inside the captured step: frame b -> L[b] / R[b] (fp32 NCHW, the reference's input)
                from visualdet3d_amd import _lib
                for b in range(2 * B):
                    dst = Lr[b] if b < B else Rr[b - B]
                    _lib.check(_lib.lib().vd3d_preprocess_image(src[b].data_ptr(), feed.HS, feed.WS, 0, Hr, Wr, dst.data_ptr(), None,
                                                                args.height, args.width, mean_c, std_c, hip_ops._stream()), 'vd3d_preprocess_image')
            return preprocess
        pre_r = [make_preprocess(0), make_preprocess(1)]
        feed.preprocess = pre_r[0]

    stepper = Stepper(model, inputs, B, device, use_graph=not args.no_graph, pre=feed.preprocess if feed else None)
    graph, pack_static = stepper.graph, stepper.pack_static
    # Steps in flight: replica r = (its Stepper, the stream its steps are enqueued on).  One replica: the current stream, steps back to back.  Two: a second
    # detector object with the same weights (nothing is shared between the two but the read-only inputs), each on its own stream; step i runs on replica i & 1.
    n_fl = 1 if (args.no_graph or graph is None or args.no_overlap) else args.in_flight      # (--no-overlap: ONE stream, kernel durations additive)
    reps = [(stepper, torch.cuda.current_stream())]
    if n_fl == 2:
        try:
            model2 = build_model(args, device)[0]
            apply_knobs(model2)
            reps = [(stepper, torch.cuda.Stream()), (Stepper(model2, inputs_r[1], B, device, pre=pre_r[1]), torch.cuda.Stream())]
            torch.cuda.synchronize()
        except torch.cuda.OutOfMemoryError as e:      # (not on a 288 GB part; the line says what ran: config.in_flight)
            print('[bench] second replica does not fit (%s): ONE step in flight' % str(e).splitlines()[0], file=sys.stderr, flush=True)
            n_fl, model2 = 1, None
            torch.cuda.empty_cache()

    # Results leave the device every step, inside the timed region: the packed record of the step (B x (KDET + 1) x 13 floats,
    # ~53 KB) is copied to pinned host memory and its counts are checked on the host.
    # N > 1: the only collective is the gather of that record -- ONE RCCL all_gather per step, latency bound.  It runs on its
    # own stream, double buffered: the comm stream first copies the graph's static record into send buffer (i & 1) (so the
    # next replay may overwrite the record at once), then gathers, then copies the gathered block to the host; the host reads
    # step k's results while step k+1 is already enqueued, so the collective's latency overlaps the next forward.
    gather_ev = []
    if dist:
        import torch.distributed as td
        from visualdet3d_amd.distributed import DetectionGather
        # ring of RD step slots (send buffer, events, pinned block).  One step in flight: RD = 2, the host reads step i - 1 after enqueuing step i (rounds 1 - 5).
        # Two: RD = 4, the host-ordered loop in run() (VD3D_BENCH_COMM_STREAM=1: the GPU-ordered loop with two replicas, for the A/B).
        RD = 2 * n_fl
        lag = 1
        host_ordered = n_fl == 2 and not os.environ.get('VD3D_BENCH_COMM_STREAM')
        gatherers = [DetectionGather(B, KDET, device, world) for _ in range(RD)]
        comm_stream = torch.cuda.Stream()
        packed_ev = [torch.cuda.Event() for _ in range(RD)]
        taken_ev = [torch.cuda.Event() for _ in range(RD)]
        done_ev = [torch.cuda.Event() for _ in range(RD)]
        pinned = [torch.empty((world, B, KDET + 1, 13), dtype=torch.float32).pin_memory() for _ in range(RD)]
    else:
        pinned = [stepper.pinned]

    def run(n):
        """n steps; returns the (host) detection counts [ranks, B] of the last one."""
        if not dist:
            if n_fl == 1:
                return stepper.run(n, before_step=feed.before_step if feed else None)
            return run_in_flight(reps, n, before=feed.before_step if feed else None)
        counts = None

        def collect(i):
            done_ev[i % RD].synchronize()
            return stepper.check(pinned[i % RD])

        if host_ordered:
            # Two steps in flight.  A stream that WAITS on a replica's event for a whole forward (the comm stream of the loop below) stopped the two replays
            # from overlapping (same box, world 1: 3.60 ms against 3.50 with no cross-stream wait at all; the collective on the replica's own stream is worse,
            # 3.70: RCCL runs on its own stream and waits there).  So the HOST orders a step's collective behind its forward: the replica's stream copies the
            # record into send buffer i % 4 and records; one step later the host has seen that event and enqueues gather + D2H on the comm stream, which waits
            # on nothing; the block is read one more step later.  No GPU-side wait longer than the collective itself.
            def tail(j):
                q = j % RD
                packed_ev[q].synchronize()
                with torch.cuda.stream(comm_stream):
                    t_s, t_e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t_s.record(comm_stream)
                    out = gatherers[q].gather()
                    t_e.record(comm_stream)
                    gather_ev.append((t_s, t_e))
                    pinned[q].copy_(out, non_blocking=True)
                    done_ev[q].record(comm_stream)

            for i in range(n):
                q = i % RD
                st, main = reps[i % n_fl]
                with torch.cuda.stream(main):
                    if feed:
                        feed.before_step(i, i % n_fl)
                    st.forward_step()
                    gatherers[q].pack.copy_(st.pack_static, non_blocking=True)
                    packed_ev[q].record(main)
                if i >= 1:
                    tail(i - 1)
                if i >= 2:
                    counts = collect(i - 2)
            if n >= 1:
                tail(n - 1)
            for j in range(max(0, n - 2), n):
                counts = collect(j)
            return counts

        for i in range(n):
            q = i % RD
            g = gatherers[q]
            st, main = reps[i % n_fl]
            with torch.cuda.stream(main):
                if i >= n_fl:
                    main.wait_event(taken_ev[(i - n_fl) % RD])   # this replica's previous record (step i - n_fl) has been copied out of its static buffer
                if feed:
                    feed.before_step(i, i % n_fl)
                st.forward_step()
                packed_ev[q].record(main)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(packed_ev[q])
                g.pack.copy_(st.pack_static, non_blocking=True)
                taken_ev[q].record(comm_stream)
                t_s, t_e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t_s.record(comm_stream)
                out = g.gather()
                t_e.record(comm_stream)
                gather_ev.append((t_s, t_e))
                pinned[q].copy_(out, non_blocking=True)
                done_ev[q].record(comm_stream)
            if i >= lag:
                counts = collect(i - lag)
        for j in range(max(0, n - lag), n):
            counts = collect(j)
        return counts

    dbg = bool(os.environ.get('VD3D_BENCH_DEBUG'))
    if dbg:
        print('[bench] captured=%s, entering warmup' % (graph is not None), file=sys.stderr, flush=True)
    run(args.warmup)
    if dbg:
        print('[bench] warmup done', file=sys.stderr, flush=True)
    del gather_ev[:]
    region_s = []
    for _ in range(max(1, args.regions)):
        # one timed region: EXACTLY args.steps steps between barrier + synchronize on both sides, MAX over ranks
        if dist:
            td.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        counts = run(args.steps)
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        el = time.perf_counter() - t0
        if dist:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            el = float(t.item())
        region_s.append(el)
    elapsed = sorted(region_s)[len(region_s) // 2]          # the median region is the line's value
    if dbg:
        print('[bench] timed regions done %s s' % region_s, file=sys.stderr, flush=True)
    gather_us = sum(s.elapsed_time(e) for s, e in gather_ev) / max(len(gather_ev), 1) * 1e3 if gather_ev else None
    print('[bench] rank %d/%d on cuda:%d: %d regions x %d steps, median %.4f s = %.3f ms/step (all: %s) (%d detections in the last step%s)%s; %s'
          % (rank, world, local_rank, len(region_s), args.steps, elapsed, elapsed / args.steps * 1e3, ' '.join('%.3f' % (t / args.steps * 1e3) for t in region_s),
             int(counts.sum()), ' over all ranks' if dist else '', '; all_gather %.1f us/step on the comm stream' % gather_us if gather_us is not None else '', numa),
          file=sys.stderr, flush=True)
    assert counts is not None and float(counts.min()) >= 0, 'candidate overflow in the head post-processing'
    if os.environ.get('VD3D_BENCH_DUMP'):
        # test hook (tests/test_bench_dist_gpu.py): the last step's host-side results next to forward_device's own
        last = args.steps - 1
        if dist:
            host_block = pinned[last % RD]
        elif n_fl == 1:
            host_block = stepper.pinned_ring[0 if stepper.sync_each_step else last & 1]
        else:
            host_block = reps[last % n_fl][0].pinned_ring[(last // n_fl) & 1]
        last_st = reps[last % n_fl][0]                  # (host feed: the replica whose static inputs hold the last step's frames)
        torch.save(dict(host=host_block.clone(), direct=[t.cpu() for t in last_st.forward_step()],
                        inputs=[last_st.inputs[0].cpu(), last_st.inputs[1].cpu()] if feed else None),
                   os.environ['VD3D_BENCH_DUMP'])

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        fam, dominant_all, _ = profile_ops(model, inputs)
        hbm_kernels = profile_ops.hbm_kernels
        conv, dominant = fam['conv'], dominant_all['conv']
        flops, secs, nl, alg_bytes = conv['flops'], conv['secs'], int(round(conv['launches'])), conv['bytes']
        # roofline.traffic is STATIC (PMC counters need their own rocprofv3 passes): the newest committed record, stamped by
        # tools/pmc_traffic.py with the hash of the convolution sources it measured -- `traffic_stale` says whether those sources moved since
        from visualdet3d_amd import _lib as vlib, build as vbuild
        traffic, traffic_source, traffic_stale = None, None, None
        lib_hash = vlib.lib().vd3d_source_hash().decode()
        library_current = lib_hash == vbuild.source_hash()          # the loaded .so was built from the sources in this tree
        for pmc in sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
            if args.dtype == 'bf16' and B == 8:
                rec = json.load(open(pmc))
                traffic = rec['conv_hbm_bytes_per_forward']
                stamp = rec.get('conv_sources_sha16')
                traffic_stale = (stamp is None) or (stamp != vbuild.source_hash(vbuild.CONV_SOURCES)) or not library_current
                traffic_source = ('profiles/%s (STATIC: read from the committed rocprofv3 --pmc passes of this same command, NOT measured in this run; '
                                  'conv sources then %s, now %s)' % (os.path.basename(pmc), stamp or 'unstamped', vbuild.source_hash(vbuild.CONV_SOURCES)))
                break
        peak = PEAK_BF16_TFLOPS if args.dtype == 'bf16' else PEAK_F32_TFLOPS
        ach = flops / secs / 1e12
        line = {
            'metric': 'images/sec at 384x1280 stereo (YOLOStereo3D ResNet-34 forward incl. decode+NMS)',
            'value': round(value, 2), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'spread': {'timed_regions': len(region_s), 'statistic': 'value / ms_per_step = the MEDIAN of the timed regions (each exactly `steps` steps)',
                       'ms_per_step': [round(t / args.steps * 1e3, 3) for t in region_s],
                       'value_min': round(world * B * args.steps / max(region_s), 2), 'value_max': round(world * B * args.steps / min(region_s), 2)},
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'Stereo3D_example (YOLOStereo3D, ResNet-34) %dx%d stereo pairs, batch=%d per GPU'
                                   % (args.height, args.width, B),
                       'global_batch': world * B, 'parallelism': 'dp%d' % world, 'hip_graph': graph is not None,
                       'side_streams': not no_forks, 'results_d2h_bytes_per_step': int(pack_static.numel() * 4 * world),
                       'in_flight': n_fl,
                       'feed': args.feed if not feed else 'host: %d uint8 bytes uploaded per step and rank (2 x %d frames of %dx%dx3) + vd3d_preprocess_image inside the step'
                               % (feed.bytes_per_step, B, feed.HS, feed.WS)},
            'roofline': {'bound': 'mfma', 'kernel': 'vd3d_conv2d_igemm family: conv_igemm_dma / conv_halo / conv_resident64 / conv_regw / conv_ksplit256 / conv_small / conv_pw, split-K launches incl. their splitk_reduce (all %d vd3d_conv2d_igemm calls per step)' % nl,
                         'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                         'traffic': traffic, 'traffic_source': traffic_source, 'traffic_stale': traffic_stale,
                         'library_sources_sha16': lib_hash, 'library_matches_tree': library_current,
                         'traffic_unit': 'bytes per step over the conv launches (PMC FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes)',
                         'algorithmic_bytes': alg_bytes,
                         'dominant_layer': dict(dominant, unit='TFLOP/s', frac=round(dominant['achieved'] / peak, 4)),
                         'whole_path_frac': round(value / world * GFLOP_PER_PAIR * (args.height * args.width) / (384 * 1280) / 1e3 / peak, 4),
                         # the HBM-bound stages of the step (north_star: "rocprof HBM GB/s"): algorithmic bytes / HIP-event duration per launch
                         # against the HBM roof; the counter side (FETCH / WRITE) is profiles/r*_pmc_traffic.json `hbm_kernels`
                         'hbm_kernels': hbm_kernels, 'hbm_peak_gbps': PEAK_HBM_GBPS, 'hbm_achievable_gbps': ACHIEVABLE_HBM_GBPS,
                         'launch_time_statistic': 'per launch: the median of 3 serial HIP-event passes; a family = the sum over its launches'},
        }
        if gather_us is not None:
            line['config']['all_gather_us_per_step_rank0'] = round(gather_us, 1)
        if n_fl == 2 and not feed:
            # for the record: the same steps one at a time (this rank alone, no collective), with the intra-step side streams that suit a lone step: rounds
            # 1 - 5's loop.  A third detector object for it (the replicas' graphs were captured without forks)
            model1 = build_model(args, device)[0]
            apply_knobs(model1, no_forks=False)
            stepper1 = Stepper(model1, inputs, B, device)
            torch.cuda.synchronize()
            stepper1.run(args.warmup)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            stepper1.run(args.steps)
            torch.cuda.synchronize()
            el1 = time.perf_counter() - t1
            del stepper1, model1
            line['one_in_flight'] = {'ms_per_step': round(el1 / args.steps * 1e3, 3), 'value_this_rank': round(B * args.steps / el1, 2), 'unit': 'img/s',
                                     'what': 'ONE detector object, steps back to back, intra-step side streams on (--in-flight 1: the loop of rounds 1 - 5), one timed region on rank 0 after the headline regions'}
        if world == 1 and not dist and not args.no_other_configs and args.feed == 'resident' and args.dtype == 'bf16' and not args.no_graph:
            # BASELINE configs 3 and 5 as stated, driver-observed: same rules, after the headline's timed region (~10 s extra)
            del stepper
            del reps[:]                     # (the second replica's graph and buffers go too)
            torch.cuda.empty_cache()
            others = []
            for c in OTHER_CONFIGS:
                try:
                    others.append(time_other_config(c, device, steps=max(5, min(args.steps, 20)), warmup=max(2, min(args.warmup, 5)), in_flight=n_fl))
                except Exception as e:      # noqa: BLE001  (an extra config must never take the headline line down with it)
                    others.append(dict(config=c['key'], workload=c['workload'], error='%s: %s' % (type(e).__name__, e)))
            for c in API_CONFIGS:               # batch-1 calls through the reference's own entry point
                try:
                    others.append(time_api_config(c, device))
                except Exception as e:      # noqa: BLE001
                    others.append(dict(config=c['key'], workload=c['workload'], error='%s: %s' % (type(e).__name__, e)))
            line['other_configs'] = others
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(cfg, sd, args)
        # LAST key, compact: every configuration's [ms per step or call, value (img/s), whole-path fraction of the 2.5 PF MFMA peak, detections per
        # frame (mean)] -- so that a 2 000-character tail of this line still carries all of them
        summ = {'C2': [round(ms, 3), round(value, 1), line['roofline']['whole_path_frac'], round(float(counts.sum()) / counts.numel(), 1)]}
        for e in line.get('other_configs', []):
            if 'error' in e and 'value' not in e:
                summ[e['config']] = 'error'
                continue
            n_fr = len(e.get('detections_per_frame', [1]))
            summ[e['config']] = [e.get('ms_per_step', e.get('ms_per_call')), e['value'], e['whole_path_frac'],
                                 round(e.get('detections_last_step', e.get('detections', 0)) / n_fr, 1)]
            if e.get('dcn_sampling'):
                summ[e['config']] += [e['dcn_sampling'].get('dcn_offset_rms_px'), e['dcn_sampling'].get('dcn_corners_in_image_frac')]
        summ['key'] = '[ms, img/s, whole_path_frac_of_2.5PF, detections_per_frame(, dcn_offset_rms_px, dcn_corners_in_image_frac)]'
        line['summary'] = summ
        json_out.write(json.dumps(line) + '\n')
        json_out.flush()
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == '__main__':
    main()
