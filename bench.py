"""bench.py -- images/sec of the YOLOStereo3D forward path (BASELINE.json configs[1]: Stereo3D ResNet-34,
384x1280 stereo pairs, bf16, batch 8 per MI355X), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one test_forward_batched-equivalent pass over one batch of synthetic pairs already resident in HBM:
stem -> ResNet-34 (L and R stacked) -> cost volumes -> ghost pyramid -> head towers -> device-side decode/NMS ->
(N > 1: RCCL all_gather of the padded detections, the only collective) -> one host sync for the counts.
Prints ONE JSON line (rank 0) with the contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GFLOP_PER_PAIR = 473.82          # conv/GEMM 2*MAC per 384x1280 pair (Stereo3D R34)
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='stereo pairs per GPU per step')
    ap.add_argument('--height', type=int, default=384)
    ap.add_argument('--width', type=int, default=1280)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-graph', action='store_true', help='launch kernels eagerly instead of replaying a hipGraph')
    ap.add_argument('--no-overlap', action='store_true',
                    help='no side streams (cls tower / stereo neck run serially on the main stream): the configuration whose '
                         'rocprofv3 --kernel-trace durations are additive (profiles/*_serial_*), NOT the headline configuration')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true',
                    help='no GPU work: time the CPU baseline(s) on this host and print them (used in the build container, where '
                         'the reference tree exists, to calibrate the "port" against the reference itself)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    return ap.parse_args()


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


def profile_convs(model, inputs, reps=3):
    """Per-launch HIP-event timing of the dominant kernel family (conv_igemm) on the launch stream.
    Returns (total_flops_per_step, total_seconds_per_step, n_launches)."""
    from visualdet3d_amd import hip_ops as ops
    records = []
    orig = ops.conv2d

    def timed(x, pc, out=None, residual=None, relu=False, out_f32=False):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        o = orig(x, pc, out=out, residual=residual, relu=relu, out_f32=out_f32)
        e.record()
        B, Ho, Wo, Co = o.shape
        nbytes = (x.shape[0] * x.shape[1] * x.shape[2] * pc.Cin * x.element_size() + pc.Cout * pc.kh * pc.kw * pc.Cin * x.element_size()
                  + o.numel() * o.element_size() + (residual.numel() * residual.element_size() if residual is not None else 0))
        records.append((2.0 * B * Ho * Wo * Co * pc.kh * pc.kw * pc.Cin, s, e, nbytes,
                        '%dx%d s%d %4d->%4d @ %dx%dx%d' % (pc.kh, pc.kw, pc.stride, pc.Cin, pc.Cout, B, Ho, Wo)))
        return o

    ops.conv2d = timed
    overlap = model.bbox_head.overlap_towers
    overlap_neck = model.core.overlap_neck
    model.bbox_head.overlap_towers = False     # serial launches: per-kernel event times must not overlap
    model.core.overlap_neck = False
    try:
        with torch.no_grad():
            for _ in range(reps):
                model.forward_device(*inputs)
        torch.cuda.synchronize()
    finally:
        ops.conv2d = orig
        model.bbox_head.overlap_towers = overlap
        model.core.overlap_neck = overlap_neck
    # No event-overhead correction: per-launch HIP events were compared with the kernel durations of a rocprofv3 --kernel-trace
    # of this very command (`bench.py --no-overlap`, tools/profile_round.sh -> profiles/*_serial_roofline_check.txt): the raw
    # event sums agree with the profiler within ~1 % (an empty event pair reads 5-6 us, but around a kernel that cost is not
    # inside the reading: subtracting it made the bench read 3.5-8 % short of the profiler).
    ovh_ms = 0.0
    if os.environ.get('VD3D_BENCH_LAYERS'):
        per = len(records) // reps
        for i in range(per):
            fl = records[i][0]
            t = sum(records[i + r * per][1].elapsed_time(records[i + r * per][2]) - ovh_ms for r in range(reps)) / reps
            print('  conv %2d  %-32s %8.2f GF  %8.1f us  %7.1f TF/s' % (i, records[i][4], fl / 1e9, t * 1e3, fl / (t * 1e-3) / 1e12), file=sys.stderr)
    flops = sum(r[0] for r in records) / reps
    secs = sum(max(r[1].elapsed_time(r[2]) - ovh_ms, 0.0) for r in records) * 1e-3 / reps
    # the single most expensive layer shape of the family (its launches all run the same kernel): reported next to the family figure
    by_shape = {}
    for r in records:
        d = by_shape.setdefault(r[4], [0.0, 0.0, 0])
        d[0] += r[0]
        d[1] += max(r[1].elapsed_time(r[2]) - ovh_ms, 0.0) * 1e-3
        d[2] += 1
    name, (fl, tt, n) = max(by_shape.items(), key=lambda kv: kv[1][1])
    dominant = dict(layer=name, launches_per_step=n // reps, share_of_family_time=round(tt / (secs * reps), 4),
                    avg_launch_us=round(tt / n * 1e6, 1), achieved=round(fl / tt / 1e12, 1))
    return flops, secs, len(records) // reps, sum(r[3] for r in records) / reps, ovh_ms * 1e3, dominant


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


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = world > 1 or bool(os.environ.get('VD3D_BENCH_FORCE_DIST'))   # the env: exercise the RCCL path on one GPU
    if args.cpu_baseline_only:
        _, cfg, sd = build_model(args, torch.device('cpu'))
        os.environ['VD3D_BENCH_CPU_PORT'] = '1'
        port = cpu_baseline(cfg, sd, args)
        del os.environ['VD3D_BENCH_CPU_PORT']
        ref = cpu_baseline(cfg, sd, args)
        print(json.dumps({'cpu_baseline_port': port, 'cpu_baseline': ref}))
        return
    assert torch.cuda.is_available(), 'bench.py measures the MI355X HIP path; no GPU visible'
    torch.cuda.set_device(local_rank)              # before the process group: RCCL binds to the current device
    device = torch.device('cuda', local_rank)
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

    if os.environ.get('VD3D_BENCH_NONECK') or args.no_overlap:
        model.core.overlap_neck = False
    if os.environ.get('VD3D_BENCH_NOTOWER') or args.no_overlap:
        model.bbox_head.overlap_towers = False
    from visualdet3d_amd import hip_ops
    KDET = 128                                   # detections per frame that travel (device -> host, rank -> ranks)
    pack_static = torch.zeros((B, KDET + 1, 13), dtype=torch.float32, device=device)

    def step_device():
        """One step on the device: the whole forward incl. decode + NMS, then ONE launch that packs the padded results (scores,
        boxes, labels, per-frame count) into the [B, KDET + 1, 13] record that is copied to the host / gathered."""
        scores, boxes, labels, aidx, count = model.forward_device(*inputs)
        hip_ops.pack_detections(scores, boxes, labels, count, KDET, out=pack_static)
        return scores, boxes, labels, aidx, count

    graph = None
    static_out = None
    with torch.no_grad():
        for _ in range(2):                       # packs weights, builds anchor tables, warms the allocator
            static_out = step_device()
        torch.cuda.synchronize()
        if not args.no_graph:
            if not os.environ.get('VD3D_BENCH_NOSIDE'):
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    step_device()
                torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = step_device()

    # Results leave the device every step, inside the timed region: the packed record of the step (B x (KDET + 1) x 13 floats,
    # ~53 KB) is copied to pinned host memory and its counts are checked on the host.
    # N > 1: the only collective is the gather of that record -- ONE RCCL all_gather per step, latency bound.  It runs on its
    # own stream, double buffered: the comm stream first copies the graph's static record into send buffer (i & 1) (so the
    # next replay may overwrite the record at once), then gathers, then copies the gathered block to the host; the host reads
    # step k's results while step k+1 is already enqueued, so the collective's latency overlaps the next forward.
    gatherers, comm_stream, taken_ev, packed_ev, done_ev = None, None, None, None, None
    if dist:
        import torch.distributed as td
        from visualdet3d_amd.distributed import DetectionGather
        gatherers = [DetectionGather(B, KDET, device, world) for _ in range(2)]
        comm_stream = torch.cuda.Stream()
        packed_ev = [torch.cuda.Event() for _ in range(2)]
        taken_ev = [torch.cuda.Event() for _ in range(2)]
        done_ev = [torch.cuda.Event() for _ in range(2)]
        pinned = [torch.empty((world, B, KDET + 1, 13), dtype=torch.float32).pin_memory() for _ in range(2)]
    else:
        pinned = [torch.empty((1, B, KDET + 1, 13), dtype=torch.float32).pin_memory()]

    def forward_step():
        if graph is not None:
            graph.replay()
            return static_out
        with torch.no_grad():
            return step_device()

    def check(host_block):
        """host_block [ranks, B, KDET + 1, 13] (pinned): per-frame counts ride in row KDET."""
        c = host_block[:, :, KDET, 0]
        assert float(c.min()) >= 0, 'candidate overflow in the head post-processing'
        return c.clone()

    def run(n):
        """n steps; returns the (host) detection counts [ranks, B] of the last one."""
        counts = None
        if not dist:
            for _ in range(n):
                forward_step()
                pinned[0][0].copy_(pack_static, non_blocking=True)      # device -> host copy of the step's results
                torch.cuda.current_stream().synchronize()               # the one host sync per step
                counts = check(pinned[0])
            return counts
        main = torch.cuda.current_stream()

        def collect(i):
            done_ev[i & 1].synchronize()
            return check(pinned[i & 1])

        for i in range(n):
            g = gatherers[i & 1]
            if i >= 1:
                main.wait_event(taken_ev[(i - 1) & 1])       # step i-1's record has been copied out of the static buffer
            forward_step()
            packed_ev[i & 1].record(main)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(packed_ev[i & 1])
                g.pack.copy_(pack_static, non_blocking=True)
                taken_ev[i & 1].record(comm_stream)
                out = g.gather()
                pinned[i & 1].copy_(out, non_blocking=True)
                done_ev[i & 1].record(comm_stream)
            if i >= 1:
                counts = collect(i - 1)
        if n >= 1:
            counts = collect(n - 1)
        return counts

    dbg = bool(os.environ.get('VD3D_BENCH_DEBUG'))
    if dbg:
        print('[bench] captured=%s, entering warmup' % (graph is not None), file=sys.stderr, flush=True)
    run(args.warmup)
    if dbg:
        print('[bench] warmup done', file=sys.stderr, flush=True)
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    counts = run(args.steps)
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    elapsed = time.perf_counter() - t0
    if dbg:
        print('[bench] timed region done %.3f s' % elapsed, file=sys.stderr, flush=True)
    print('[bench] rank %d/%d on cuda:%d: %d steps in %.4f s = %.3f ms/step (%d detections in the last step%s)'
          % (rank, world, local_rank, args.steps, elapsed, elapsed / args.steps * 1e3, int(counts.sum()),
             ' over all ranks' if dist else ''), file=sys.stderr, flush=True)
    if dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
    assert counts is not None and float(counts.min()) >= 0, 'candidate overflow in the head post-processing'
    if os.environ.get('VD3D_BENCH_DUMP'):
        # test hook (tests/test_bench_dist_gpu.py): the last step's host-side results next to forward_device's own
        torch.save(dict(host=pinned[(args.steps - 1) & 1 if dist else 0].clone(), direct=[t.cpu() for t in forward_step()]),
                   os.environ['VD3D_BENCH_DUMP'])

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        flops, secs, nl, alg_bytes, ev_us, dominant = profile_convs(model, inputs)
        traffic = None
        pmc = os.path.join(REPO, 'profiles', 'r02_pmc_traffic.json')
        if not os.path.exists(pmc):
            pmc = os.path.join(REPO, 'profiles', 'r01_pmc_traffic.json')
        if args.dtype == 'bf16' and B == 8 and os.path.exists(pmc):
            traffic = json.load(open(pmc))['conv_hbm_bytes_per_forward']   # PMC pass of this same command, see profiles/
        peak = PEAK_BF16_TFLOPS if args.dtype == 'bf16' else PEAK_F32_TFLOPS
        ach = flops / secs / 1e12
        line = {
            'metric': 'images/sec at 384x1280 stereo (YOLOStereo3D ResNet-34 forward incl. decode+NMS)',
            'value': round(value, 2), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'Stereo3D_example (YOLOStereo3D, ResNet-34) %dx%d stereo pairs, batch=%d per GPU'
                                   % (args.height, args.width, B),
                       'global_batch': world * B, 'parallelism': 'dp%d' % world, 'hip_graph': graph is not None,
                       'side_streams': not args.no_overlap, 'results_d2h_bytes_per_step': int(pack_static.numel() * 4 * world)},
            'roofline': {'bound': 'mfma', 'kernel': 'vd3d_conv2d_igemm family: conv_igemm_dma / conv_halo / conv_resident64 / conv_regw / conv_ksplit256 / conv_small / conv_pw (all %d launches per step)' % nl,
                         'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                         'traffic': traffic, 'traffic_unit': 'bytes per step over the conv launches (PMC FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes)',
                         'algorithmic_bytes': alg_bytes,
                         'dominant_layer': dict(dominant, unit='TFLOP/s', frac=round(dominant['achieved'] / peak, 4)),
                         'whole_path_frac': round(value / world * GFLOP_PER_PAIR * (args.height * args.width) / (384 * 1280) / 1e3 / peak, 4)},
        }
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(cfg, sd, args)
        print(json.dumps(line))
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == '__main__':
    main()
