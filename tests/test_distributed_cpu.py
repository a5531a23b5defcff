"""CPU, world_size 2 over gloo: the batch-shard + detection-gather path gives exactly the single-process result."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visualdet3d_amd import distributed as vdist


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world2(target, args_after_port, timeout):
    return _run_world(2, target, args_after_port, timeout)


def _run_world(world, target, args_after_port, timeout):
    """spawn `world` ranks of `target(rank, world, port, *args_after_port, q)`, return what rank 0 put on the queue.  The rendezvous port is chosen by
    binding port 0 and releasing it: another process can take it in between (seen once in ~50 runs of the suite) -- a failed rendezvous is
    retried on a fresh port, a failure of the ranks' own assertions is not masked (it fails all three attempts the same way)."""
    last = None
    for attempt in range(3):
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args_after_port) + (q,)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            out = q.get(timeout=timeout)
            for p in procs:
                p.join(timeout=timeout)
            codes = [p.exitcode for p in procs]
            if codes == [0] * world:
                return out
            last = AssertionError('rank exit codes %s (attempt %d)' % (codes, attempt + 1))
        except Exception as e:                       # noqa: BLE001 -- queue.Empty: a rank died before reporting
            last = e
        for p in procs:
            if p.is_alive():
                p.terminate()
            p.join(timeout=30)
    raise last


def _fake_padded(n_frames, K=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    count = torch.randint(0, K + 1, (n_frames,), generator=g, dtype=torch.int32)
    scores = torch.rand(n_frames, K, generator=g).sort(dim=1, descending=True)[0]
    boxes = torch.randn(n_frames, K, 11, generator=g)
    labels = torch.randint(0, 3, (n_frames, K), generator=g, dtype=torch.int32)
    return scores, boxes, labels, count


def _worker(rank, world, port, n_frames, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    scores, boxes, labels, count = _fake_padded(n_frames)
    lo, hi = vdist.shard_range(n_frames, rank, world)
    assert hi - lo == n_frames // world          # equal shards in this test (all_gather needs equal shapes)
    pack, cnt = vdist.pack_detections(scores[lo:hi], boxes[lo:hi], labels[lo:hi], count[lo:hi], k=16)
    allp, allc = vdist.gather_detections(pack, cnt)
    # the single-collective, allocation-free variant bench.py uses every step (twice: the buffers are reused)
    g1 = vdist.DetectionGather(hi - lo, 16, 'cpu')
    for _ in range(2):
        g1(scores[lo:hi], boxes[lo:hi], labels[lo:hi], count[lo:hi])
    p1, c1 = g1.detections()
    if rank == 0:
        q.put((allp, allc, p1.clone(), c1.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2_matches_single_process():
    n = 6
    allp, allc, p1, c1 = _run_world2(_worker, (n,), 120)
    scores, boxes, labels, count = _fake_padded(n)
    want_p, want_c = vdist.pack_detections(scores, boxes, labels, count, k=16)
    assert torch.equal(allp, want_p) and torch.equal(allc, want_c)
    assert torch.equal(p1, want_p) and torch.equal(c1, want_c)
    dets = vdist.unpack_detections(allp, allc)
    assert len(dets) == n and all(d[0].numel() == int(c) for d, c in zip(dets, count))
    assert dets[2][2].dtype == torch.int64


def test_shard_range_covers_batch():
    for n in (1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [vdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


# ---- detector output through the gather (world 2): each rank runs the CPU oracle of the detector on ITS shard of the frames ------
def _oracle_detections(lo, hi):
    """(scores [n,K], boxes [n,K,11], labels [n,K], count [n]) padded like ``get_bboxes_batched`` returns them, from the oracle's
    Stereo3D forward on the frames [lo, hi) of the 96x320 golden case (weights / inputs are seeded: every rank builds the same)."""
    from oracle import detector_oracle as orc
    from tests.common import load_golden, stereo_case_from_golden
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3D
    from visualdet3d_amd.utils import synthetic as syn
    g = load_golden('stereo3d_r34_96x320')
    cfg, (L, R, P2, P3), winit = stereo_case_from_golden(g)
    sd = syn.seeded_state_dict(Stereo3D(cfg).state_dict(), **winit)
    with torch.no_grad():
        outs = orc.stereo3d_forward(sd, cfg, L[lo:hi], R[lo:hi], P2[lo:hi])
    K = 128
    n = hi - lo
    scores, boxes = torch.zeros(n, K), torch.zeros(n, K, 11)
    labels, count = torch.zeros(n, K, dtype=torch.int32), torch.zeros(n, dtype=torch.int32)
    for f, (s_, b_, l_, _) in enumerate(outs):
        k = len(s_)
        assert k <= K
        scores[f, :k], boxes[f, :k], labels[f, :k], count[f] = s_, b_, l_.reshape(-1).int(), k
    return scores, boxes, labels, count, L.shape[0]


def _detector_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tests.common import load_golden
    n_frames = int(load_golden('stereo3d_r34_96x320')['meta'][3])
    lo, hi = vdist.shard_range(n_frames, rank, world)
    scores, boxes, labels, count, _ = _oracle_detections(lo, hi)
    gather = vdist.DetectionGather(hi - lo, 128, 'cpu')
    gather(scores, boxes, labels, count)
    pack, cnt = gather.detections()
    if rank == 0:
        q.put((pack.clone(), cnt.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_detector_output_through_the_gather_world2():
    """What bench.py does per step with N ranks, on real detector output: every rank runs the detector (here its CPU oracle) on its
    shard of the golden case's frames, DetectionGather moves the padded results to every rank in one collective, and rank 0's
    unpacked detections equal the reference's golden detections frame by frame."""
    from tests.common import assert_detections_close, load_golden
    g = load_golden('stereo3d_r34_96x320')
    n_frames = int(g['meta'][3])
    assert n_frames % 2 == 0
    pack, cnt = _run_world2(_detector_worker, (), 600)
    dets = vdist.unpack_detections(pack, cnt)
    assert len(dets) == n_frames and sum(int(c) for c in cnt) > 0
    for f, (s_, b_, l_) in enumerate(dets):
        assert_detections_close((s_, b_, l_), (g['f%d_scores' % f], g['f%d_boxes' % f], g['f%d_labels' % f]), rtol=1e-4,
                                what='gathered frame %d' % f)


# ---- world 8: BASELINE config 4's partition (64 frames -> 8 x 8), per-rank data, one collective per step ---------------------------
def _worker8(rank, world, port, n_frames, steps, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = vdist.shard_range(n_frames, rank, world)
    B = hi - lo
    gathers = [vdist.DetectionGather(B, 32, 'cpu') for _ in range(2)]          # double buffered like bench.py's comm stream
    last = None
    for i in range(steps):
        # every rank produces ITS frames of step i from its own seed (bench.py: stereo_pair(seed = 100 + rank)); frame f of the job is rank f // B's
        scores, boxes, labels, count = _fake_padded(B, K=40, seed=1000 * i + rank)
        g = gathers[i & 1]
        g(scores, boxes, labels, count)
        last = g.detections()
    if rank == 0:
        q.put((last[0].clone(), last[1].clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_world8_gather_of_64_frames_equals_the_single_process_result():
    """64 frames on 8 ranks (8 each, contiguous shards, per-rank seeds): what rank 0 holds after the step's ONE collective is exactly what a single
    process computes for all 64 frames, in frame order -- for the last of three double-buffered steps."""
    world, n, steps = 8, 64, 3
    pack, cnt = _run_world(world, _worker8, (n, steps), 300)
    want_p, want_c = [], []
    for r in range(world):
        lo, hi = vdist.shard_range(n, r, world)
        assert (lo, hi) == (8 * r, 8 * r + 8)
        scores, boxes, labels, count = _fake_padded(hi - lo, K=40, seed=1000 * (steps - 1) + r)
        p, c = vdist.pack_detections(scores, boxes, labels, count, k=32)
        want_p.append(p)
        want_c.append(c)
    assert torch.equal(pack, torch.cat(want_p)) and torch.equal(cnt, torch.cat(want_c))
    assert len(vdist.unpack_detections(pack, cnt)) == n


# ---- host placement: eight ranks of one node never share a core ---------------------------------------------------------------------
def _fake_sysfs(root, gpu_nodes, node_cpus):
    for i, node in enumerate(gpu_nodes):
        d = os.path.join(root, 'bus/pci/devices/0000:%02x:00.0' % (0x10 + i))
        os.makedirs(d)
        open(os.path.join(d, 'numa_node'), 'w').write('%d\n' % node)
    for node, cpus in node_cpus.items():
        d = os.path.join(root, 'devices/system/node/node%d' % node)
        os.makedirs(d)
        open(os.path.join(d, 'cpulist'), 'w').write(cpus + '\n')
    return ['0000:%02x:00.0' % (0x10 + i) for i in range(len(gpu_nodes))]


def test_rank_cpu_sets_partition_each_numa_node_among_its_gpus(tmp_path):
    # an MI355X node: 8 GPUs, 4 per socket; 2 x 64 cores with SMT siblings numbered 128+
    bdfs = _fake_sysfs(str(tmp_path), [0, 0, 0, 0, 1, 1, 1, 1], {0: '0-63,128-191', 1: '64-127,192-255'})
    sets = vdist.rank_cpu_sets(bdfs, sysfs=str(tmp_path))
    assert all(len(s) == 32 for s in sets)
    for a in range(8):
        node_cpus = vdist._parse_cpulist('0-63,128-191' if a < 4 else '64-127,192-255')
        assert sets[a] <= node_cpus, 'rank %d left its GPU\'s NUMA node' % a
        for b in range(a + 1, 8):
            assert not (sets[a] & sets[b]), 'ranks %d and %d share cores' % (a, b)
    # restricted by the cgroup: only the allowed CPUs are divided; a GPU without a node is not pinned
    sets = vdist.rank_cpu_sets(bdfs, sysfs=str(tmp_path), allowed=range(0, 8))
    assert [sorted(s) for s in sets[:4]] == [[0, 1], [2, 3], [4, 5], [6, 7]] and sets[4:] == [None] * 4
    bdfs2 = _fake_sysfs(str(tmp_path / 'b'), [-1, 0], {0: '0-3'})
    assert vdist.rank_cpu_sets(bdfs2, sysfs=str(tmp_path / 'b')) == [None, {0, 1, 2, 3}]


def _producer(rank, sysfs, bdfs, q):
    """one --feed host producer: pin like bench.py does, then make the ring's uint8 frames (HostFeed.slot_frames) and report where it ran"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cpus = vdist.rank_cpu_sets(bdfs, sysfs=sysfs, allowed=os.sched_getaffinity(0))[rank]
    os.sched_setaffinity(0, cpus)
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(rank)
    L, R = torch.randn(1, 3, 32, 64, generator=g), torch.randn(1, 3, 32, 64, generator=g)
    frames = bench.HostFeed.slot_frames(L, R)
    q.put((rank, sorted(os.sched_getaffinity(0)), int(frames[0].shape[0]), str(frames[0].dtype)))


def test_eight_concurrent_host_feed_producers_run_on_disjoint_cores(tmp_path):
    """Eight producer processes (one per rank of a node), pinned by the rule bench.py applies (`pin_to_gpu_numa_node` -> `rank_cpu_sets`), on a fake
    topology laid over THIS machine's CPUs (two 'NUMA nodes' = the two halves of the allowed CPUs, four 'GPUs' each): every process ends up on its
    own core(s), inside its GPU's node."""
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 8:
        import pytest
        pytest.skip('needs 8 CPUs')
    half = len(allowed) // 2
    lists = {0: ','.join(map(str, allowed[:half])), 1: ','.join(map(str, allowed[half:]))}
    bdfs = _fake_sysfs(str(tmp_path), [0, 0, 0, 0, 1, 1, 1, 1], lists)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_producer, args=(r, str(tmp_path), bdfs, q)) for r in range(8)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(8))
    for p in procs:
        p.join(timeout=60)
    assert [p.exitcode for p in procs] == [0] * 8
    seen = set()
    for rank, cpus, n, dt in got:
        assert cpus and not (set(cpus) & seen), 'rank %d shares a core with another producer' % rank
        seen |= set(cpus)
        assert set(cpus) <= set(allowed[:half] if rank < 4 else allowed[half:])
        assert n == 2 and dt == 'torch.uint8'
