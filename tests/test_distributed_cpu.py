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
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    allp, allc, p1, c1 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
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
