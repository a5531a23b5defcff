"""GPU: bench.py's multi-GPU step on ONE rank over RCCL (VD3D_BENCH_FORCE_DIST=1): process group init, the graph-captured
pack kernel, the double-buffered all_gather on the comm stream and the device->host copy of the gathered record -- and the
gathered detections equal what ``forward_device`` returns directly.  (8-GPU runs are the driver's; world-size-2 semantics
are covered on CPU over gloo in tests/test_distributed_cpu.py.)"""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('graph', [True, False])
def test_bench_force_dist_gathers_the_detections_forward_device_returns(graph):
    dump = os.path.join(tempfile.mkdtemp(), 'dump.pt')
    env = dict(os.environ, VD3D_BENCH_FORCE_DIST='1', VD3D_BENCH_DUMP=dump, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '3', '--warmup', '2', '--no-cpu-baseline'] + ([] if graph else ['--no-graph'])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, 'stdout must carry the ONE JSON line and nothing else (RCCL prints its banner through C stdio):\n' + r.stdout[-1500:]
    line = json.loads(r.stdout.strip())
    assert line['n_gpus'] == 1 and line['config']['hip_graph'] == graph and line['value'] > 100
    assert '[bench] rank 0/1' in r.stderr                              # per-rank timing line (diagnosable SCALE runs)
    d = torch.load(dump)
    host = d['host']                                                    # [world=1, B, KDET + 1, 13]
    scores, boxes, labels, aidx, count = d['direct']
    B, K = scores.shape
    k = host.shape[2] - 1
    assert host.shape[0] == 1 and host.shape[1] == B
    total = 0
    for b in range(B):
        n = int(count[b])
        assert n >= 0 and int(host[0, b, k, 0]) == n
        assert torch.equal(host[0, b, :n, 0], scores[b, :n])
        assert torch.equal(host[0, b, :n, 1:12], boxes[b, :n])
        assert torch.equal(host[0, b, :n, 12].long(), labels[b, :n].long())
        assert bool((host[0, b, n:k] == 0).all())
        total += n
    assert total >= 8


@pytest.mark.parametrize('steps,in_flight', [(4, 2), (3, 2), (4, 1)])
def test_bench_in_flight_records_are_the_detections_forward_device_returns(steps, in_flight):
    """`python bench.py` (one process, no collective): with two steps in flight (two replicas of the detector on two streams, the default) the LAST step's
    host-side record -- replica (steps - 1) % 2, pinned slot ((steps - 1) // 2) & 1 -- holds exactly what `forward_device` returns, and the line says how many steps
    were in flight and what one in flight measures."""
    dump = os.path.join(tempfile.mkdtemp(), 'dump.pt')
    env = dict(os.environ, VD3D_BENCH_DUMP=dump, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'VD3D_BENCH_FORCE_DIST'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--steps', str(steps), '--warmup', '2', '--no-cpu-baseline', '--no-other-configs', '--regions', '1',
           '--in-flight', str(in_flight)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip())
    assert line['config']['in_flight'] == in_flight and ('one_in_flight' in line) == (in_flight == 2)
    if in_flight == 2:
        assert line['one_in_flight']['ms_per_step'] > 0
    d = torch.load(dump)
    host = d['host']
    scores, boxes, labels, aidx, count = d['direct']
    k = host.shape[2] - 1
    total = 0
    for b in range(scores.shape[0]):
        n = int(count[b])
        assert n >= 0 and int(host[0, b, k, 0]) == n
        assert torch.equal(host[0, b, :n, 0], scores[b, :n]) and torch.equal(host[0, b, :n, 1:12], boxes[b, :n])
        assert torch.equal(host[0, b, :n, 12].long(), labels[b, :n].long()) and bool((host[0, b, n:k] == 0).all())
        total += n
    assert total >= 8


def test_pack_detections_kernel_matches_host_pack():
    from visualdet3d_amd import distributed as vdist, hip_ops
    g = torch.Generator().manual_seed(0)
    B, K, k = 5, 40, 16
    scores = torch.rand(B, K, generator=g)
    boxes = torch.randn(B, K, 11, generator=g)
    labels = torch.randint(0, 3, (B, K), generator=g, dtype=torch.int32)
    count = torch.tensor([0, 3, 16, 40, -1], dtype=torch.int32)
    got = hip_ops.pack_detections(scores.cuda(), boxes.cuda(), labels.cuda(), count.cuda(), k).cpu()
    want, _ = vdist.pack_detections(scores, boxes, labels, count, k)
    assert torch.equal(got[:, :k], want) and torch.equal(got[:, k, 0], count.float()) and bool((got[:, k, 1:] == 0).all())
    gth = vdist.DetectionGather(B, k, 'cuda', world=1)
    gth.fill(scores.cuda(), boxes.cuda(), labels.cuda(), count.cuda())
    assert torch.equal(gth.pack.cpu(), got)


def test_bench_feed_host_uploads_frames_and_preprocesses_inside_the_step():
    """``bench.py --feed host`` (world = 1): every step uploads 2 x B uint8 camera frames (the resident workload's images as bytes) from pinned host memory into the
    device ring on the copy stream and runs vd3d_preprocess_image inside the captured step.  The network inputs the step produced
    equal the oracle's preprocessing (oracle/preprocess_ref.py, pinned to the reference's augmentation classes) of the frames of
    the LAST step's ring slot, and the detections that reached the host are the ones forward_device returns for those inputs."""
    import numpy as np
    from oracle import preprocess_ref
    dump = os.path.join(tempfile.mkdtemp(), 'dump.pt')
    steps, B = 3, 2
    env = dict(os.environ, VD3D_BENCH_DUMP=dump, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--steps', str(steps), '--warmup', '2', '--batch', str(B), '--no-cpu-baseline',
           '--feed', 'host']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['config']['feed'].startswith('host:') and 'other_configs' not in line and line['value'] > 50
    d = torch.load(dump)
    L, R = d['inputs']
    sys.path.insert(0, REPO)
    import bench
    from visualdet3d_amd.utils import synthetic as syn
    slots = bench.HostFeed.slot_frames(*syn.stereo_pair(B, 384, 1280, seed=100))      # (bench.py: VD3D_BENCH_SEED 100 + rank 0) slot 0 then slot 1
    frames = slots[(steps - 1) & 1]
    assert frames.shape == (2 * B, 384, 1280, 3) and frames.dtype == torch.uint8
    import re
    assert int(re.search(r'\((\d+) detections in the last step', r.stderr).group(1)) >= 1, 'the host-fed workload must not time decode / NMS on an empty candidate list'
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    for b, dst in ((0, L[0]), (B - 1, L[B - 1]), (B, R[0]), (2 * B - 1, R[B - 1])):
        want = preprocess_ref.preprocess(frames[b].numpy(), 0, (384, 1280), mean, std)
        np.testing.assert_allclose(dst.numpy(), want, rtol=0, atol=2e-6)
    host = d['host']
    scores, boxes, labels, aidx, count = d['direct']
    k = host.shape[2] - 1
    for b in range(B):
        n = int(count[b])
        assert n >= 0 and int(host[0, b, k, 0]) == n
        assert torch.equal(host[0, b, :n, 0], scores[b, :n]) and torch.equal(host[0, b, :n, 1:12], boxes[b, :n])
