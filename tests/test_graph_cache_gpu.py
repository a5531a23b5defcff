"""GPU: the hipGraph cache behind ``test_forward`` / ``test_forward_batched`` (visualdet3d_amd/networks/lib/graphed.py) -- the
reference's call pattern, one frame per ``module([...])`` call (networks/pipelines/testers.py:15-42), served by graph replays.

  * replayed results are bit-identical to the eager launches of the same kernels;
  * results handed out are the caller's own (a later call does not overwrite them);
  * the cache is invalidated by what invalidates the packed weights: load_state_dict / in-place updates, .to(), changed head
    settings, the library's test hooks."""
import pytest
import torch

from tests.common import load_golden, mono_case_from_golden, stereo_case_from_golden
from visualdet3d_amd import _lib
from visualdet3d_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def _stereo(dtype=torch.bfloat16):
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3D
    g = load_golden('stereo3d_r34_96x320')
    cfg, (L, R, P2, P3), winit = stereo_case_from_golden(g)
    m = Stereo3D(cfg)
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), **winit))
    m = m.cuda().eval()
    m.compute_dtype = dtype
    return m, cfg, [t.cuda() for t in (L, R, P2, P3)], winit


def _same(a, b):
    return len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_replay_equals_eager_and_results_are_owned(dtype):
    m, cfg, (L, R, P2, P3), _ = _stereo(dtype)
    L2, R2 = syn.stereo_pair(L.shape[0], L.shape[2], L.shape[3], seed=77)
    L2, R2 = L2.cuda(), R2.cuda()
    a = m([L[:1], R[:1], P2[:1], P3[:1]])                       # capture + replay
    a_copy = [t.clone() for t in a]
    b = m([L2[:1], R2[:1], P2[:1], P3[:1]])                     # replay on other data
    assert m.graph_stats == dict(captures=1, replays=2, eager=0, cached=1)
    assert _same(a, a_copy), 'a later replay overwrote results already handed to the caller'
    assert a[0].numel() > 0 and not _same(a, b)
    m.use_graph = False
    ea = m([L[:1], R[:1], P2[:1], P3[:1]])
    eb = m([L2[:1], R2[:1], P2[:1], P3[:1]])
    assert _same(a, ea) and _same(b, eb), 'graph replay differs from the eager launches'
    assert ea[2].dtype == torch.int64 and a[2].dtype == torch.int64


def test_invalidation_on_weights_settings_and_hooks():
    m, cfg, (L, R, P2, P3), winit = _stereo()
    x = [L[:1], R[:1], P2[:1], P3[:1]]
    r0 = m(x)
    # new weights through load_state_dict (in-place copies: the parameters' versions move)
    sd2 = syn.seeded_state_dict(m.state_dict(), seed=winit['seed'] + 1, head_std=winit['head_std'])
    m.load_state_dict(sd2)
    r1 = m(x)
    assert m.graph_stats['captures'] == 2
    m.use_graph = False
    assert _same(r1, m(x)), 'stale graph after load_state_dict'
    m.use_graph = True
    # a head setting that the launches read
    m.bbox_head.test_cfg.score_thr = float(m.bbox_head.test_cfg.score_thr) + 0.05
    r2 = m(x)
    assert m.graph_stats['captures'] == 3
    assert r2[0].numel() <= r1[0].numel()
    # the library's test hooks
    with _lib.test_switch('VD3D_NO_LINE_STORE'):
        r3 = m(x)
        assert m.graph_stats['captures'] == 4
    assert _same(r2, r3)            # both settings of the switch are the same arithmetic
    # .to() / .float() replace storages: everything is dropped
    m.float()
    assert m.graph_stats['cached'] == 0
    # a batched call is its own graph; batch-1 slices of it equal the batch-1 calls
    outs = m.test_forward_batched(L, R, P2, P3)
    assert _same(outs[0], m(x))
    assert _same(r2, outs[0])
    del r0


def test_mono_calls_replay_and_host_calibration():
    from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT
    import visualdet3d_amd.networks.detectors  # noqa: F401
    g = load_golden('groundaware_r34_96x320')
    cfg, (img, P2), winit = mono_case_from_golden(g, 'groundaware_r34_96x320')
    m = DETECTOR_DICT[cfg.name](cfg)
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), **winit))
    m = m.cuda().eval()
    img, P2 = img.cuda(), P2.cuda()
    a = m([img[:1], P2[:1]])
    b = m([img[:1], P2[:1].cpu().double()])        # calibration handed over as a host float64 matrix (what a dataset yields): a new key, same result
    m.use_graph = False
    e = m([img[:1], P2[:1]])
    assert _same(a, e) and _same(b, e)
    assert m.graph_stats['eager'] == 1 and m.graph_stats['replays'] == 2


def _other_calib(P2, scale=1.07, shift=9.0):
    """a different camera: focal length scaled, principal point moved (the ground filter, the decode and the clip all read it)"""
    Q = P2.clone()
    Q[:, 0, 0] *= scale
    Q[:, 1, 1] *= scale
    Q[:, 0, 2] += shift
    Q[:, 1, 2] -= shift / 2
    return Q


@pytest.mark.parametrize('form', ['device_f32', 'host_f64', 'device_f64_strided'])
def test_replay_with_a_different_calibration_equals_eager(form):
    """the calibration of a LATER frame must reach the kernels of a replay, whatever form it arrives in.  The
    first call captures with P2; the second call hands over a DIFFERENT calibration as host float64 / strided device float64 / plain
    device fp32; each must equal the eager result for that calibration -- and differ from the first frame's."""
    m, cfg, (L, R, P2, P3), _ = _stereo()
    Q = _other_calib(P2[:1])
    if form == 'host_f64':
        q = Q.cpu().double()
    elif form == 'device_f64_strided':
        wide = torch.zeros((1, 3, 8), dtype=torch.float64, device=Q.device)
        wide[:, :, :4] = Q.double()
        q = wide[:, :, :4]
        assert not q.is_contiguous()
    else:
        q = Q
    first_form = P2[:1].cpu().double() if form != 'device_f32' else P2[:1]
    a = m([L[:1], R[:1], first_form, P3[:1]])                   # capture with the first calibration
    b = m([L[:1], R[:1], q, P3[:1]])                            # replay with another one
    c = m([L[:1], R[:1], first_form, P3[:1]])                   # and back
    assert m.graph_stats['captures'] == 1 and m.graph_stats['replays'] == 3
    m.use_graph = False
    ea = m([L[:1], R[:1], P2[:1], P3[:1]])
    eb = m([L[:1], R[:1], Q, P3[:1]])
    assert _same(a, ea) and _same(c, ea)
    assert _same(b, eb), 'replay decoded with a stale calibration'
    assert not _same(ea, eb), 'the second calibration does not change the result: the test proves nothing'


def test_mono_replay_with_a_different_host_calibration():
    """the mono detectors read P2 in three places (LookGround, ground filter, decode): same statement as above"""
    from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT
    import visualdet3d_amd.networks.detectors  # noqa: F401
    g = load_golden('groundaware_r34_96x320')
    cfg, (img, P2), winit = mono_case_from_golden(g, 'groundaware_r34_96x320')
    m = DETECTOR_DICT[cfg.name](cfg)
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), **winit))
    m = m.cuda().eval()
    img, P2 = img.cuda(), P2.cuda()
    Q = _other_calib(P2[:1])
    a = m([img[:1], P2[:1].cpu().double()])
    b = m([img[:1], Q.cpu().double()])
    assert m.graph_stats['captures'] == 1
    m.use_graph = False
    assert _same(a, m([img[:1], P2[:1]]))
    eb = m([img[:1], Q])
    assert _same(b, eb) and not _same(a, eb)


def test_a_larger_batch_does_not_free_the_scratch_of_a_cached_graph():
    """the B = 1 graph bakes the address of the head's candidate scratch; a later B = 8 call used to replace (and
    free) that buffer.  Now every batch size keeps its own for the life of the head: the B = 1 replay after the batched call -- and
    after other allocations have had the chance to land on a freed block -- still equals the eager launches."""
    m, cfg, (L, R, P2, P3), _ = _stereo()
    B = L.shape[0]
    x1 = [L[:1], R[:1], P2[:1], P3[:1]]
    a = m(x1)
    ws1 = m.bbox_head._workspace
    p1 = ws1.data_ptr()
    reps = (8 + B - 1) // B
    Lb, Rb, P2b = L.repeat(reps, 1, 1, 1)[:8], R.repeat(reps, 1, 1, 1)[:8], P2.repeat(reps, 1, 1)[:8]
    outs = m.test_forward_batched(Lb, Rb, P2b)
    assert m.bbox_head._workspace.data_ptr() != p1 and m.bbox_head._workspaces[(1, m.bbox_head.max_candidates, L.device)] is ws1
    junk = [torch.full((n,), 255, dtype=torch.uint8, device='cuda') for n in (ws1.numel(), ws1.numel() // 2, 4096, 1 << 20)]   # would land on a freed block
    b = m(x1)
    torch.cuda.synchronize()
    assert m.graph_stats['captures'] == 2 and m.graph_stats['cached'] == 2
    assert _same(a, b)                   # (outs[0] is NOT compared with a: in bf16 a batch-1 call takes the split-K kernels, another summation order)
    assert len(outs) == 8
    m.use_graph = False
    assert _same(a, m(x1))
    del junk


def test_cache_is_lru_and_keyed_on_the_anchor_filter():
    """a hit refreshes the entry (least recently USED is evicted); the anchor filter thresholds the select kernel
    reads at launch are part of the key."""
    from visualdet3d_amd.networks.lib import graphed
    m, cfg, (L, R, P2, P3), _ = _stereo()
    x = [L[:1], R[:1], P2[:1], P3[:1]]
    r0 = m(x)
    st = m._graph_state()
    k0 = next(iter(st['entries']))
    old = graphed._MAX_GRAPHS
    graphed._MAX_GRAPHS = 2
    try:
        m.bbox_head.anchors.filter_x_threshold = 20.0            # a launch-time setting: its own graph
        r1 = m(x)
        assert m.graph_stats['captures'] == 2
        m.bbox_head.anchors.filter_x_threshold = 40.0
        assert _same(m(x), r0) and m.graph_stats['captures'] == 2          # hit: k0 is now the most recently used
        m.bbox_head.anchors.filter_x_threshold = 10.0
        m(x)                                                               # third key: evicts the 20.0 graph, not k0
        assert m.graph_stats['captures'] == 3 and k0 in st['entries'] and len(st['entries']) == 2
    finally:
        graphed._MAX_GRAPHS = old
    m.use_graph = False
    m.bbox_head.anchors.filter_x_threshold = 20.0
    assert _same(r1, m(x))


def test_candidate_overflow_inside_a_replayed_graph_is_rerun_off_graph():
    """More candidates than the captured capacity (here: `max_candidates` lowered to 1 so that the golden workload overflows) must not be an
    error: the frame is re-run eagerly on the graph's own logits with a capacity that fits (detectors `_overflow_retry`), in the first (capturing)
    call, in a replay, and for one frame of a batched call -- results equal the uncapped ones bit for bit (same kernels, same arithmetic);
    and the test_cfg fingerprint of the graph key sees a key swapped for another of equal value."""
    m, cfg, (L, R, P2, P3), _ = _stereo()
    x = [L[:1], R[:1], P2[:1], P3[:1]]
    want = m(x)
    want_b = m.test_forward_batched(L, R, P2, P3)
    assert want[0].numel() >= 2                 # >= 2 detections => >= 2 candidates: a capacity of ONE must overflow
    m.bbox_head.max_candidates = 1
    got = m(x)                                  # captures a new graph (the capacity is part of the key) whose count is -1, then retries
    assert _same(got, want)
    got = m(x)                                  # replay + retry
    assert _same(got, want) and m.graph_stats['replays'] >= 2
    got_b = m.test_forward_batched(L, R, P2, P3)
    assert all(_same(a, b) for a, b in zip(got_b, want_b))
    m.use_graph = False
    assert _same(m(x), want)
    m.use_graph = True
    caps = m.graph_stats['captures']
    tc = m.bbox_head.test_cfg
    v = tc.pop('nms_iou_thr')
    tc['nms_iou_thr_renamed'] = v               # same values, other keys: the head now reads its default 0.5
    m(x)
    assert m.graph_stats['captures'] == caps + 1, 'a test_cfg key swapped for another of equal value replayed the stale graph'
