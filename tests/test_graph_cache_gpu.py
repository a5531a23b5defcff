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
