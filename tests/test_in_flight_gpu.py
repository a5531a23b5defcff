"""GPU: `visualdet3d_amd.networks.pipelines.in_flight` -- a detector's forward as a replayable step (`CapturedStep`) and k of them overlapped on k streams
(`InFlight`: k detector objects with the same weights; bench.py's default loop with k = 2).

The point of the test is RACE freedom: two replicas run concurrently on one GPU for many steps with a DIFFERENT batch in every step, and every step's
record must hold exactly the detections a lone, synchronous `forward_device` call gives for that batch -- anything the replicas shared by accident (post-processing
scratch, packed-weight caches, static buffers) would show up as a wrong or mixed-up record."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(n):
    """n Stereo3D objects with the same (seeded) weights: the golden 96 x 320 case's configuration, whose threshold / head scale give a few detections per frame"""
    from tests.common import load_golden, stereo_case_from_golden
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3D
    from visualdet3d_amd.utils import synthetic as syn
    g = load_golden('stereo3d_r34_96x320')
    cfg, _, winit = stereo_case_from_golden(g)
    ms, sd = [], None
    for i in range(n):
        m = Stereo3D(cfg)
        if sd is None:
            sd = syn.seeded_state_dict(m.state_dict(), **winit)
        m.load_state_dict(sd)
        m = m.cuda().eval()
        m.compute_dtype = torch.bfloat16
        m.core.overlap_neck = False
        m.bbox_head.overlap_towers = False
        ms.append(m)
    return ms


def _batches(n, B, H, W):
    from visualdet3d_amd.utils import synthetic as syn
    P2, _ = syn.kitti_calib(W, batch=B)
    out = []
    for i in range(n):
        L, R = syn.stereo_pair(B, H, W, seed=40 + i)
        out.append((L.cuda(), R.cuda(), P2.cuda()))
    return out


def _direct(m, batch, k):
    """the synchronous answer: forward_device on a lone object, unpadded"""
    with torch.no_grad():
        out = m.forward_device(*batch)
        scores, boxes, labels, count = out[0], out[1], out[2], out[-1]          # (the anchor heads return the anchor indices too)
    torch.cuda.synchronize()
    res = []
    for b in range(scores.shape[0]):
        n = min(int(count[b]), k)
        res.append((scores[b, :n].cpu(), boxes[b, :n].cpu(), labels[b, :n].long().cpu()))
    return res


@pytest.mark.parametrize('k_rep', [2, 3])
def test_records_of_overlapped_steps_equal_lone_forward_calls(k_rep):
    from visualdet3d_amd.networks.pipelines.in_flight import CapturedStep, InFlight
    B, H, W, K = 2, 96, 320, 64
    ms = _models(k_rep + 1)
    ref_model, reps = ms[0], ms[1:]
    batches = _batches(7, B, H, W)
    want = [_direct(ref_model, bt, K) for bt in batches]
    assert sum(len(f[0]) for w in want for f in w) >= 7, 'the workload must produce detections'
    assert any(not torch.equal(want[0][0][0], w[0][0]) for w in want[1:] if len(w[0][0]) == len(want[0][0][0])) or len({len(w[0][0]) for w in want}) > 1, \
        'the batches must differ'
    pipe = InFlight([CapturedStep(m, batches[0], B, k=K, own_inputs=True) for m in reps])
    for rnd in range(3):                                   # the slots and replicas are reused round after round
        tickets = []
        got = {}
        for i, bt in enumerate(batches):
            tickets.append(pipe.submit(*bt))               # returns at once: up to k_rep steps are in flight
            if i >= k_rep:                                 # read a record while later steps are still running
                j = i - k_rep
                got[j] = [(s.clone(), b.clone(), l.clone()) for s, b, l in pipe.detections(tickets[j])]
        for j in range(len(batches)):
            if j not in got:
                got[j] = [(s.clone(), b.clone(), l.clone()) for s, b, l in pipe.detections(tickets[j])]
        for j, w in enumerate(want):
            for f in range(B):
                assert torch.equal(got[j][f][0], w[f][0]) and torch.equal(got[j][f][1], w[f][1]) and torch.equal(got[j][f][2], w[f][2]), (rnd, j, f)


def test_run_on_static_inputs_counts_and_slot_lifetime():
    from visualdet3d_amd.networks.pipelines.in_flight import CapturedStep, InFlight
    B, H, W, K = 2, 96, 320, 64
    ms = _models(2)
    bt = _batches(1, B, H, W)[0]
    want = _direct(ms[0], bt, K)
    pipe = InFlight([CapturedStep(m, bt, B, k=K) for m in ms])          # the caller's tensors are the static inputs (a resident batch)
    counts = pipe.run(9)
    assert counts.shape == (1, B) and [int(c) for c in counts[0]] == [len(f[0]) for f in want]
    with pytest.raises(AssertionError):
        pipe.steps[0].set_inputs(*bt)                                   # not this object's buffers
    t = pipe.submit()
    for _ in range(4):
        pipe.submit()
    with pytest.raises(AssertionError):
        pipe.collect(t)                                                 # 2 x replicas later the slot has been rewritten


@pytest.mark.parametrize('name', ['groundaware_r34_96x320', 'km3d_dla34_96x320'])
def test_monocular_detectors_in_flight(name):
    """The same API over the monocular detectors (`forward_device(img, P2)`): GroundAwareYolo3D and the KM3D keypoint detector, two replicas, a different batch per step."""
    from tests.common import load_golden, mono_case_from_golden
    from visualdet3d_amd.networks.pipelines.in_flight import CapturedStep, InFlight
    from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT
    import visualdet3d_amd.networks.detectors  # noqa: F401
    from visualdet3d_amd.utils import synthetic as syn
    g = load_golden(name)
    if name.startswith('km3d'):
        cfg = syn.km3d_cfg(output_w=320 // 4)
        winit = dict(seed=int(g['meta'][4]), head_std=float(g['head_std'])) if 'head_std' in g else dict(seed=1)
    else:
        cfg, _, winit = mono_case_from_golden(g, name)
    ms, sd = [], None
    for i in range(3):
        m = DETECTOR_DICT[cfg.name](cfg)
        if sd is None:
            sd = syn.seeded_state_dict(m.state_dict(), **winit)
        m.load_state_dict(sd)
        ms.append(m.cuda().eval())
    B, H, W, K = 2, 96, 320, 64
    P2, _ = syn.kitti_calib(W, batch=B)
    batches = [(syn.mono_image(B, H, W, seed=70 + i).cuda(), P2.cuda()) for i in range(5)]
    want = [_direct(ms[0], bt, K) for bt in batches]
    pipe = InFlight([CapturedStep(m, batches[0], B, k=K, own_inputs=True) for m in ms[1:]])
    for rnd in range(2):
        tickets = [pipe.submit(*bt) for bt in batches[:4]]                # four submits = 2 x replicas: every record still valid
        got = [[(s.clone(), b.clone(), l.clone()) for s, b, l in pipe.detections(t)] for t in tickets]
        for j in range(4):
            for f in range(B):
                assert torch.equal(got[j][f][0], want[j][f][0]) and torch.equal(got[j][f][1], want[j][f][1]) and torch.equal(got[j][f][2], want[j][f][2]), (rnd, j, f)
