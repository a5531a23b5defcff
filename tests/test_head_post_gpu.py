"""GPU: device-side get_bboxes (vd3d_head_postprocess) and vd3d_nms against the oracle on IDENTICAL inputs.
Index selection (which anchors survive mask / threshold / z-prior filter / NMS, and in which order) must be
bit-exact; decoded fields agree to fp32 round-off (expf/atan2f/sigmoid are within a few ulp of the host's)."""
import tempfile

import numpy as np
import pytest
import torch

from oracle import detector_oracle as orc
from oracle.nms_ref import nms_numpy
from visualdet3d_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def _setup(H, W, B, seed, logit_bias=-1.0, logit_std=1.0):
    from visualdet3d_amd.networks.heads.detection_3d_head import StereoHead
    tmp = tempfile.mkdtemp()
    cfg = syn.stereo3d_cfg(tmp, score_thr=0.6, nms_iou_thr=0.4)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    cfg.head.layer_cfg.num_features_in = 16   # tiny towers: only the post-processing is under test
    cfg.head.layer_cfg.reg_feature_size = 16
    cfg.head.layer_cfg.cls_feature_size = 16
    head = StereoHead(**cfg.head).cuda().eval()
    N = (H // 16) * (W // 16) * 48
    g = torch.Generator().manual_seed(seed)
    cls = torch.randn(B, N, 3, generator=g) * logit_std + logit_bias
    reg = torch.randn(B, N, 12, generator=g) * 0.7
    P2, _ = syn.kitti_calib(W, batch=B)
    P2[:, 1, 2] += torch.arange(B) * 7.0   # per-sample calibration -> per-sample ground masks
    return cfg, head, cls, reg, P2


@pytest.mark.parametrize('H,W,B,seed,std', [(96, 320, 3, 0, 1.0), (384, 1280, 2, 1, 0.8), (96, 320, 1, 2, 2.5)])
def test_head_postprocess_matches_oracle(H, W, B, seed, std):
    cfg, head, cls, reg, P2 = _setup(H, W, B, seed, logit_std=std)
    mean_npy, std_npy = orc.load_priors(cfg.head.preprocessed_path, cfg.obj_types)
    anchors, means, mean_std = orc.anchors_for_image(H, W, cfg.head.anchors_cfg, mean_npy, std_npy)
    mask = orc.anchor_mask(anchors, means, P2)
    padded = head.get_bboxes_batched(cls.cuda(), reg.cuda(), P2.cuda(), (H, W))
    torch.cuda.synchronize()
    scores, boxes, labels, aidx, count = [t.cpu() for t in padded]
    total = 0
    for b in range(B):
        s, bx, l, idx = orc.get_bboxes(cls[b], reg[b], anchors, mean_std, mask[b], (H, W), 2, 0.6, 0.4)
        k = int(count[b])
        assert k == len(s), (b, k, len(s))
        assert torch.equal(aidx[b, :k].long(), idx), 'sample %d: anchor selection / order differs' % b
        assert torch.equal(labels[b, :k].long(), l)
        assert torch.allclose(scores[b, :k], s, rtol=1e-5, atol=1e-6)
        sc = bx.abs().amax(dim=0).clamp_min(1.0) if k else 1.0
        assert k == 0 or ((boxes[b, :k] - bx).abs() / sc).max().item() < 1e-5
        total += k
    assert total >= 5, 'test inputs produced too few detections to be meaningful'


@pytest.mark.parametrize('max_cand', [256, 2048, 4096, 8192])
def test_head_postprocess_every_nms_path_same_selection(max_cand):
    """The three NMS routes of vd3d_head_postprocess -- one wave per frame (<= 256 candidates), the block path with its boxes in LDS
    (capacity <= 4096) and the block path reading them from the workspace (8192) -- against the oracle on inputs sized for each:
    ~60 candidates for the wave path, 1 300 - 1 500 (twenty-odd chunks of 64, most of them with survivors AND suppressions) for the others."""
    H, W, B = (96, 320, 2) if max_cand == 256 else (192, 640, 2)
    cfg, head, cls, reg, P2 = _setup(H, W, B, 11, logit_bias=-2.2 if max_cand == 256 else -0.8, logit_std=1.2)
    head.max_candidates = max_cand
    mean_npy, std_npy = orc.load_priors(cfg.head.preprocessed_path, cfg.obj_types)
    anchors, means, mean_std = orc.anchors_for_image(H, W, cfg.head.anchors_cfg, mean_npy, std_npy)
    mask = orc.anchor_mask(anchors, means, P2)
    padded = head.get_bboxes_batched(cls.cuda(), reg.cuda(), P2.cuda(), (H, W))
    torch.cuda.synchronize()
    scores, boxes, labels, aidx, count = [t.cpu() for t in padded]
    for b in range(B):
        s, bx, l, idx = orc.get_bboxes(cls[b], reg[b], anchors, mean_std, mask[b], (H, W), 2, 0.6, 0.4)
        ncand = int(((torch.sigmoid(cls[b][:, :2]).amax(dim=1) > 0.6) & mask[b]).sum())
        if max_cand == 256:
            assert 20 < ncand <= 256, ncand
        else:
            assert 1000 < ncand <= 2048, ncand
        k = int(count[b])
        assert k == len(s) and k >= 5, (b, k, len(s))
        assert torch.equal(aidx[b, :k].long(), idx), 'sample %d: anchor selection / order differs' % b
        assert torch.equal(labels[b, :k].long(), l)


def test_head_candidate_overflow_is_reported():
    cfg, head, cls, reg, P2 = _setup(96, 320, 1, 3, logit_bias=3.0)
    head.max_candidates = 64
    padded = head.get_bboxes_batched(cls.cuda(), reg.cuda(), P2.cuda(), (96, 320))
    with pytest.raises(RuntimeError):
        head.unpad(padded)                     # the bare padded arrays carry no logits to retry with: the overflow is an error HERE ...
    assert int(padded[4][0]) == -1


@pytest.mark.parametrize('H,W,bias,std,lo,hi', [(192, 640, 0.9, 1.0, 4096, 8192),        # one doubling: 8192, still the LDS path
                                                (384, 1280, 0.0, 1.0, 8192, 16384),      # config 2's frame size; capacity 16 384: the global-memory path
                                                (384, 1280, 1.5, 1.0, 16384, 32768)])    # nearly every anchor the ground filter lets through
def test_more_candidates_than_the_captured_capacity_equal_the_oracle(H, W, bias, std, lo, hi):
    """The reference's candidate list has no cap (detection_3d_head.py:341-400: boolean indexing, then nms).  A frame with more candidates than
    `max_candidates` (4096, the capacity of the captured launch) is re-run with doubled capacities (get_bboxes_unbounded; beyond 8192 the lists are
    sorted in global memory): same selection, same order, same labels as the oracle -- through the reference-signature `get_bboxes` and through `unpad(retry=)`."""
    B = 2
    cfg, head, cls, reg, P2 = _setup(H, W, B, 21, logit_bias=bias, logit_std=std)
    cls[1] -= bias + 1.0                             # frame 1: the ordinary path in the same call (_setup's default level)
    mean_npy, std_npy = orc.load_priors(cfg.head.preprocessed_path, cfg.obj_types)
    anchors, means, mean_std = orc.anchors_for_image(H, W, cfg.head.anchors_cfg, mean_npy, std_npy)
    mask = orc.anchor_mask(anchors, means, P2)
    ncand = int(((torch.sigmoid(cls[0][:, :2]).amax(dim=1) > 0.6) & mask[0]).sum())
    assert lo < ncand <= hi, ncand
    clsd, regd, P2d = cls.cuda(), reg.cuda(), P2.cuda()
    padded = head.get_bboxes_batched(clsd, regd, P2d, (H, W))
    assert padded[4].tolist()[0] == -1 and padded[4].tolist()[1] >= 0
    outs = head.unpad(padded, retry=lambda b: head.get_bboxes_unbounded(clsd[b:b + 1], regd[b:b + 1], P2d[b:b + 1], (H, W)))
    img = torch.zeros(1, 3, H, W)
    head.get_anchor(img.cuda(), P2d[0:1])
    ref_sig = head.get_bboxes(clsd[0:1], regd[0:1], None, P2d[0:1], img.cuda())
    for b in range(B):
        s, bx, l, idx = orc.get_bboxes(cls[b], reg[b], anchors, mean_std, mask[b], (H, W), 2, 0.6, 0.4)
        for got in ([outs[b]] + ([ref_sig] if b == 0 else [])):
            gs, gb, gl = [t.cpu() for t in got]
            assert len(gs) == len(s) and len(s) >= 1, (b, len(gs), len(s))
            assert torch.equal(gl, l)
            assert torch.equal(gs, s) or torch.allclose(gs, s, rtol=1e-5, atol=1e-6)
            sc = bx.abs().amax(dim=0).clamp_min(1.0)
            assert ((gb - bx).abs() / sc).max().item() < 1e-5


@pytest.mark.parametrize('n,seed', [(0, 0), (1, 1), (37, 2), (64, 3), (65, 4), (1000, 5), (5000, 6)])
def test_nms_bit_exact(n, seed):
    from visualdet3d_amd import hip_ops as ops
    rng = np.random.default_rng(seed)
    ctr = rng.uniform(0, 300, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 80, (n, 2)).astype(np.float32)
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    if n > 10:
        scores[5] = scores[3]          # exact ties -> lower index first
        boxes[7] = boxes[2]            # duplicate boxes
        boxes[9, 2:] = boxes[9, :2]    # zero-area box (0/0 -> NaN IoU, never suppresses)
    want = nms_numpy(boxes, scores, 0.45)
    got = ops.nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.45).cpu().numpy()
    assert np.array_equal(got, want)


def _nms_case(kind, n, rng):
    if kind == 'identical':            # one survivor: box 0 suppresses every other box of every chunk
        boxes = np.tile(np.array([[10, 10, 60, 50]], np.float32), (n, 1))
    elif kind == 'disjoint':           # nothing suppresses anything: every chunk's 64 x 64 matrix is empty
        i = np.arange(n, dtype=np.float32)
        boxes = np.stack([20 * (i % 97), 20 * (i // 97), 20 * (i % 97) + 10, 20 * (i // 97) + 10], 1).astype(np.float32)
    elif kind == 'chain':              # a sliding row in score order: i suppresses i+1 (IoU 0.82), i+1 is dead so i+2 (IoU 0.67) is tested
        i = np.arange(n, dtype=np.float32)      # against i only, ... -- the greedy order INSIDE a chunk and across chunk borders decides
        boxes = np.stack([i * 2.0, np.zeros(n, np.float32), i * 2.0 + 20, np.full(n, 10, np.float32)], 1).astype(np.float32)
    elif kind == 'clusters':           # ~n/40 clusters of heavily overlapping boxes, scores interleaved across clusters
        c = rng.integers(0, max(1, n // 40), n)
        ctr = np.stack([60.0 * (c % 25), 60.0 * (c // 25)], 1) + rng.uniform(-6, 6, (n, 2))
        wh = rng.uniform(30, 44, (n, 2))
        boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    else:
        raise KeyError(kind)
    if kind == 'chain':
        scores = np.linspace(1.0, 0.1, n).astype(np.float32)       # score order = position order
    else:
        scores = rng.uniform(0, 1, n).astype(np.float32)
    return boxes, scores


@pytest.mark.parametrize('kind', ['identical', 'disjoint', 'chain', 'clusters'])
@pytest.mark.parametrize('n', [63, 64, 65, 128, 129, 1025, 4096, 4097])
def test_nms_structured_cases_bit_exact(kind, n):
    """vd3d_nms against torchvision's algorithm (oracle/nms_ref.py) on inputs built to stress the chunked greedy pass: sizes on either side of the
    64-box chunk and of the 4096-box capacity steps; a lone survivor, no suppression at all, a chain whose outcome depends on the greedy
    order inside a chunk and across chunk borders, and dense clusters."""
    from visualdet3d_amd import hip_ops as ops
    rng = np.random.default_rng(n * 7 + len(kind))
    boxes, scores = _nms_case(kind, n, rng)
    for thr in (0.45, 0.7):
        want = nms_numpy(boxes, scores, thr)
        got = ops.nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr).cpu().numpy()
        assert np.array_equal(got, want), (kind, n, thr, len(got), len(want))
    if kind == 'identical':
        assert len(want) == 1
    if kind == 'disjoint':
        assert len(want) == n
    if kind == 'chain':
        assert 1 < len(want) < n


def test_get_bboxes_reference_signature_no_clip_and_cls_agnostic_flag():
    """Reference-signature ``get_bboxes`` (detection_3d_head.py:341): ``img_batch=None`` skips ClipBoxes (:375-376) -- compared
    with the oracle's no-clip branch; the unsupported per-class NMS flag fails loudly."""
    H, W = 96, 320
    cfg, head, cls, reg, P2 = _setup(H, W, 1, 7, logit_std=1.5)
    mean_npy, std_npy = orc.load_priors(cfg.head.preprocessed_path, cfg.obj_types)
    anchors, means, mean_std = orc.anchors_for_image(H, W, cfg.head.anchors_cfg, mean_npy, std_npy)
    mask = orc.anchor_mask(anchors, means, P2)
    img = torch.zeros(1, 3, H, W).cuda()
    with pytest.raises(RuntimeError):
        head.get_bboxes(cls.cuda(), reg.cuda(), None, P2.cuda(), img_batch=None)       # shape unknown yet
    a = head.get_anchor(img, P2.cuda())
    s, b, l = [t.cpu() for t in head.get_bboxes(cls.cuda(), reg.cuda(), a, P2.cuda(), img_batch=None)]
    ws, wb, wl, _ = orc.get_bboxes(cls[0], reg[0], anchors, mean_std, mask[0], (H, W), 2, 0.6, 0.4, clip=False)
    assert len(s) == len(ws) and len(s) >= 3 and torch.equal(l, wl) and torch.allclose(s, ws, rtol=1e-5, atol=1e-6)
    assert ((b - wb).abs() / wb.abs().amax(dim=0).clamp_min(1.0)).max().item() < 1e-5
    assert bool((b[:, 0] < 0).any() or (b[:, 1] < 0).any() or (b[:, 2] > W).any() or (b[:, 3] > H).any()), 'case must have a box the clamp would move'
    sc, bc, lc = [t.cpu() for t in head.get_bboxes(cls.cuda(), reg.cuda(), a, P2.cuda(), img_batch=img)]
    assert bool((bc[:, 2] <= W).all()) and bool((bc[:, 3] <= H).all()) and bool((bc[:, :2] >= 0).all())
    head.test_cfg.cls_agnositc = False
    with pytest.raises(NotImplementedError):
        head.get_bboxes(cls.cuda(), reg.cuda(), a, P2.cuda(), img_batch=img)


def test_pack_detections_record_wider_than_capacity_and_nan_padding():
    """vd3d_pack_detections with K < k (KM3D's decode returns K = 100 rows; the gather record has 128) and NaN in the padding
    rows (the result buffers are torch.empty): rows past the count and past K are exactly zero, the count rides in row k."""
    from visualdet3d_amd import hip_ops as ops
    B, K, k = 3, 100, 128
    g = torch.Generator().manual_seed(3)
    scores = torch.rand(B, K, generator=g)
    boxes = torch.randn(B, K, 11, generator=g)
    labels = torch.randint(0, 3, (B, K), generator=g, dtype=torch.int32)
    count = torch.tensor([100, 0, 37], dtype=torch.int32)
    for b in range(B):
        scores[b, int(count[b]):] = float('nan')
        boxes[b, int(count[b]):] = float('inf')
    out = ops.pack_detections(scores.cuda(), boxes.cuda(), labels.cuda(), count.cuda(), k).cpu()
    assert out.shape == (B, k + 1, 13) and bool(torch.isfinite(out).all())
    for b in range(B):
        n = int(count[b])
        assert torch.equal(out[b, :n, 0], scores[b, :n]) and torch.equal(out[b, :n, 1:12], boxes[b, :n])
        assert torch.equal(out[b, :n, 12], labels[b, :n].float())
        assert bool((out[b, n:k] == 0).all()) and out[b, k, 0] == n and bool((out[b, k, 1:] == 0).all())
    from visualdet3d_amd.distributed import DetectionGather
    dg = DetectionGather(B, k, 'cuda', world=1)
    dg.fill(scores.cuda(), boxes.cuda(), labels.cuda(), count.cuda())
    assert torch.equal(dg.pack.cpu(), out)
