"""Shared helpers for the parity tests."""
import os
import tempfile

import numpy as np
import torch

from visualdet3d_amd.utils import synthetic as syn

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + '.npz')))


def subsample(t, n=4096):
    flat = t.reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n].clone()


def stereo_case_from_golden(g):
    depth, H, W, frames, wseed, iseed = [int(v) for v in g['meta']]
    tmp = tempfile.mkdtemp()
    cfg = syn.stereo3d_cfg(tmp, depth=depth, score_thr=float(g['score_thr']))
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    L, R = syn.stereo_pair(frames, H, W, seed=iseed)
    P2, P3 = syn.kitti_calib(W, batch=frames)
    return cfg, (L, R, P2, P3), dict(seed=wseed, head_std=float(g['head_std']))


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def assert_detections_close(got, want, rtol=1e-3, what='', allow_missing=0, loose_fields=None):
    """(scores, boxes, labels) vs (scores, boxes, labels).

    Detections come out in decreasing-score order; two detections whose scores differ by less than the fp32 noise
    of a ~60-layer network may legitimately swap places, so the comparison is a one-to-one MATCHING (same label,
    every box field and the score within ``rtol`` of the field's scale) rather than a positional one.
    ``allow_missing``: number of detections that may be unmatched on either side (threshold-flip hazard,
    SURVEY.md 7.3 item 7) -- 0 for fp32 comparisons."""
    gs, gb, gl = [torch.as_tensor(np.asarray(x)) for x in got]
    ws, wb, wl = [torch.as_tensor(np.asarray(x)) for x in want]
    gl, wl = gl.long().reshape(-1), wl.long().reshape(-1)
    assert abs(gs.numel() - ws.numel()) <= allow_missing, '%s: %d detections, expected %d' % (what, gs.numel(), ws.numel())
    if ws.numel() == 0 or gs.numel() == 0:
        return
    scale = torch.cat([wb.abs().double().amax(dim=0).clamp_min(1.0), torch.ones(1, dtype=torch.double)])
    G = torch.cat([gb.double(), gs.double().reshape(-1, 1)], dim=1) / scale
    Wt = torch.cat([wb.double(), ws.double().reshape(-1, 1)], dim=1) / scale
    diff = (G[None, :, :] - Wt[:, None, :]).abs()
    if loose_fields:
        # {box field: (atol, period)}: fields decided by a discrete search (post_optimization's hill-climbed yaw, steps >= 0.0125 rad)
        # are matched with their own absolute tolerance (modulo `period`), not with rtol
        for fld, (atol, period) in loose_fields.items():
            d = (gb[None, :, fld].double() - wb[:, None, fld].double()).abs()
            if period:
                d = torch.minimum(d % period, period - d % period)
            diff[:, :, fld] = torch.where(d <= atol, torch.zeros_like(d), torch.full_like(d, float('inf')))
    err = diff.amax(dim=2)          # [want, got]
    err[wl[:, None] != gl[None, :]] = float('inf')
    best, arg = err.min(dim=1)
    ok = best <= rtol
    matched = arg[ok]
    assert matched.unique().numel() == matched.numel(), what + ': two expected detections matched the same result'
    missing = int((~ok).sum())
    extra = gs.numel() - matched.numel()
    assert missing <= allow_missing and extra <= allow_missing, \
        '%s: %d expected detections unmatched (worst err %.3e), %d unexpected (rtol %.1e)' % (
            what, missing, float(best[~ok].min()) if missing else 0.0, extra, rtol)
    # order must still be non-increasing in score
    assert bool((gs[1:] <= gs[:-1]).all()), what + ': scores not in decreasing order'


def mono_case_from_golden(g, name):
    depth, H, W, frames, wseed, iseed = [int(v) for v in g['meta']]
    tmp = tempfile.mkdtemp()
    det = 'GroundAwareYolo3D' if name.startswith('groundaware') else 'Yolo3D'
    cfg = syn.mono3d_cfg(tmp, depth=depth, score_thr=float(g['score_thr']), name=det,
                         post_optimization=bool(int(g['post_optimization'])) if 'post_optimization' in g else False)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 2)
    img = syn.mono_image(frames, H, W, seed=iseed)
    P2, _ = syn.kitti_calib(W, batch=frames)
    return cfg, (img, P2), dict(seed=wseed, head_std=float(g['head_std']))


def matched_fraction(got, want, rtol):
    """Greedy ONE-TO-ONE matching of detections (same label, every box field and the score within ``rtol`` of the field's scale),
    closest pairs first: the fraction of expected detections that found a partner.  For the reduced-precision end-to-end
    comparisons of the keypoint detector, whose ~100 detections per frame contain near-duplicates (a positional matching with a
    loose tolerance would pair two expected detections with the same result)."""
    gs, gb, gl = [torch.as_tensor(np.asarray(x)) for x in got]
    ws, wb, wl = [torch.as_tensor(np.asarray(x)) for x in want]
    gl, wl = gl.long().reshape(-1), wl.long().reshape(-1)
    if ws.numel() == 0:
        return 1.0
    if gs.numel() == 0:
        return 0.0
    scale = torch.cat([wb.abs().double().amax(dim=0).clamp_min(1.0), torch.ones(1, dtype=torch.double)])
    G = torch.cat([gb.double(), gs.double().reshape(-1, 1)], dim=1) / scale
    Wt = torch.cat([wb.double(), ws.double().reshape(-1, 1)], dim=1) / scale
    err = (G[None, :, :] - Wt[:, None, :]).abs().amax(dim=2)          # [want, got]
    err[wl[:, None] != gl[None, :]] = float('inf')
    pairs = sorted((float(err[i, j]), i, j) for i in range(err.shape[0]) for j in range(err.shape[1]) if err[i, j] <= rtol)
    used_w, used_g = set(), set()
    for e, i, j in pairs:
        if i not in used_w and j not in used_g:
            used_w.add(i)
            used_g.add(j)
    return len(used_w) / float(err.shape[0])
