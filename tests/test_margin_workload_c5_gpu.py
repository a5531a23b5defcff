"""GPU: BASELINE config 5 in its TIMED type at its TIMED size -- KM3D DLA-34, 16 images of 512 x 1760, fp16, the whole HIP network end to
end -- held to the LITERAL bar on a margin-controlled workload (what tests/test_margin_workload_gpu.py does for configs 2 and 3):

    the detection set of every frame is identical to the fp16-rounded oracle's (same heat-map peaks, same classes, same NMS survivors,
    same order) and every score / box field is within 1e-3 of its scale
    (reference semantics: heads/km3d_head.py:155-314, networks/utils/rtm3d_utils.py:122-127,195-228,314-455).

Why a special workload.  With bench.py's seeded random weights two correct fp16 evaluations of this network leave each other by 1-2 % RMS
on the head maps: the sixteen stacked DCNv2 blocks sample at positions computed by RANDOM offset convs (~1 px rms), so every 1-ulp flip
upstream moves sampling positions downstream -- an artefact of the random initialisation (the reference zero-initialises `conv_offset`,
lib/ops/dcn/deform_conv.py:453-457, and trained offsets are smooth).  And a random heat-map head puts thousands of near-tied local maxima
next to the threshold -- and so does ANY linear head on the features of a noise image: the local maxima of a projection of homogeneous
texture crowd at every level (measured: the widest gap between consecutive maxima is 1-2 x the implementation noise).  A real frame is not
homogeneous, it has objects.  Here the network and its backbone / up-path weights stay config 5's; controlled are

  * the INPUT: config 5's noise image at a quarter of its amplitude plus twelve Gaussian blobs per frame (random colour, sigma 10-16 px,
    amplitude 2-4): a scene with objects, so that a heat map can have isolated peaks at all;

  * the offset convs of the 16 DCNv2 blocks: scaled so that offsets are SUB-PIXEL (printed; asserted < 1 px);
  * the LAST (1 x 1) conv of the heat-map head: ONE filter -- the regularised Fisher discriminant between the oracle's hidden heat-map
    features at the blob centres and everywhere else -- drives class 0 (gain g) and class 1 (gain g / 2, higher level: its peaks sit on
    class-0 peaks, identical boxes, and are suppressed by the class-agnostic NMS -- the two logits are affine functions of the same
    response, so the order is fixed by construction); levels are chosen in the widest gaps of the sorted LOCAL MAXIMA such that every
    decision -- above / below the threshold, is / is not a 3 x 3 local maximum -- has a margin.  The design is noise-aware: a TWIN run of
    the oracle on the image perturbed by 1e-6 (every rounding downstream may flip, like between two implementations) gives a proxy of
    the noise, and the (regulariser, level) with the best margin-to-proxy ratio is taken.  The margins are then ASSERTED against the
    noise the test measures between the HIP path and the oracle (> 10 x).  Class 2 sits at sigmoid(-9);
  * the keypoint heat map `hm_hp` sits at sigmoid(-9) < 0.1: no keypoint is snapped to a heat-map peak (km3d_head.py:225-244 keeps the
    regressed keypoints) -- the association thresholds are the one part of the decode NOT under this statement (they are covered on
    identical maps by tests/test_km3d_gpu.py::test_decode_matches_oracle_on_identical_maps);
  * the last convs of the regression heads (wh, hps, rot, dim, reg): the seeded random conv x 1/16 plus a bias that describes ONE
    plausible car (2-D box 96 x 56 px, the nine key points of a 1.6 x 1.5 x 3.9 m box 20 m away, the rotation bins away from their
    decision `rot[1] > rot[5]`): the 16 x 3 least squares of `gen_position` is then well conditioned everywhere (with near-coincident
    random key points it is singular -- the reference adds a random 1e-8 jitter for exactly that reason, rtm3d_utils.py:437).

The fp16-rounded oracle runs TWO frames on the host; the HIP network runs the bench's batch of 16 (the two frames x 8 replicas: the at-size
dispatch -- level pair, persistent fused head, 64 -> 64 DCN at 128 x 440 -- is what runs) and EVERY replica must reproduce the oracle."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import detector_oracle as orc

pytestmark = pytest.mark.gpu

MARGIN = 10.0
THR = 0.3                                   # score threshold of the config (synthetic.km3d_cfg)
TOP_LOGIT = 9.0                             # the strongest designed peak (sigmoid stays strictly monotone in fp32 below ~15)


def _nb_max(r):
    """max over the 8 neighbours (centre excluded; outside the map = -inf, like max_pool2d's implicit padding) of r [N,H,W]"""
    p = F.pad(r, (1, 1, 1, 1), value=float('-inf'))
    H, W = r.shape[1:]
    best = None
    for dy in range(3):
        for dx in range(3):
            if dy == 1 and dx == 1:
                continue
            s = p[:, dy:dy + H, dx:dx + W]
            best = s if best is None else torch.maximum(best, s)
    return best


def _thr_logit():
    return math.log(THR / (1 - THR))


def scene_images(N, H, W, blobs=12, seed=0):
    """config 5's noise image at a quarter of its amplitude + `blobs` Gaussian blobs per frame.  -> (images [N,3,H,W], blob centres [(n, y, x)])"""
    import numpy as np

    from visualdet3d_amd.utils import synthetic as syn
    rng = np.random.default_rng(seed)
    img = 0.25 * syn.mono_image(N, H, W, seed=3).numpy()
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    pos = []
    for n in range(N):
        for _ in range(blobs):
            while True:
                y, x = rng.uniform(60, H - 60), rng.uniform(80, W - 80)
                if all((abs(y - py) > 90 or abs(x - px) > 130) for (nn, py, px) in pos if nn == n):
                    break
            pos.append((n, y, x))
            amp, sig = rng.uniform(2.0, 4.0), rng.uniform(10, 16)
            col = rng.normal(0, 1, 3)
            col /= np.linalg.norm(col)
            img[n] += col[:, None, None].astype(np.float32) * (amp * np.exp(-((yy - y) ** 2 + (xx - x) ** 2) / (2 * sig * sig))).astype(np.float32)
    return torch.from_numpy(img.astype(np.float32)), pos


def _levels(ra, nu, lo_cnt, hi_cnt):
    """every candidate level (between consecutive sorted local maxima, lo_cnt..hi_cnt of them above) with its margin-to-noise ratio:
    -> list of (ratio, count, level, gap, sep), best first.  sep = the smallest |r(p) - best neighbour of p| over the positions within
    half a gap of the level or above it."""
    nb = _nb_max(ra)
    peaks = torch.sort(ra[ra > nb], descending=True).values
    out = []
    for n in range(lo_cnt, min(hi_cnt, len(peaks) - 1) + 1):
        gap, lvl = (peaks[n - 1] - peaks[n]).item(), 0.5 * (peaks[n - 1] + peaks[n]).item()
        sep = (ra - nb).abs()[ra > lvl - 0.5 * gap].min().item()
        out.append((min(0.5 * gap, sep) / nu, n, lvl, gap, sep))
    out.sort(reverse=True)
    return out, peaks[0].item()


def design_hm_head(hid, hid_twin, centres, n_cls, lo_cnt, hi_cnt):
    """hid / hid_twin [N,C,H,W]: the oracle's hidden features of the heat-map branch and of its perturbed twin; centres: feature-map
    indices (n, y, x) of the blobs.  -> (weight [n_cls, C, 1, 1], bias [n_cls], info)."""
    N, C, H, W = hid.shape
    X = hid.permute(0, 2, 3, 1).reshape(-1, C).double()
    mu = X.mean(0)
    V = X - mu
    cov = V.t() @ V / V.shape[0]
    ev_max = torch.linalg.eigvalsh(cov)[-1].item()
    T = X[[n * H * W + y * W + x for n, y, x in centres]] - mu
    best = None
    for grade in (0.0, 1.0, 3.0):                                            # graded target amplitudes (the targets should not tie) x regulariser
        a = torch.linspace(1.0, 1.0 + grade, T.shape[0], dtype=torch.double)
        for lam in (1e-1, 3e-2, 1e-2, 3e-3, 1e-3, 3e-4):
            u = torch.linalg.solve(cov + lam * ev_max * torch.eye(C, dtype=torch.double), (a[:, None] * T).sum(0) / a.sum()).float()
            u = u / u.norm()
            ra = torch.einsum('nchw,c->nhw', hid, u)
            nu = (ra - torch.einsum('nchw,c->nhw', hid_twin, u)).abs().max().item()
            lv, top = _levels(ra, nu, 3, hi_cnt)
            for l0 in [l for l in lv if l[1] >= lo_cnt][:6]:                 # class 0's level; class 1: the best level with fewer peaks above it
                lv1 = [l for l in lv if 3 <= l[1] <= l0[1] - 2]
                if not lv1:
                    continue
                fom = min(l0[0], lv1[0][0])
                if best is None or fom > best['ratio']:
                    best = dict(ratio=fom, lam=lam, grade=grade, u=u, nu=nu, top=top, l0=l0, l1=lv1[0])
    assert best is not None, 'no usable design'
    thr_l = _thr_logit()
    # the gain maps the STRONGEST peak to TOP_LOGIT (beyond ~15 the fp32 sigmoid stops being strictly monotone and neighbours tie); every
    # margin-to-noise ratio is independent of the gain (logit noise scales with it)
    # ... except the SCORE error p (1 - p) g nu <= g nu / 4, which the literal bar holds to 1e-3: the gain is capped by the proxy noise
    g = min((TOP_LOGIT - thr_l) / (best['top'] - best['l0'][2]), 2.5e-3 / best['nu'])
    g = 2.0 ** math.floor(math.log2(g))                                      # a power of two: scales u's fp16 mantissas exactly
    w = torch.zeros(n_cls, C)
    b = torch.full((n_cls,), -9.0)
    w[0], b[0] = g * best['u'], thr_l - g * best['l0'][2]
    w[1], b[1] = 0.5 * g * best['u'], thr_l - 0.5 * g * best['l1'][2]
    best['gain'] = g
    return w.view(n_cls, C, 1, 1), b, best


def _iou_matrix(bx):
    x1, y1 = torch.max(bx[:, None, 0], bx[None, :, 0]), torch.max(bx[:, None, 1], bx[None, :, 1])
    x2, y2 = torch.min(bx[:, None, 2], bx[None, :, 2]), torch.min(bx[:, None, 3], bx[None, :, 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    area = (bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1])
    return inter / (area[:, None] + area[None, :] - inter)


def nms_stability(bx, lg, iou_thr, score_unit, iou_unit=0.04):
    """Greedy NMS on (boxes, logits) and how far its OUTCOME is from changing under perturbation.  The kept set is unchanged as long as (a) no two
    kept boxes come within reach of the threshold (neither can suppress the other, in any order) and (b) every suppressed box has a KEPT
    suppressor that overlaps it clearly and outscores it clearly (it is then removed before it could suppress anything).  -> (kept indices,
    stability in units: min over (a) (thr - IoU) / iou_unit, over (b) the best suppressor's min((IoU - thr) / iou_unit, score gap / score_unit))."""
    n = len(lg)
    iou = _iou_matrix(bx)
    order = torch.argsort(lg, descending=True).tolist()
    kept, dead = [], set()
    for i in order:
        if i in dead:
            continue
        kept.append(i)
        for j in order:
            if j != i and j not in dead and j not in kept and iou[i, j] > iou_thr:
                dead.add(j)
    stab = float('inf')
    for a in range(len(kept)):
        for b in range(a + 1, len(kept)):
            stab = min(stab, (iou_thr - iou[kept[a], kept[b]].item()) / iou_unit)
    for j in range(n):
        if j in kept:
            continue
        stab = min(stab, max(min((iou[k, j].item() - iou_thr) / iou_unit, (lg[k] - lg[j]).item() / score_unit) for k in kept))
    return kept, stab


def canonical_car(P2, stride=4.0):
    """biases of the regression heads: ONE plausible car, consistent with gen_position's own corner model (rtm3d_utils.py:340-420 as restated in
    oracle km3d_gen_position): dims (w, h, l) = (1.6, 1.5, 3.9) m, centre 20 m away on the optical axis 1 m below it, observation angle from the
    rotation bins below.  -> dict of bias vectors (stride-4 feature pixels for wh / hps)."""
    f, cx, cy = float(P2[0, 0]), float(P2[0, 2]), float(P2[1, 2])
    w_, h_, l_ = 1.6, 1.5, 3.9
    rot = torch.tensor([0.0, 1.0, 0.5, 1.0, 0.0, -1.0, 0.5, 1.0])           # bin 1 wins by 2.0; alpha = atan(0.5) - pi/2
    ry = math.atan(0.5) - 0.5 * math.pi                                      # rot_y on the optical axis
    X, Y, Z = 0.0, 1.0, 20.0
    co, sn = math.cos(ry), math.sin(ry)
    lc, ls, wc, ws, hh = l_ * 0.5 * co, l_ * 0.5 * sn, w_ * 0.5 * co, w_ * 0.5 * sn, h_ * 0.5
    Bx = [-lc - ws, -lc + ws, -lc + ws, lc + ws, lc + ws, lc - ws, lc - ws, -lc - ws]
    By = [-hh, -hh, hh, hh, -hh, -hh, hh, hh]
    Cz = [ls - wc, ls + wc, ls + wc, -ls + wc, -ls + wc, -ls - wc, -ls - wc, ls - wc]
    u0, v0 = f * X / Z + cx, f * Y / Z + cy
    hps = []
    for i in range(8):
        hps += [(f * (X + Bx[i]) / (Z + Cz[i]) + cx - u0) / stride, (f * (Y + By[i]) / (Z + Cz[i]) + cy - v0) / stride]
    hps += [0.0, 0.0]                                                        # ninth key point: the centre
    return dict(wh=torch.tensor([24.0, 14.0]), hps=torch.tensor(hps), rot=rot, dim=torch.tensor([w_, h_, l_]), reg=torch.tensor([0.3, 0.6]))


def controlled_state_dict(sd, offset_scale):
    """config 5's seeded weights with the DCN offset convs scaled to sub-pixel offsets"""
    out = dict(sd)
    n = 0
    for k in sd:
        if '.conv_offset.' in k:
            out[k] = sd[k] * offset_scale
            n += 1
    assert n == 32, n                                                        # 16 blocks x (weight, bias)
    return out


_PASS1 = {}
LIVE_JOINTS = (8, 0, 1)         # key point heat-map channels with live peaks in the `live_hp` variant: the centre point (snaps), a near corner (regressed ~9.5 px off: snaps or not by the box size), a far corner (~24 px: refused by the distance term)


def build_case(N=2, H=512, W=1760, offset_scale=1.0 / 256, live_hp=False):
    """-> everything the test (and the CPU dry run of the design) needs; the oracle runs here (twice: the frames and their perturbed twin).
    ``live_hp``: the keypoint heat map carries live peaks (see test_config5_..._live_keypoint_heat_map)."""
    from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT
    import visualdet3d_amd.networks.detectors  # noqa: F401
    from visualdet3d_amd.utils import synthetic as syn
    cfg = syn.km3d_cfg(output_w=W // 4)
    m = DETECTOR_DICT[cfg.name](cfg)
    # bench.py time_other_config(C5)'s weights; offset convs x 1/256: the seeded ones give 49 px rms over their 64 ... 512 input channels
    sd = controlled_state_dict(syn.seeded_state_dict(m.state_dict(), seed=1, head_std=0.0005), offset_scale)
    img, blob_pos = scene_images(N, H, W)
    P2, _ = syn.kitti_calib(W, batch=N)
    torch.set_num_threads(min(64, torch.get_num_threads()))
    p = 'bbox_head.head_layers.'

    def run(im):
        taps = []
        with torch.no_grad():
            orc.km3d_forward(sd, cfg, im, P2, rnd=orc.fp16_round, stage_taps=taps)
        return {t['key']: t for t in taps if t['kind'] == 'conv'}, [t['logits'][:, :18] for t in taps if t['kind'] == 'dcn']

    # pass 1: the oracle up to the hidden features of the nine head branches (their LAST convs do not feed back), and its twin
    key1 = (N, H, W, offset_scale)
    if key1 not in _PASS1:                                                   # (shared by the two tests of this file: two minutes of host time)
        _PASS1[key1] = (run(img), run(img * (1 + 1e-6 * torch.randn(img.shape, generator=torch.Generator().manual_seed(5)))))
    (convs, offs), (twin, _) = _PASS1[key1]
    off_rms = max(o.pow(2).mean().sqrt().item() for o in offs)
    off_max = max(o.abs().max().item() for o in offs)
    hid = convs[p + 'hm.2']['x']
    n_cls = len(cfg.obj_types)
    centres = [(n, int(round(y / 4)), int(round(x / 4))) for n, y, x in blob_pos]
    w_hm, b_hm, design = design_hm_head(hid, twin[p + 'hm.2']['x'], centres, n_cls, 4 * N, 20 * N)
    car = canonical_car(P2[0])
    sd2 = dict(sd)
    sd2[p + 'hm.2.weight'], sd2[p + 'hm.2.bias'] = w_hm, b_hm
    # the 2-D box size (bias of `wh`): a blob carries several local maxima a few pixels apart, the class-agnostic NMS decides between them
    # -- of the sizes tried, the one that keeps every pairwise IoU farthest from the NMS threshold
    with torch.no_grad():
        hm = F.conv2d(hid, orc.fp16_round(w_hm), b_hm)
        wh0 = F.conv2d(convs[p + 'wh.2']['x'], orc.fp16_round(sd[p + 'wh.2.weight'] * 0.0625))
        rg0 = F.conv2d(convs[p + 'reg.2']['x'], orc.fp16_round(sd[p + 'reg.2.weight'] * 0.0625)) + car['reg'].view(1, 2, 1, 1)
    pk = torch.nonzero((hm[:, :2] > _nb_max(hm[:, :2].reshape(-1, *hm.shape[2:])).view(N, 2, *hm.shape[2:])) & (hm[:, :2] > _thr_logit()))   # (frame, class, y, x)
    lg = hm[pk[:, 0], pk[:, 1], pk[:, 2], pk[:, 3]]
    cx = pk[:, 3].float() + rg0[pk[:, 0], 0, pk[:, 2], pk[:, 3]]
    cy = pk[:, 2].float() + rg0[pk[:, 0], 1, pk[:, 2], pk[:, 3]]
    best_wh = None
    for bw in range(8, 41, 2):
        for bh in range(6, 25, 2):
            w_ = wh0[pk[:, 0], 0, pk[:, 2], pk[:, 3]] + bw
            h_ = wh0[pk[:, 0], 1, pk[:, 2], pk[:, 3]] + bh
            bx = torch.stack([cx - w_ / 2, cy - h_ / 2, cx + w_ / 2, cy + h_ / 2], 1)
            stab, n_kept = float('inf'), 0
            for n in range(N):
                sel = pk[:, 0] == n
                kept, st_n = nms_stability(bx[sel], lg[sel], 0.5, 2 * MARGIN * design['gain'] * design['nu'])
                stab, n_kept = min(stab, st_n), n_kept + len(kept)
            # the most detections among the clearly stable sizes, then the most stable
            key = (min(stab, 1.5), n_kept, stab)
            if best_wh is None or key > best_wh[4]:
                best_wh = (stab, float(bw), float(bh), n_kept, key)
    car['wh'] = torch.tensor([best_wh[1], best_wh[2]])
    design['wh'] = best_wh
    sd2[p + 'hm_hp.2.weight'] = torch.zeros_like(sd[p + 'hm_hp.2.weight'])
    sd2[p + 'hm_hp.2.bias'] = torch.full_like(sd[p + 'hm_hp.2.bias'], -9.0)
    hidden_of = {h: p + h + '.2' for h in orc.KM3D_HEADS}
    if live_hp:
        # The keypoint branch shares the heat-map branch's FIRST conv (identical hidden features in both implementations), and its channels LIVE_JOINTS
        # are the class-0 logit moved so that the decode's 0.1 falls where the class threshold 0.3 falls: hm_hp[j] = hm[0] + logit(0.1) - logit(0.3).
        # Their peaks above 0.1 are then exactly the class-0 peaks above 0.3 -- every margin of the heat-map design carries over -- and sit on the object
        # centres: the centre key point (joint 8, regressed offset ~0) snaps to them, a far corner (joint 1, regressed ~24 px away) is refused by the
        # distance term of the six-term mask (km3d_head.py:236-238).  hp_offset: the seeded conv x 1/16 + a constant sub-pixel shift, like `reg`.
        for leaf in ('0.weight', '0.bias'):
            sd2[p + 'hm_hp.' + leaf] = sd[p + 'hm.' + leaf].clone()
        hidden_of['hm_hp'] = p + 'hm.2'
        shift = math.log(0.1 / 0.9) - _thr_logit()
        for j in LIVE_JOINTS:
            sd2[p + 'hm_hp.2.weight'][j] = w_hm[0]
            sd2[p + 'hm_hp.2.bias'][j] = b_hm[0] + shift
        sd2[p + 'hp_offset.2.weight'] = sd[p + 'hp_offset.2.weight'] * 0.0625
        sd2[p + 'hp_offset.2.bias'] = torch.tensor([0.25, 0.4])
    for h, bias in car.items():
        sd2[p + h + '.2.weight'] = sd[p + h + '.2.weight'] * 0.0625
        sd2[p + h + '.2.bias'] = bias.clone()
    # pass 2 (cheap): the nine last convs on the oracle's hidden features, then the reference decode
    out = {}
    with torch.no_grad():
        for h in orc.KM3D_HEADS:
            out[h] = F.conv2d(convs[hidden_of[h]]['x'], orc.fp16_round(sd2[p + h + '.2.weight']), sd2[p + h + '.2.bias'])
        dets = orc.km3d_get_bboxes(out, P2, (H, W), THR, 0.5, const=sd.get('bbox_head.const'))
    return dict(cfg=cfg, model=m, sd=sd2, img=img, P2=P2, maps=out, dets=dets, design=design, off_rms=off_rms, off_max=off_max, N=N, H=H, W=W, live_hp=live_hp)


def oracle_side_margins(case, noise, iou_thr=0.5):
    """every decision of the decode on the ORACLE's maps against `noise` (logit units, per designed class): threshold, local maximum,
    top-K cut, NMS IoU / score order, rotation bin.  Returns a summary; asserts the margins."""
    maps, N = case['maps'], case['N']
    thr_l = _thr_logit()
    hm = maps['hm']
    n_peaks = []
    for c in range(2):
        lg = hm[:, c]
        nb = _nb_max(lg)
        near = lg > thr_l - MARGIN * noise[c]
        sep = (lg - nb).abs()[near].min().item()
        tm = (lg - thr_l).abs()[near & (lg > nb)].min().item() if bool((near & (lg > nb)).any()) else float('inf')
        assert sep > MARGIN * noise[c], 'class %d: a position near / above the threshold is within %.3e of its best neighbour (noise %.3e)' % (c, sep, noise[c])
        assert tm > MARGIN * noise[c], 'class %d: a local maximum sits %.3e from the threshold (noise %.3e)' % (c, tm, noise[c])
        assert lg.max().item() < TOP_LOGIT + 1.0, 'class %d: strongest logit %.2f would saturate the sigmoid' % (c, lg.max().item())
        n_peaks.append([int(((lg[b] > nb[b]) & (lg[b] > thr_l)).sum()) for b in range(N)])
    dead = [j for j in range(9) if not (case.get('live_hp') and j in LIVE_JOINTS)]
    assert hm[:, 2:].max().item() < -5 and maps['hm_hp'][:, dead].max().item() < -5
    per_frame = [sum(n_peaks[c][b] for c in range(len(n_peaks))) for b in range(N)]
    assert max(per_frame) < 100, 'more peaks than the top-K keeps: %s' % per_frame
    # NMS: the outcome must be out of reach of the noise (see nms_stability), and the rotation-bin decision of every detection too
    n_sup = n_pairs = 0
    worst_noise = max(noise)
    stab = float('inf')
    for b in range(N):
        heat = torch.sigmoid(hm[b])
        nbm = F.max_pool2d(heat[None], 3, 1, 1)[0]
        pk = torch.nonzero((heat == nbm) & (heat > THR))
        Wm = hm.shape[3]
        ind = pk[:, 1] * Wm + pk[:, 2]
        lg = hm[b][pk[:, 0], pk[:, 1], pk[:, 2]]
        reg = maps['reg'][b].reshape(2, -1)[:, ind].t()
        wh = maps['wh'][b].reshape(2, -1)[:, ind].t()
        xs, ys = pk[:, 2].float() + reg[:, 0], pk[:, 1].float() + reg[:, 1]
        bx = torch.stack([xs - wh[:, 0] / 2, ys - wh[:, 1] / 2, xs + wh[:, 0] / 2, ys + wh[:, 1] / 2], 1) * 4
        bx[:, 0].clamp_(min=0); bx[:, 1].clamp_(min=0); bx[:, 2].clamp_(max=case['W']); bx[:, 3].clamp_(max=case['H'])   # noqa: E702
        kept, st_b = nms_stability(bx, lg, iou_thr, MARGIN * worst_noise)
        stab = min(stab, st_b)
        iou = _iou_matrix(bx)
        iou.fill_diagonal_(0)
        n_pairs += int((iou > 0.02).sum()) // 2
        n_sup += len(lg) - len(kept)
        assert len(kept) == len(case['dets'][b][0]), (len(kept), len(case['dets'][b][0]))
        rot = maps['rot'][b].reshape(8, -1)[:, ind]
        assert (rot[1] - rot[5]).abs().min().item() > 0.5 and rot[3].abs().min().item() > 0.5 and rot[7].abs().min().item() > 0.5
    assert stab >= 1.0, 'the NMS outcome is within reach of the noise: stability %.2f (1 = IoU 0.04 from the threshold / scores 10 x noise apart)' % stab
    return dict(peaks=n_peaks, pairs=n_pairs, suppressions=n_sup, nms_stability=stab)


def keypoint_margins(case, pos_noise):
    """`live_hp`: every decision of the key point <-> heat-map association (km3d_head.py:205-244) for every candidate above the class threshold, on the
    ORACLE's maps: which live peak is nearest, and each of the six terms of the reject mask, against `pos_noise` (feature pixels).  -> (snapped, refused by
    distance) counts per frame; asserts the margins."""
    maps, N = case['maps'], case['N']
    hm = maps['hm']
    Wm = hm.shape[3]
    snapped, refused = [0] * N, [0] * N
    for b in range(N):
        heat = torch.sigmoid(hm[b])
        pk = torch.nonzero((heat == F.max_pool2d(heat[None], 3, 1, 1)[0]) & (heat > THR))      # candidates (class, y, x) above the class threshold
        ind = pk[:, 1] * Wm + pk[:, 2]
        reg = maps['reg'][b].reshape(2, -1)[:, ind].t()
        wh = maps['wh'][b].reshape(2, -1)[:, ind].t()
        cx, cy = pk[:, 2].float() + reg[:, 0], pk[:, 1].float() + reg[:, 1]
        l_, t_, r_, b_ = cx - wh[:, 0] / 2, cy - wh[:, 1] / 2, cx + wh[:, 0] / 2, cy + wh[:, 1] / 2
        lim = torch.maximum(b_ - t_, r_ - l_) * 0.3
        hps = maps['hps'][b].reshape(18, -1)[:, ind].t()
        for j in LIVE_JOINTS:
            hh = torch.sigmoid(maps['hm_hp'][b, j])
            hp = torch.nonzero((hh == F.max_pool2d(hh[None, None], 3, 1, 1)[0, 0]) & (hh > 0.1))  # live peaks (y, x): the designed class-0 peaks
            assert len(hp) >= 1 and len(hp) < 100
            hi = hp[:, 0] * Wm + hp[:, 1]
            off = maps['hp_offset'][b].reshape(2, -1)[:, hi].t()
            hx, hy = hp[:, 1].float() + off[:, 0], hp[:, 0].float() + off[:, 1]
            kx, ky = pk[:, 2].float() + hps[:, 2 * j], pk[:, 1].float() + hps[:, 2 * j + 1]     # regressed key point j of every candidate
            d = ((kx[:, None] - hx[None]) ** 2 + (ky[:, None] - hy[None]) ** 2).sqrt()
            ds, order = d.sort(dim=1)
            if d.shape[1] > 1:
                assert (ds[:, 1] - ds[:, 0]).min().item() > MARGIN * pos_noise, 'joint %d: two live peaks at nearly the same distance of a regressed key point' % j
            sx, sy, dmin = hx[order[:, 0]], hy[order[:, 0]], ds[:, 0]
            terms = torch.stack([sx - l_, r_ - sx, sy - t_, b_ - sy, lim - dmin], 1)             # each > 0 <=> that reject term is false
            assert terms.abs().min().item() > MARGIN * pos_noise, 'joint %d: a term of the reject mask within %.3e px of flipping' % (j, terms.abs().min().item())
            ok = (terms > 0).all(dim=1)
            snapped[b] += int(ok.sum())
            refused[b] += int((terms[:, 4] < 0).sum())
    return snapped, refused


def test_config5_km3d_fp16_batch16_live_keypoint_heat_map_snaps_identically():
    """The association branch in the timed type at the timed size (VERDICT r5 missing #3): the same margin-controlled workload with LIVE peaks in the key point
    heat map -- the centre key point of every detection snaps to a heat-map peak, a corner key point is refused by the distance term -- and the
    detection set, every box field (the 3-D position is the least squares over the nine key points, snapped ones included) and every score still
    meet the literal bar in all 8 replicas of the two oracle frames."""
    _run_margin_case(build_case(2, live_hp=True))


def test_config5_km3d_fp16_batch16_margin_controlled_detection_set_is_identical():
    _run_margin_case(build_case(2))


def _run_margin_case(case):
    N, REP = case['N'], 8
    m, sd, cfg, H, W = case['model'], case['sd'], case['cfg'], case['H'], case['W']
    print('\n[margin workload C5] DCN offsets (16 blocks): worst rms %.3f px, max %.2f px' % (case['off_rms'], case['off_max']))
    assert case['off_max'] < 1.0, 'offsets are meant to be sub-pixel'
    d = case['design']
    print('[margin workload C5] Fisher filter (regulariser %g of the top eigenvalue, target grading %g), gain %g, proxy noise %.2e; (ratio to the proxy, peaks above, '
          'level, gap, neighbour separation): class 0 %s, class 1 %s; 2-D box %g x %g feature px (NMS stability %.2f units against the proxy, %d survivors)'
          % (d['lam'], d['grade'], d['gain'], d['nu'], tuple(round(v, 4) for v in d['l0']), tuple(round(v, 4) for v in d['l1']), d['wh'][1], d['wh'][2], d['wh'][0], d['wh'][3]))
    assert d['ratio'] > 1.5 * MARGIN, 'the design has no margin even against the proxy noise'
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = torch.float16
    B = N * REP
    img = case['img'].repeat(REP, 1, 1, 1).cuda()
    P2 = case['P2'].repeat(REP, 1, 1).cuda()
    with torch.no_grad():
        scores, boxes, cls, count = [t.cpu() for t in m.forward_device(img, P2)]
        maps_h = {k: v.float().cpu() for k, v in m._last_raw.items()}          # NHWC fp32
    assert int(count.min()) >= 0
    # ---- observed noise between the two implementations on the designed heat-map channels (logits), worst over every replica
    hm_o = case['maps']['hm']                                                 # [N, 3, H/4, W/4]
    hm_h = maps_h['hm'].permute(0, 3, 1, 2).reshape(REP, N, *hm_o.shape[1:])
    noise = [max((hm_h[r, :, c] - hm_o[:, c]).abs().max().item() for r in range(REP)) for c in range(2)]
    rel = {k: max(((maps_h[k].permute(0, 3, 1, 2).reshape(REP, N, *case['maps'][k].shape[1:])[r] - case['maps'][k]).abs().max()
                   / case['maps'][k].abs().max().clamp_min(1e-6)).item() for r in range(REP)) for k in orc.KM3D_HEADS}
    print('[margin workload C5] observed logit noise: class 0 %.3e, class 1 %.3e; head maps max-norm vs the fp16-rounded oracle: %s'
          % (noise[0], noise[1], ', '.join('%s %.1e' % kv for kv in rel.items())))
    summary = oracle_side_margins(case, noise)
    print('[margin workload C5] peaks above the threshold per class and frame %s; %d overlapping pairs, %d suppressions decided by NMS'
          % (summary['peaks'], summary['pairs'], summary['suppressions']))
    # ---- the literal bar, every replica
    worst_f = worst_s = 0.0
    n_det = 0
    for f in range(B):
        b = f % N
        k = int(count[f])
        s_o, b_o, l_o = case['dets'][b]
        n_det += len(s_o) if f < N else 0
        assert k == len(s_o), 'frame %d (oracle frame %d): %d detections, the oracle has %d' % (f, b, k, len(s_o))
        # one-to-one by 2-D box centre (the heat-map peak x 4 + the sub-pixel offset): identical sets <=> a bijection within half a pixel
        cg = torch.stack([(boxes[f, :k, 0] + boxes[f, :k, 2]) / 2, (boxes[f, :k, 1] + boxes[f, :k, 3]) / 2], 1)
        co = torch.stack([(b_o[:, 0] + b_o[:, 2]) / 2, (b_o[:, 1] + b_o[:, 3]) / 2], 1)
        d = (cg[:, None, :] - co[None, :, :]).norm(dim=2)
        j_of = d.argmin(dim=1)
        assert sorted(j_of.tolist()) == list(range(k)) and d.min(dim=1).values.max().item() < 0.5, 'frame %d: detection sets differ' % f
        scale = b_o.abs().amax(dim=0).clamp_min(1.0)
        so = torch.logit(s_o.double().clamp(max=1 - 1e-12))
        for i in range(k):
            j = int(j_of[i])
            assert int(cls[f, i]) == int(l_o[j, 0]), 'frame %d detection %d: class' % (f, i)
            worst_f = max(worst_f, float(((boxes[f, i] - b_o[j]).abs() / scale).max()))
            worst_s = max(worst_s, abs(float(scores[f, i] - s_o[j])))
        assert bool((scores[f, 1:k] <= scores[f, :k - 1]).all())
        for j in range(k - 1):                                                # output order: wherever consecutive oracle scores are clearly apart
            if float(so[j] - so[j + 1]) > MARGIN * max(noise):
                gi = [int(x) for x in torch.nonzero(j_of == j)[:, 0]][0]
                gj = [int(x) for x in torch.nonzero(j_of == j + 1)[:, 0]][0]
                assert gi < gj, 'frame %d: order of detections %d / %d' % (f, j, j + 1)
    print('[margin workload C5] %d detections over %d oracle frames x %d replicas: sets, classes and order identical in every replica; '
          'worst box field %.2e of its scale, worst score difference %.2e' % (n_det, N, REP, worst_f, worst_s))
    assert n_det >= 2 * N and summary['suppressions'] >= 3, 'workload must exercise the peak selection and NMS'
    assert worst_f <= 1e-3 and worst_s <= 1e-3
    if case['live_hp']:
        # positions entering the association: regressed offsets, sub-pixel shifts and box sizes -- their worst deviation between the two implementations (feature px)
        pos_noise = max(max((maps_h[k].permute(0, 3, 1, 2).reshape(REP, N, *case['maps'][k].shape[1:])[r] - case['maps'][k]).abs().max().item() for r in range(REP))
                        for k in ('hps', 'hp_offset', 'reg', 'wh'))
        snapped, refused = keypoint_margins(case, pos_noise)
        print('[margin workload C5, live key point heat map] position noise %.2e px; per frame: %s key points snapped to a heat-map peak, %s refused by the distance term'
              % (pos_noise, snapped, refused))
        assert min(snapped) >= 1 and min(refused) >= 1, 'every frame must exercise both outcomes of the association'
        for j in LIVE_JOINTS:                                                 # the live channels ARE the class-0 logit moved by a constant
            dj = (maps_h['hm_hp'][..., j] - (maps_h['hm'][..., 0] + (math.log(0.1 / 0.9) - _thr_logit()))).abs().max().item()
            assert dj < 1e-3, dj
