"""ORACLE (test infrastructure only -- the product path never imports this file).

Functional CPU restatement (torch fp32 on the host) of the reference's detector forward path, driven by a
``state_dict`` with the reference's key names.  Each function cites the reference lines it follows
(paths relative to the reference root).  It is pinned against the reference itself, imported on CPU through
``oracle/ref_shim.py`` in the build container (``oracle/make_golden.py`` -> ``tests/golden/*.npz``;
``tests/test_oracle_vs_reference.py``).  The reference ships no tests / golden vectors for this path
(SURVEY.md 0.2), so those self-generated fixtures are the pin.

``rnd`` is the activation-rounding hook: identity for the fp32 oracle; ``bf16_round`` reproduces the
rounding points of the HIP bf16 path (weights, every fused-layer output) with fp32 accumulation, which is
the oracle the bf16 kernels are compared against.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .nms_ref import nms_numpy

BN_EPS = 1e-5


def identity(x):
    return x


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


class Ctx:
    """state_dict + rounding policy."""

    def __init__(self, sd, rnd=identity):
        self.sd = {k: v.detach().cpu().float() if torch.is_floating_point(v) else v.detach().cpu() for k, v in sd.items()}
        self.rnd = rnd

    def w(self, key):
        """conv weight, rounded like the packed HIP weights are."""
        return self.rnd(self.sd[key])

    def has(self, key):
        return key in self.sd

    def bn(self, prefix):
        """eval-mode BatchNorm as an affine map (scale, shift); torch BatchNorm2d eval semantics."""
        g, b = self.sd[prefix + '.weight'], self.sd[prefix + '.bias']
        m, v = self.sd[prefix + '.running_mean'], self.sd[prefix + '.running_var']
        scale = g / torch.sqrt(v + BN_EPS)
        return scale, b - m * scale


def _affine(y, scale, shift):
    shp = [1, -1] + [1] * (y.dim() - 2)
    return y * scale.view(shp) + shift.view(shp)


def conv_bn_act(c, x, conv, bn=None, relu=True, stride=1, padding=1, dilation=1, groups=1, residual=None, out_round=True):
    """conv (+bias) (+eval BN) (+residual) (+ReLU) == one fused HIP layer; output rounded once."""
    w = c.w(conv + '.weight')
    y = F.conv2d(x, w, None, stride=stride, padding=padding, dilation=dilation, groups=groups)
    n = w.shape[0]
    scale = torch.ones(n)
    shift = torch.zeros(n)
    if c.has(conv + '.bias'):
        shift = c.sd[conv + '.bias'].clone()
    if bn is not None:
        s, t = c.bn(bn)
        shift = shift * s + t
        scale = s
    y = _affine(y, scale, shift)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return c.rnd(y) if out_round else y


# ------------------------------------------------------------------------------------------- backbone
def basic_block(c, p, x, stride=1, dilation=1):
    """backbones/resnet.py:23-52."""
    out = conv_bn_act(c, x, p + '.conv1', p + '.bn1', True, stride=stride, padding=1)
    res = x
    if c.has(p + '.downsample.0.weight'):
        res = conv_bn_act(c, x, p + '.downsample.0', p + '.downsample.1', False, stride=stride, padding=0)
    return conv_bn_act(c, out, p + '.conv2', p + '.bn2', True, padding=dilation, dilation=dilation, residual=res)


def bottleneck(c, p, x, stride=1, dilation=1):
    """backbones/resnet.py:55-91 (stride on the 3x3)."""
    out = conv_bn_act(c, x, p + '.conv1', p + '.bn1', True, padding=0)
    out = conv_bn_act(c, out, p + '.conv2', p + '.bn2', True, stride=stride, padding=dilation, dilation=dilation)
    res = x
    if c.has(p + '.downsample.0.weight'):
        res = conv_bn_act(c, x, p + '.downsample.0', p + '.downsample.1', False, stride=stride, padding=0)
    return conv_bn_act(c, out, p + '.conv3', p + '.bn3', True, padding=0, residual=res)


_LAYERS = {18: (basic_block, [2, 2, 2, 2]), 34: (basic_block, [3, 4, 6, 3]), 50: (bottleneck, [3, 4, 6, 3]),
           101: (bottleneck, [3, 4, 23, 3]), 152: (bottleneck, [3, 8, 36, 3])}


def resnet(c, p, x, depth=34, num_stages=3, out_indices=(0, 1, 2), strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1)):
    """backbones/resnet.py:184-198 (+ _make_layer :133-152: first block ignores dilation)."""
    block, layers = _LAYERS[depth]
    outs = []
    x = conv_bn_act(c, c.rnd(x), p + '.conv1', p + '.bn1', True, stride=2, padding=3)
    if -1 in out_indices:
        outs.append(x)
    x = F.max_pool2d(x, 3, 2, 1)
    for i in range(num_stages):
        for j in range(layers[i]):
            bp = '%s.layer%d.%d' % (p, i + 1, j)
            x = block(c, bp, x, strides[i], 1) if j == 0 else block(c, bp, x, 1, dilations[i])
        if i in out_indices:
            outs.append(x)
    return outs


# ------------------------------------------------------------------------------------------- stereo neck
def psm_cosine(c, left, right, max_disp, downsample):
    """lib/PSM_cost_volume.py:81-96: cost[b,d,y,x] = mean_c L[b,c,y,x] * R[b,c,y,x-d], 0 where x < d."""
    D = int(max_disp / downsample)
    B, C, H, W = left.shape
    cost = torch.zeros(B, D, H, W)
    for d in range(D):
        if d > 0:
            cost[:, d, :, d:] = (left[:, :, :, d:] * right[:, :, :, :-d]).mean(dim=1)
        else:
            cost[:, d] = (left * right).mean(dim=1)
    return c.rnd(cost)


def cost_volume(c, p, left, right, max_disp=192, downsample=16):
    """lib/PSM_cost_volume.py:45-68: 1x1 down-sample, concat volume, 2x (Conv3d+BN3d+ReLU), channel = f*D + d."""
    D = int(max_disp / downsample)
    lf = conv_bn_act(c, left, p + '.down_sample.0', p + '.down_sample.1', True, padding=0)
    rf = conv_bn_act(c, right, p + '.down_sample.0', p + '.down_sample.1', True, padding=0)
    B, Fc, H, W = lf.shape
    vol = torch.zeros(B, 2 * Fc, D, H, W)
    for d in range(D):
        if d > 0:
            vol[:, :Fc, d, :, d:] = lf[:, :, :, d:]
            vol[:, Fc:, d, :, d:] = rf[:, :, :, :-d]
        else:
            vol[:, :Fc, d] = lf
            vol[:, Fc:, d] = rf
    x = vol
    for i in (0, 3):
        w = c.w('%s.conv3d.%d.weight' % (p, i))
        y = F.conv3d(x, w, None, padding=1)
        s, t = c.bn('%s.conv3d.%d' % (p, i + 1))
        t = c.sd['%s.conv3d.%d.bias' % (p, i)] * s + t
        x = c.rnd(F.relu(_affine(y, s, t)))
    return x.reshape(B, -1, H, W)


def res_ghost(c, p, x):
    """lib/ghost_module.py:46-64 (stride 1): cat[x, primary(x), cheap(primary(x))][:oup]."""
    x1 = conv_bn_act(c, x, p + '.primary_conv.1', p + '.primary_conv.2', True, padding=1)
    x2 = conv_bn_act(c, x1, p + '.cheap_operation.0', p + '.cheap_operation.1', True, padding=1, groups=x1.shape[1])
    return torch.cat([x, x1, x2], dim=1)


def cost_volume_pyramid(c, p, v4, v8, v16):
    """detectors/yolostereo3d_core.py:63-71 (eval branch)."""
    x = res_ghost(c, p + '.four_to_eight.0', v4)
    x = c.rnd(F.avg_pool2d(x, 2))
    x = basic_block(c, p + '.four_to_eight.2', x)
    x = torch.cat([x, v8], dim=1)
    x = res_ghost(c, p + '.eight_to_sixteen.0', x)
    x = c.rnd(F.avg_pool2d(x, 2))
    x = basic_block(c, p + '.eight_to_sixteen.2', x)
    x = torch.cat([x, v16], dim=1)
    x = res_ghost(c, p + '.depth_reason.0', x)
    return basic_block(c, p + '.depth_reason.1', x)


def stereo_core(c, left, right, depth=34):
    """detectors/yolostereo3d_core.py:110-126 + StereoMerging.forward :88-94.  Returns features [B,C,H/16,W/16]
    and the three cost volumes (for stage-level parity)."""
    B = left.shape[0]
    feats = resnet(c, 'core.backbone', torch.cat([left, right], dim=0), depth=depth)
    lf = [f[:B] for f in feats]
    rf = [f[B:] for f in feats]
    v4 = psm_cosine(c, lf[0], rf[0], 96, 4)
    v8 = psm_cosine(c, lf[1], rf[1], 192, 8)
    v16 = cost_volume(c, 'core.neck.cost_volume_2', lf[2], rf[2], 192, 16)
    psv = cost_volume_pyramid(c, 'core.neck.depth_reasoning', v4, v8, v16)
    features = torch.cat([lf[2], psv], dim=1)
    return features, dict(s4=lf[0], s8=lf[1], s16=lf[2], r4=rf[0], v4=v4, v8=v8, v16=v16, psv=psv)


# ------------------------------------------------------------------------------------------- heads
def anchor_flatten(x, n_out):
    """lib/blocks.py:133-136."""
    return x.permute(0, 2, 3, 1).contiguous().view(x.shape[0], -1, n_out)


def stereo_head(c, feats, num_cls_output, num_reg_output=12, p='bbox_head'):
    """heads/detection_3d_head.py:501-533 (StereoHead) + forward :84-88.  Final convs stay fp32."""
    x = conv_bn_act(c, feats, p + '.cls_feature_extraction.0', None, True)
    x = conv_bn_act(c, x, p + '.cls_feature_extraction.3', None, True)
    cls = conv_bn_act(c, x, p + '.cls_feature_extraction.6', None, False, out_round=False)
    r = conv_bn_act(c, feats, p + '.reg_feature_extraction.0.sequence.0', p + '.reg_feature_extraction.0.sequence.1', True)
    r = basic_block(c, p + '.reg_feature_extraction.1', r)
    r = F.relu(r)
    reg = conv_bn_act(c, r, p + '.reg_feature_extraction.3', None, False, out_round=False)
    return anchor_flatten(cls, num_cls_output), anchor_flatten(reg, num_reg_output)


# ------------------------------------------------------------------------------------------- anchors
def generate_anchors(base_size, ratios, scales):
    """heads/anchors.py:152-183 (ratio-major, scale-minor; float64)."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    n = len(ratios) * len(scales)
    wh = base_size * np.tile(scales, len(ratios))
    areas = wh * wh
    rr = np.repeat(ratios, len(scales))
    w = np.sqrt(areas / rr)
    h = w * rr
    a = np.zeros((n, 4))
    a[:, 0] = -0.5 * w
    a[:, 1] = -0.5 * h
    a[:, 2] = w - 0.5 * w
    a[:, 3] = h - 0.5 * h
    return a


def anchors_for_image(H, W, anchors_cfg, mean_npy, std_npy):
    """heads/anchors.py:59-91: dense anchors [N,4] fp32 (cell-major, then anchor), priors [N,types,6,2] fp32."""
    all_anchors = np.zeros((0, 4), dtype=np.float32)
    sizes_tab = (np.array(anchors_cfg['sizes'], dtype=np.float64) * np.array(anchors_cfg['scales'], dtype=np.float64))
    for idx, lvl in enumerate(anchors_cfg['pyramid_levels']):
        fh, fw = (H + 2 ** lvl - 1) // (2 ** lvl), (W + 2 ** lvl - 1) // (2 ** lvl)
        base = generate_anchors(anchors_cfg['sizes'][idx], anchors_cfg['ratios'], anchors_cfg['scales'])
        sx = (np.arange(fw) + 0.5) * anchors_cfg['strides'][idx]
        sy = (np.arange(fh) + 0.5) * anchors_cfg['strides'][idx]
        gx, gy = np.meshgrid(sx, sy)
        shifts = np.stack([gx.ravel(), gy.ravel(), gx.ravel(), gy.ravel()], axis=1)
        lv = (base[None, :, :] + shifts[:, None, :]).reshape(-1, 4)
        all_anchors = np.append(all_anchors, lv, axis=0)
    # anchors2indexes (:45-57), on the float64 anchors
    bw = all_anchors[:, 2] - all_anchors[:, 0]
    bh = all_anchors[:, 3] - all_anchors[:, 1]
    size_idx = np.argmin(np.abs(np.sqrt(bw * bh)[None, :] - sizes_tab[:, None]), axis=0)
    ratio_idx = np.argmin(np.abs((bh / bw)[None, :] - np.asarray(anchors_cfg['ratios'], dtype=np.float64)[:, None]), axis=0)
    means = torch.tensor(mean_npy[:, size_idx, ratio_idx], dtype=torch.float32)  # [types, N, 6] (image.new -> fp32)
    stds = torch.tensor(std_npy[:, size_idx, ratio_idx], dtype=torch.float32)
    mean_std = torch.stack([means, stds], dim=-1).permute(1, 0, 2, 3)  # [N, types, 6, 2]
    anchors = torch.tensor(all_anchors.astype(np.float32))  # [N, 4]
    return anchors, means, mean_std


def anchor_mask(anchors, means, P2, y_min_max=(-0.5, 1.8), x_thr=40.0):
    """heads/anchors.py:99-111: ground-plane filter; note x AND y back-projections divide by fy."""
    xc = anchors[:, 0:4:2].mean(dim=1)
    yc = anchors[:, 1:4:2].mean(dim=1)
    fy = P2[:, 1:2, 1:2]
    cy = P2[:, 1:2, 2:3]
    cx = P2[:, 0:1, 2:3]
    z = means[:, :, 0]  # [types, N]
    x3d = (xc * z - cx * z) / fy
    y3d = (yc * z - cy * z) / fy
    return torch.any((y3d > y_min_max[0]) * (y3d < y_min_max[1]) * (x3d.abs() < x_thr), dim=1)  # [B, N]


def decode(anchor, deltas, mean_std_sel, alpha_score):
    """heads/detection_3d_head.py:218-263 with the class already selected: mean_std_sel [K,6,2]."""
    w = anchor[:, 2] - anchor[:, 0]
    h = anchor[:, 3] - anchor[:, 1]
    cx = anchor[:, 0] + 0.5 * w
    cy = anchor[:, 1] + 0.5 * h
    pcx = cx + deltas[:, 0] * 0.1 * w
    pcy = cy + deltas[:, 1] * 0.1 * h
    pw = torch.exp(deltas[:, 2] * 0.2) * w
    ph = torch.exp(deltas[:, 3] * 0.2) * h
    x1, y1, x2, y2 = pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph
    c3x = cx + deltas[:, 4] * 0.1 * w
    c3y = cy + deltas[:, 5] * 0.1 * h
    six = [deltas[:, 6 + i] * mean_std_sel[:, i, 1] + mean_std_sel[:, i, 0] for i in range(6)]
    z, sin2a, cos2a, w3, h3, l3 = six
    alpha = torch.atan2(sin2a, cos2a) / 2.0
    boxes = torch.stack([x1, y1, x2, y2, c3x, c3y, z, w3, h3, l3, alpha], dim=1)
    boxes[alpha_score < 0.5, -1] += np.pi
    return boxes, mean_std_sel[:, 0, 0] > 0


def get_bboxes(cls_preds, reg_preds, anchors, mean_std, mask, img_hw, num_classes, score_thr, nms_iou_thr):
    """heads/detection_3d_head.py:341-400 for ONE sample (class-agnostic NMS).
    Returns (scores[K], boxes[K,11], labels[K], anchor_index[K])."""
    cls = cls_preds.sigmoid()
    idx = torch.nonzero(mask, as_tuple=False)[:, 0]
    cls_score = cls[idx, :num_classes]
    alpha_score = cls[idx, num_classes]
    max_score, label = cls_score.max(dim=-1)
    hi = max_score > score_thr
    idx, max_score, label, alpha_score = idx[hi], max_score[hi], label[hi], alpha_score[hi]
    sel = mean_std[idx, label]  # [K,6,2]
    boxes, zmask = decode(anchors[idx], reg_preds[idx], sel, alpha_score)
    H, W = img_hw
    boxes[:, 0].clamp_(min=0)
    boxes[:, 1].clamp_(min=0)
    boxes[:, 2].clamp_(max=W)
    boxes[:, 3].clamp_(max=H)
    # QUIRK kept from the reference (detection_3d_head.py:375-379,392-394): cls_score / max_score / bboxes are
    # filtered by the z-prior mask but `label` is NOT, and is then indexed with the NMS keep indices of the
    # FILTERED list -> labels are those of the unfiltered list at the same positions.
    boxes, max_score, idx = boxes[zmask], max_score[zmask], idx[zmask]
    keep = torch.from_numpy(nms_numpy(boxes[:, :4].numpy(), max_score.numpy(), nms_iou_thr))
    return max_score[keep], boxes[keep], label[keep], idx[keep]


# ------------------------------------------------------------------------------------------- mono
def look_ground(c, p, x, P2, baseline=0.54, elevation=1.65):
    """lib/look_ground.py:24-71."""
    P2 = P2.clone().float()
    P2[:, 0:2] /= 16.0
    w = c.w(p + '.disp_create.0.weight')
    disp = torch.tanh(F.conv2d(x, w, c.sd[p + '.disp_create.0.bias'], padding=1))
    disp = 0.1 * (0.05 * disp + 0.95 * disp)
    B, _, H, W = x.shape
    yy = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(1, H, W)
    fy, cy, Ty = P2[:, 1:2, 1:2], P2[:, 1:2, 2:3], P2[:, 1:2, 3:4]
    disparity = F.relu(fy * baseline * (yy - cy) / (torch.abs(fy * elevation + Ty) + 1e-10))
    x_base = torch.linspace(-1, 1, W).repeat(B, H, 1)
    y_base = torch.linspace(-1, 1, H).repeat(B, W, 1).transpose(1, 2)
    h_mean = 1.535
    y_shifts_base = F.relu(h_mean * (yy - cy) / (2 * (elevation - 0.5 * h_mean))) / (H * 0.5)
    y_shifts = y_shifts_base + disp[:, 0]
    flow = torch.stack((x_base, y_base + y_shifts), dim=3)
    feats = torch.cat([disparity.unsqueeze(1), x], dim=1)
    out = c.rnd(F.grid_sample(feats, flow, mode='bilinear', padding_mode='border', align_corners=True))
    we = c.w(p + '.extract.weight')
    ext = F.conv2d(out, we, c.sd[p + '.extract.bias'])
    return c.rnd(F.relu(x + ext * c.sd[p + '.alpha']))


def cls_tower(c, feats, p):
    x = conv_bn_act(c, feats, p + '.cls_feature_extraction.0', None, True)
    x = conv_bn_act(c, x, p + '.cls_feature_extraction.3', None, True)
    return conv_bn_act(c, x, p + '.cls_feature_extraction.6', None, False, out_round=False)


def ground_aware_head(c, feats, P2, num_cls_output, p='bbox_head'):
    """detectors/yolomono3d_detector.py:12-53 (GroundAwareHead)."""
    cls = cls_tower(c, feats, p)
    r = look_ground(c, p + '.reg_feature_extraction.0', feats, P2)
    r = conv_bn_act(c, r, p + '.reg_feature_extraction.1', p + '.reg_feature_extraction.2', True)
    r = conv_bn_act(c, r, p + '.reg_feature_extraction.4', p + '.reg_feature_extraction.5', True)
    reg = conv_bn_act(c, r, p + '.reg_feature_extraction.7', None, False, out_round=False)
    return anchor_flatten(cls, num_cls_output), anchor_flatten(reg, 12)


def dcn_head(c, feats, num_cls_output, p='bbox_head'):
    """heads/detection_3d_head.py:47-88 (base head): ModulatedDeformConvPack + BN + ReLU, conv + BN + ReLU, conv."""
    from . import dcn_ref
    cls = cls_tower(c, feats, p)
    q = p + '.reg_feature_extraction.0'
    off_logits = F.conv2d(feats, c.w(q + '.conv_offset.weight'), c.sd[q + '.conv_offset.bias'], padding=1)
    o1, o2, mask = torch.chunk(off_logits, 3, dim=1)
    y = dcn_ref.deform_conv_forward(feats, torch.cat((o1, o2), 1), torch.sigmoid(mask), c.sd[q + '.weight'], c.sd[q + '.bias'],
                                    1, 1, 1, 1, 1, rnd=(c.rnd if c.rnd is not identity else None))
    s, t = c.bn(p + '.reg_feature_extraction.1')
    r = c.rnd(F.relu(_affine(y, s, t)))
    r = conv_bn_act(c, r, p + '.reg_feature_extraction.3', p + '.reg_feature_extraction.4', True)
    reg = conv_bn_act(c, r, p + '.reg_feature_extraction.6', None, False, out_round=False)
    return anchor_flatten(cls, num_cls_output), anchor_flatten(reg, 12)


def mono3d_forward(sd, cfg, img, P2, rnd=identity, return_stages=False):
    """Yolo3D / GroundAwareYolo3D test_forward (detectors/yolomono3d_detector.py:100-120), B >= 1, no post-optimisation."""
    c = Ctx(sd, rnd)
    bb = cfg.backbone
    feats = resnet(c, 'core.backbone', img.float(), depth=bb.depth, num_stages=bb.num_stages, out_indices=bb.out_indices)[0]
    ncls = len(cfg.obj_types)
    if cfg.name == 'GroundAwareYolo3D':
        cls_preds, reg_preds = ground_aware_head(c, feats, P2, ncls + 1)
    else:
        cls_preds, reg_preds = dcn_head(c, feats, ncls + 1)
    mean_npy, std_npy = load_priors(cfg.head.preprocessed_path, cfg.obj_types)
    H, W = img.shape[2:]
    anchors, means, mean_std = anchors_for_image(H, W, cfg.head.anchors_cfg, mean_npy, std_npy)
    mask = anchor_mask(anchors, means, P2.float())
    tc = cfg.head.test_cfg
    outs = [get_bboxes(cls_preds[b], reg_preds[b], anchors, mean_std, mask[b], (H, W), ncls,
                       getattr(tc, 'score_thr', 0.5), getattr(tc, 'nms_iou_thr', 0.5)) for b in range(img.shape[0])]
    if return_stages:
        return outs, dict(features=feats, cls_preds=cls_preds, reg_preds=reg_preds, mask=mask)
    return outs


# ------------------------------------------------------------------------------------------- detectors
def load_priors(preprocessed_path, obj_types):
    import os
    d = os.path.join(preprocessed_path, 'training')
    mean = np.stack([np.load(os.path.join(d, 'anchor_mean_%s.npy' % t)) for t in obj_types])
    std = np.stack([np.load(os.path.join(d, 'anchor_std_%s.npy' % t)) for t in obj_types])
    return mean, std


def stereo3d_forward(sd, cfg, left, right, P2, rnd=identity, return_stages=False):
    """Stereo3D.test_forward (detectors/yolostereo3d_detector.py:77-96) generalised to B >= 1: per-sample
    post-processing identical to the reference's batch-1 path.  Returns a list of (scores, boxes, labels)."""
    c = Ctx(sd, rnd)
    depth = cfg.backbone.depth
    feats, stages = stereo_core(c, left.float(), right.float(), depth)
    ncls = len(cfg.obj_types)
    cls_preds, reg_preds = stereo_head(c, feats, ncls + 1)
    mean_npy, std_npy = load_priors(cfg.head.preprocessed_path, cfg.obj_types)
    H, W = left.shape[2:]
    anchors, means, mean_std = anchors_for_image(H, W, cfg.head.anchors_cfg, mean_npy, std_npy)
    mask = anchor_mask(anchors, means, P2.float())
    tc = cfg.head.test_cfg
    outs = []
    for b in range(left.shape[0]):
        outs.append(get_bboxes(cls_preds[b], reg_preds[b], anchors, mean_std, mask[b], (H, W), ncls,
                               getattr(tc, 'score_thr', 0.5), getattr(tc, 'nms_iou_thr', 0.5)))
    if return_stages:
        stages.update(features=feats, cls_preds=cls_preds, reg_preds=reg_preds, anchors=anchors, mask=mask, mean_std=mean_std)
        return outs, stages
    return outs
