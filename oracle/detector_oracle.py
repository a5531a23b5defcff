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


def fp16_round(x):
    """rounding points of the HIP fp16 path (BASELINE config 5: KM3D in half precision)."""
    return x.to(torch.float16).to(torch.float32)


class Ctx:
    """state_dict + rounding policy."""

    def __init__(self, sd, rnd=identity, stage_taps=None):
        self.sd = {k: v.detach().cpu().float() if torch.is_floating_point(v) else v.detach().cpu() for k, v in sd.items()}
        self.rnd = rnd
        # optional list receiving one record per fused operation of the path (kind, state_dict key, input / output tensors, the
        # op's parameters): what tests/test_stage_taps_gpu.py feeds, op by op, to the HIP kernels (teacher forcing: every HIP stage
        # sees the ORACLE's input, so a 1-ulp flip cannot cascade and each stage can be held to ulps)
        self.stage_taps = stage_taps

    def tap(self, kind, key, **kw):
        if self.stage_taps is not None:
            self.stage_taps.append(dict(kind=kind, key=key, **kw))

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
    # depth-wise (ghost cheap_operation) weights stay fp32 in the HIP path (hip_ops.pack_dwconv: 9 fp32 taps per channel, VALU
    # kernel): only the MFMA operands are rounded to the 16-bit compute type
    w = c.w(conv + '.weight') if groups == 1 else c.sd[conv + '.weight']
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
    y = c.rnd(y) if out_round else y
    c.tap('dwconv' if groups > 1 else 'conv', conv, bn=bn, x=x, residual=residual, y=y, relu=relu, stride=stride, padding=padding,
          dilation=dilation, out_round=out_round)
    return y


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
    img = x
    taps, c.stage_taps = c.stage_taps, None          # the stem is tapped as ONE fused stage (conv + BN + ReLU + max pool), below
    x = conv_bn_act(c, c.rnd(x), p + '.conv1', p + '.bn1', True, stride=2, padding=3)
    c.stage_taps = taps
    if -1 in out_indices:
        outs.append(x)
    x = F.max_pool2d(x, 3, 2, 1)
    c.tap('stem', p, x=img, y=x)
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
    cost = c.rnd(cost)
    c.tap('psm_cosine', 'D%d' % D, left=left, right=right, y=cost)
    return cost


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
    c.tap('costvol_build', p, left=lf, right=rf, y=vol)
    for i in (0, 3):
        w = c.w('%s.conv3d.%d.weight' % (p, i))
        y = F.conv3d(x, w, None, padding=1)
        s, t = c.bn('%s.conv3d.%d' % (p, i + 1))
        t = c.sd['%s.conv3d.%d.bias' % (p, i)] * s + t
        y = c.rnd(F.relu(_affine(y, s, t)))
        c.tap('conv3d', '%s.conv3d.%d' % (p, i), x=x, y=y, last=(i == 3))
        x = y
    c.tap('cost_volume', p, left=lf, right=rf, y=x.reshape(B, -1, H, W))      # the three launches as one stage
    return x.reshape(B, -1, H, W)


def res_ghost(c, p, x):
    """lib/ghost_module.py:46-64 (stride 1): cat[x, primary(x), cheap(primary(x))][:oup]."""
    x1 = conv_bn_act(c, x, p + '.primary_conv.1', p + '.primary_conv.2', True, padding=1)
    x2 = conv_bn_act(c, x1, p + '.cheap_operation.0', p + '.cheap_operation.1', True, padding=1, groups=x1.shape[1])
    return torch.cat([x, x1, x2], dim=1)


def cost_volume_pyramid(c, p, v4, v8, v16):
    """detectors/yolostereo3d_core.py:63-71 (eval branch)."""
    def pool(x, key):
        y = c.rnd(F.avg_pool2d(x, 2))
        c.tap('avgpool', key, x=x, y=y)
        return y

    x = res_ghost(c, p + '.four_to_eight.0', v4)
    x = pool(x, p + '.four_to_eight.1')
    x = basic_block(c, p + '.four_to_eight.2', x)
    x = torch.cat([x, v8], dim=1)
    x = res_ghost(c, p + '.eight_to_sixteen.0', x)
    x = pool(x, p + '.eight_to_sixteen.1')
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


def get_bboxes(cls_preds, reg_preds, anchors, mean_std, mask, img_hw, num_classes, score_thr, nms_iou_thr, clip=True):
    """heads/detection_3d_head.py:341-400 for ONE sample (class-agnostic NMS).
    Returns (scores[K], boxes[K,11], labels[K], anchor_index[K]).  ``clip=False`` = the reference's ``img_batch is None``
    (:375-376: ClipBoxes skipped)."""
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
    if clip:
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
    c.tap('dcn_head', q, bn=p + '.reg_feature_extraction.1', x=feats, y=r, logits=off_logits)
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


# ------------------------------------------------------------------------------------------- KM3D (DLA-34)
def dla_block(c, p, x, residual=None, stride=1):
    """backbones/dla.py:37-61 (BasicBlock with external residual)."""
    if residual is None:
        residual = x
    out = conv_bn_act(c, x, p + '.conv1', p + '.bn1', True, stride=stride, padding=1)
    return conv_bn_act(c, out, p + '.conv2', p + '.bn2', True, padding=1, residual=residual)


def dla_tree(c, p, x, levels, in_ch, out_ch, stride, level_root, residual=None, children=None):
    """backbones/dla.py:177-230."""
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
    if stride > 1:
        c.tap('maxpool', p + '.downsample', x=x, y=bottom, k=stride)
    if in_ch != out_ch:
        residual = conv_bn_act(c, bottom, p + '.project.0', p + '.project.1', False, padding=0)
    else:
        residual = bottom
    if level_root:
        children.append(bottom)
    if levels == 1:
        x1 = dla_block(c, p + '.tree1', x, residual, stride)
        x2 = dla_block(c, p + '.tree2', x1)
        cat = torch.cat([x2, x1] + children, dim=1)
        return conv_bn_act(c, cat, p + '.root.conv', p + '.root.bn', True, padding=0)
    x1 = dla_tree(c, p + '.tree1', x, levels - 1, in_ch, out_ch, stride, False, residual)
    children.append(x1)
    return dla_tree(c, p + '.tree2', x1, levels - 1, out_ch, out_ch, 1, False, children=children)


def dla34(c, p, img):
    """backbones/dla.py:317-326, DLA-34: levels [1,1,1,2,2,1], channels [16,32,64,128,256,512]; returns levels 0..5."""
    ch = [16, 32, 64, 128, 256, 512]
    lv = [1, 1, 1, 2, 2, 1]
    x = conv_bn_act(c, c.rnd(img), p + '.base_layer.0', p + '.base_layer.1', True, padding=3)
    y = []
    x = conv_bn_act(c, x, p + '.level0.0', p + '.level0.1', True, padding=1)
    y.append(x)
    x = conv_bn_act(c, x, p + '.level1.0', p + '.level1.1', True, stride=2, padding=1)
    y.append(x)
    for i in range(2, 6):
        x = dla_tree(c, '%s.level%d' % (p, i), x, lv[i], ch[i - 1], ch[i], 2, i > 2)
        y.append(x)
    return y


def dcn_bn_relu(c, p, x):
    """backbones/dla_utils.py:42-56: ModulatedDeformConvPack 3x3 + BN + ReLU."""
    from . import dcn_ref
    q = p + '.conv'
    logits = F.conv2d(x, c.w(q + '.conv_offset.weight'), c.sd[q + '.conv_offset.bias'], padding=1)
    o1, o2, mask = torch.chunk(logits, 3, dim=1)
    y = dcn_ref.deform_conv_forward(x, torch.cat((o1, o2), 1), torch.sigmoid(mask), c.sd[q + '.weight'], c.sd[q + '.bias'],
                                    1, 1, 1, 1, 1, rnd=(c.rnd if c.rnd is not identity else None))
    s, t = c.bn(p + '.actf.0')
    out = c.rnd(F.relu(_affine(y, s, t)))
    taps = getattr(c, 'taps', None)
    if taps is not None:              # stage taps for teacher-forced per-layer parity (tests/test_km3d_gpu.py)
        taps[p] = (x, out)
    c.tap('dcn', p, x=x, y=out, logits=logits)
    return out


def ida_up(c, p, layers, startp, endp, up_f):
    """backbones/dla_utils.py:76-85."""
    for i in range(startp + 1, endp):
        k = i - startp
        f = int(up_f[k])
        w = c.sd['%s.up_%d.weight' % (p, k)]
        proj = dcn_bn_relu(c, '%s.proj_%d' % (p, k), layers[i])
        up = c.rnd(F.conv_transpose2d(proj, w, None, stride=f, padding=f // 2, groups=w.shape[0]))
        summed = c.rnd(up + layers[i - 1])
        c.tap('dwconvT', '%s.up_%d' % (p, k), x=proj, add=layers[i - 1], y=summed, f=f)
        layers[i] = dcn_bn_relu(c, '%s.node_%d' % (p, k), summed)


def dla_seg_upsample(c, p, tensors, first_level=2, last_level=5):
    """backbones/dla_utils.py:126-155 (+ DLAUp :89-112)."""
    layers = list(tensors)
    channels = [64, 128, 256, 512]
    scales = np.array([1, 2, 4, 8])
    out = [layers[-1]]
    for i in range(len(channels) - 1):
        j = -i - 2
        ida_up(c, '%s.dla_up.ida_%d' % (p, i), layers, len(layers) - i - 2, len(layers), scales[j:] // scales[j])
        scales[j + 1:] = scales[j]
        out.insert(0, layers[-1])
    y = [out[i].clone() for i in range(last_level - first_level)]
    ida_up(c, p + '.ida_up', y, 0, len(y), [2 ** i for i in range(last_level - first_level)])
    return y[-1]


KM3D_HEADS = ('hm', 'wh', 'hps', 'rot', 'dim', 'prob', 'reg', 'hm_hp', 'hp_offset')


def km3d_heads(c, feat, heads, p='bbox_head.head_layers'):
    """heads/km3d_head.py:353-357: nine (conv3x3 + ReLU, conv1x1) heads; final maps stay fp32."""
    out = {}
    for h in heads:
        x = conv_bn_act(c, feat, '%s.%s.0' % (p, h), None, True)
        out[h] = conv_bn_act(c, x, '%s.%s.2' % (p, h), None, False, padding=0, out_round=False)
    c.tap('km3d_head', p, x=feat, y=dict(out))
    return out


def _gather_map(m, ind):
    """_transpose_and_gather_feat (rtm3d_utils.py:195-199): m [B,C,H,W], ind [B,K] -> [B,K,C]."""
    B, C, H, W = m.shape
    f = m.permute(0, 2, 3, 1).reshape(B, H * W, C)
    return f.gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], C))


def _heat_nms(heat):
    hmax = F.max_pool2d(heat, 3, 1, 1)
    return heat * (hmax == heat).float()


def km3d_gen_position(kps, dim, rot, calib, const):
    """networks/utils/rtm3d_utils.py:314-455 (without the random 1e-8 jitter before the 3x3 inverse)."""
    b, cn = kps.shape[:2]
    off_set = calib[:, 0, 3] / calib[:, 0, 0]
    si = torch.zeros_like(kps[:, :, 0:1]) + calib[:, 0:1, 0:1]
    alpha_idx = (rot[:, :, 1] > rot[:, :, 5]).float()
    alpha1 = torch.atan(rot[:, :, 2] / rot[:, :, 3]) + (-0.5 * np.pi)
    alpha2 = torch.atan(rot[:, :, 6] / rot[:, :, 7]) + (0.5 * np.pi)
    alpha = (alpha1 * alpha_idx + alpha2 * (1 - alpha_idx)).unsqueeze(2)
    rot_y = alpha + torch.atan2(kps[:, :, 16:17] - calib[:, 0:1, 2:3], si)
    rot_y = torch.where(rot_y > np.pi, rot_y - 2 * np.pi, rot_y)
    rot_y = torch.where(rot_y < -np.pi, rot_y + 2 * np.pi, rot_y)
    kpoint = kps[:, :, :16]
    f = calib[:, 0, 0].view(b, 1, 1)
    cxy = torch.stack([calib[:, 0, 2], calib[:, 1, 2]], dim=1).view(b, 1, 2).repeat(1, 1, 8)
    kp_norm = (kpoint - cxy) / f
    l, h, w = dim[:, :, 2:3], dim[:, :, 1:2], dim[:, :, 0:1]
    co, sn = torch.cos(rot_y), torch.sin(rot_y)
    lc, ls, wc, ws_, hh = l * 0.5 * co, l * 0.5 * sn, w * 0.5 * co, w * 0.5 * sn, h * 0.5
    Bx = [-lc - ws_, -lc + ws_, -lc + ws_, lc + ws_, lc + ws_, lc - ws_, lc - ws_, -lc - ws_]
    By = [-hh, -hh, hh, hh, -hh, -hh, hh, hh]
    Cz = [ls - wc, ls + wc, ls + wc, -ls + wc, -ls + wc, -ls - wc, -ls - wc, ls - wc]
    Bm = torch.cat([t for pair in zip(Bx, By) for t in pair], dim=2)
    Cm = torch.cat([t for z in Cz for t in (z, z)], dim=2)
    Bm = Bm - kp_norm * Cm
    A = torch.cat([const.expand(b, cn, -1, -1), kp_norm.unsqueeze(3)], dim=3).double().view(b * cn, 16, 3)
    AT = A.transpose(1, 2)
    pinv = torch.inverse(torch.bmm(AT, A))
    pinv = torch.bmm(pinv, AT).float()
    pos = torch.bmm(pinv, Bm.reshape(b * cn, 16, 1).float()).view(b, cn, 3)
    pos = pos.clone()
    pos[:, :, 0] -= off_set.unsqueeze(1)
    return pos, rot_y, alpha


def km3d_get_bboxes(out, P2, img_hw, score_thr=0.3, nms_iou_thr=0.5, K=100, const=None):
    """heads/km3d_head.py:255-314 get_bboxes + :155-252 _decode, for every sample of the batch (the reference takes
    sample 0 only).  Returns a list of (scores[N], bbox[N,11], cls[N,1] int64)."""
    heat = _heat_nms(torch.sigmoid(out['hm']))
    hm_hp = _heat_nms(torch.sigmoid(out['hm_hp']))
    B, ncls, H, W = heat.shape
    J = 9
    ts, ti = torch.topk(heat.view(B, ncls, -1), K)
    ti = ti % (H * W)
    tys, txs = (ti // W).float(), (ti % W).float()
    sc, tk = torch.topk(ts.view(B, -1), K)
    clses = (tk // K).float()
    inds = ti.view(B, -1).gather(1, tk)
    ys, xs = tys.view(B, -1).gather(1, tk), txs.view(B, -1).gather(1, tk)
    kps = _gather_map(out['hps'], inds).clone()
    kps[..., ::2] += xs.unsqueeze(2)
    kps[..., 1::2] += ys.unsqueeze(2)
    reg = _gather_map(out['reg'], inds)
    xs_, ys_ = xs.unsqueeze(2) + reg[:, :, 0:1], ys.unsqueeze(2) + reg[:, :, 1:2]
    wh = _gather_map(out['wh'], inds)
    bboxes = torch.cat([xs_ - wh[..., 0:1] / 2, ys_ - wh[..., 1:2] / 2, xs_ + wh[..., 0:1] / 2, ys_ + wh[..., 1:2] / 2], dim=2)
    dim = _gather_map(out['dim'], inds)
    rot = _gather_map(out['rot'], inds)
    # keypoint <-> heat-map association (:205-244)
    kps = kps.view(B, K, J, 2).permute(0, 2, 1, 3).contiguous()
    hs, hi = torch.topk(hm_hp.view(B, J, -1), K)
    hi = hi % (H * W)
    hys, hxs = (hi // W).float(), (hi % W).float()
    hpo = _gather_map(out['hp_offset'], hi.view(B, -1)).view(B, J, K, 2)
    hxs, hys = hxs + hpo[..., 0], hys + hpo[..., 1]
    m = (hs > 0.1).float()
    hs = (1 - m) * -1 + m * hs
    hys = (1 - m) * (-10000) + m * hys
    hxs = (1 - m) * (-10000) + m * hxs
    hm_kps = torch.stack([hxs, hys], dim=-1).unsqueeze(2).expand(B, J, K, K, 2)
    dist = ((kps.unsqueeze(3).expand(B, J, K, K, 2) - hm_kps) ** 2).sum(dim=4) ** 0.5
    min_dist, min_ind = dist.min(dim=3)
    hsc = hs.gather(2, min_ind).unsqueeze(-1)
    sel = hm_kps.gather(3, min_ind.view(B, J, K, 1, 1).expand(B, J, K, 1, 2)).view(B, J, K, 2)
    l_, t_, r_, b_ = [bboxes[:, :, i].view(B, 1, K, 1).expand(B, J, K, 1) for i in range(4)]
    bad = (sel[..., 0:1] < l_) | (sel[..., 0:1] > r_) | (sel[..., 1:2] < t_) | (sel[..., 1:2] > b_) | (hsc < 0.1) | \
          (min_dist.unsqueeze(-1) > (torch.max(b_ - t_, r_ - l_) * 0.3))
    bad = bad.float().expand(B, J, K, 2)
    kps = ((1 - bad) * sel + bad * kps).permute(0, 2, 1, 3).contiguous().view(B, K, J * 2)
    kps = kps * 4
    bboxes = bboxes * 4
    if const is None:
        const = torch.tensor([[-1, 0], [0, -1]] * 8, dtype=torch.float32).view(1, 1, 16, 2)
    pos, rot_y, alpha = km3d_gen_position(kps, dim, rot, P2.float(), const)
    res = []
    Hh, Ww = img_hw
    for b in range(B):
        keep = sc[b] > score_thr
        p2 = P2[b].float()
        fx, fy, cx, cy, tx, ty = p2[0, 0], p2[1, 1], p2[0, 2], p2[1, 2], p2[0, 3], p2[1, 3]
        position = pos[b][keep]
        z3d = position[:, 2:3]
        cx3d = (position[:, 0:1] * fx + tx + cx * z3d) / z3d
        cy3d = (position[:, 1:2] * fy + ty + cy * z3d) / z3d
        bb = bboxes[b][keep].clone()
        bb[:, 0].clamp_(min=0); bb[:, 1].clamp_(min=0); bb[:, 2].clamp_(max=Ww); bb[:, 3].clamp_(max=Hh)
        box = torch.cat([bb, cx3d, cy3d, z3d, dim[b][keep], alpha[b][keep]], dim=1)
        s = sc[b][keep]
        k = torch.from_numpy(nms_numpy(box[:, :4].numpy(), s.numpy(), nms_iou_thr))
        res.append((s[k], box[k], clses[b][keep][k].long().unsqueeze(1)))
    return res


def deconv_bn_relu(c, p, i, x):
    """ConvTranspose2d(4x4, stride 2, padding 1, no bias) + BatchNorm2d + ReLU (detectors/KM3D_core.py:37-47); one fused HIP layer,
    output rounded once.  ``F.conv_transpose2d`` is the reference's own operator."""
    w = c.w('%s.%d.weight' % (p, 3 * i))                       # [Cin, Cout, 4, 4]
    y = F.conv_transpose2d(x, w, None, stride=2, padding=1)
    s, t = c.bn('%s.%d' % (p, 3 * i + 1))
    return c.rnd(F.relu(_affine(y, s, t)))


def km3d_resnet_neck(c, p, x):
    for i in range(3):
        x = deconv_bn_relu(c, p, i, x)
    return x


def km3d_forward(sd, cfg, img, P2, rnd=identity, return_stages=False, taps=None, stage_taps=None):
    """KM3D.test_forward (detectors/KM3D.py:61-79), B >= 1, with either core of KM3D_core.py: DLA-34 + DLA-Up, or (when the
    state_dict holds ``core.deconv_layers.0.weight``) ResNet + three transposed convolutions.  ``taps``: optional dict that
    receives (input, output) of every DCNv2 + BN + ReLU block of the up-path, keyed by its state_dict prefix.  ``stage_taps``: optional
    list receiving one record per fused operation of the whole path (see ``Ctx``)."""
    c = Ctx(sd, rnd, stage_taps)
    if taps is not None:
        c.taps = taps
    if c.has('core.deconv_layers.0.weight'):
        bb = cfg.backbone
        last = resnet(c, 'core.backbone', img.float(), depth=bb.depth, num_stages=getattr(bb, 'num_stages', 4),
                      out_indices=tuple(getattr(bb, 'out_indices', (3,))), dilations=tuple(getattr(bb, 'dilations', (1, 1, 1, 1))))[-1]
        feat = km3d_resnet_neck(c, 'core.deconv_layers', last)
    else:
        levels = dla34(c, 'core.backbone', img.float())
        feat = dla_seg_upsample(c, 'core.deconv_layers', levels)
    heads = list(cfg.head.layer_cfg.head_dict.keys())
    out = km3d_heads(c, feat, heads)
    tc = cfg.head.test_cfg
    dets = km3d_get_bboxes(out, P2, img.shape[2:], getattr(tc, 'score_thr', 0.1), getattr(tc, 'nms_iou_thr', 0.5),
                           const=c.sd.get('bbox_head.const'))
    if return_stages:
        return dets, dict(features=feat, **out)
    return dets


# ------------------------------------------------------------------------------------------- detectors
def load_priors(preprocessed_path, obj_types):
    import os
    d = os.path.join(preprocessed_path, 'training')
    mean = np.stack([np.load(os.path.join(d, 'anchor_mean_%s.npy' % t)) for t in obj_types])
    std = np.stack([np.load(os.path.join(d, 'anchor_std_%s.npy' % t)) for t in obj_types])
    return mean, std


def stereo3d_forward(sd, cfg, left, right, P2, rnd=identity, return_stages=False, stage_taps=None):
    """Stereo3D.test_forward (detectors/yolostereo3d_detector.py:77-96) generalised to B >= 1: per-sample
    post-processing identical to the reference's batch-1 path.  Returns a list of (scores, boxes, labels).
    ``stage_taps``: optional list that receives one record per fused operation (see ``Ctx``)."""
    c = Ctx(sd, rnd, stage_taps)
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
