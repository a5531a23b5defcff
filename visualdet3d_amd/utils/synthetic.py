"""Seeded synthetic KITTI-shaped workloads (no dataset, no checkpoints -- there is no network).

Everything here is generated with numpy's PCG64 so that the build container (where the golden fixtures
are produced by running the reference itself) and the GPU box regenerate bit-identical weights/inputs.

Mirrors, as data only:
  * ``cfg.detector`` of ``config/Stereo3D_example:110-167`` and ``config/Yolo3D_example:110-168``;
  * the anchor-prior ``.npy`` format written by ``scripts/imdb_precompute_3d.py:148-174``
    (``[n_scales*levels, n_ratios, 6] = [z, sin2a, cos2a, w, h, l]``; invalid = mean -100 / std 1e10);
  * KITTI calibration 000000 scaled to the network input width (pattern in
    ``visualDet3D/networks/lib/fast_utils/hill_climbing.py:128-131``).
"""
import os

import numpy as np
import torch

from .config import EasyDict


# --------------------------------------------------------------------------------------------- configs
def _anchors_cfg(obj_types, ratios):
    return EasyDict({
        'obj_types': list(obj_types),
        'pyramid_levels': [4],
        'strides': [2 ** 4],
        'sizes': [24],
        'ratios': np.array(ratios, dtype=np.float64),
        'scales': np.array([2 ** (i / 4.0) for i in range(16)]),
    })


def stereo3d_cfg(preprocessed_path, depth=34, obj_types=('Car', 'Pedestrian'), score_thr=0.75, nms_iou_thr=0.4,
                 name='Stereo3D'):
    """``cfg.detector`` for YOLOStereo3D (config/Stereo3D_example:110-167)."""
    obj_types = list(obj_types)
    feat = 1408 if depth <= 34 else 1152 + 1024
    det = EasyDict()
    det.obj_types = obj_types
    det.name = name
    det.backbone = EasyDict(depth=depth, pretrained=False, frozen_stages=-1, num_stages=3,
                            out_indices=(0, 1, 2), norm_eval=True, dilations=(1, 1, 1))
    head_loss = EasyDict(fg_iou_threshold=0.5, bg_iou_threshold=0.4, L1_regression_alpha=5 ** 2,
                         focal_loss_gamma=2.0, balance_weight=[20.0, 40.0][:len(obj_types)],
                         regression_weight=[1, 1, 1, 1, 1, 1, 12, 1, 1, 0.5, 0.5, 0.5, 1])
    head_test = EasyDict(score_thr=score_thr, cls_agnostic=False, nms_iou_thr=nms_iou_thr, post_optimization=False)
    anchors = _anchors_cfg(obj_types, [0.5, 1, 2.0])
    head_layer = EasyDict(num_features_in=feat, num_cls_output=len(obj_types) + 1, num_reg_output=12,
                          cls_feature_size=256, reg_feature_size=feat)
    det.head = EasyDict(num_regression_loss_terms=13, preprocessed_path=preprocessed_path,
                        num_classes=len(obj_types), anchors_cfg=anchors, layer_cfg=head_layer,
                        loss_cfg=head_loss, test_cfg=head_test)
    det.anchors = anchors
    det.loss = head_loss
    return det


def mono3d_cfg(preprocessed_path, depth=34, obj_types=('Car',), score_thr=0.75, nms_iou_thr=0.5,
               name='GroundAwareYolo3D', post_optimization=False):
    """``cfg.detector`` for GroundAware-Mono3D / Yolo3D (config/Yolo3D_example:110-168)."""
    obj_types = list(obj_types)
    feat = 256 if depth <= 34 else 1024
    det = EasyDict()
    det.obj_types = obj_types
    det.name = name
    det.backbone = EasyDict(depth=depth, pretrained=False, frozen_stages=-1, num_stages=3,
                            out_indices=(2,), norm_eval=False, dilations=(1, 1, 1))
    head_loss = EasyDict(fg_iou_threshold=0.5, bg_iou_threshold=0.4, L1_regression_alpha=5 ** 2,
                         focal_loss_gamma=2.0, match_low_quality=False, balance_weight=[20.0],
                         regression_weight=[1, 1, 1, 1, 1, 1, 3, 1, 1, 0.5, 0.5, 0.5, 1])
    head_test = EasyDict(score_thr=score_thr, cls_agnostic=False, nms_iou_thr=nms_iou_thr,
                         post_optimization=post_optimization)
    anchors = _anchors_cfg(obj_types, [0.5, 1])
    head_layer = EasyDict(num_features_in=feat, num_cls_output=len(obj_types) + 1, num_reg_output=12,
                          cls_feature_size=512, reg_feature_size=feat)
    det.head = EasyDict(num_regression_loss_terms=13, preprocessed_path=preprocessed_path,
                        num_classes=len(obj_types), anchors_cfg=anchors, layer_cfg=head_layer,
                        loss_cfg=head_loss, test_cfg=head_test)
    det.anchors = anchors
    det.loss = head_loss
    return det


def km3d_cfg(obj_types=('Car', 'Pedestrian', 'Cyclist'), score_thr=0.3, output_w=320):
    """``cfg.detector`` for KM3D with the DLA-34 + DLA-Up core (config/KM3D_example:126-167 head; backbone as in
    config/Monoflex_example:129-131 -- KM3DCore needs ``backbone.name``)."""
    obj_types = list(obj_types)
    det = EasyDict()
    det.obj_types = obj_types
    det.name = 'KM3D'
    det.backbone = EasyDict(name='dlanet', depth=34, pretrained=None, out_indices=(0, 1, 2, 3, 4, 5))
    head_loss = EasyDict(gamma=2.0, rampup_length=100, output_w=output_w)
    head_test = EasyDict(score_thr=score_thr)
    head_layer = EasyDict(input_features=64, head_features=256,
                          head_dict={'hm': len(obj_types), 'wh': 2, 'hps': 18, 'rot': 8, 'dim': 3, 'prob': 1, 'reg': 2,
                                     'hm_hp': 9, 'hp_offset': 2})
    det.head = EasyDict(num_classes=len(obj_types), num_joints=9, max_objects=32, layer_cfg=head_layer, loss_cfg=head_loss,
                        test_cfg=head_test)
    det.loss = head_loss
    return det


def km3d_resnet_cfg(obj_types=('Car', 'Pedestrian', 'Cyclist'), score_thr=0.3, output_w=320, depth=18):
    """``cfg.detector`` of config/KM3D_example:126-167 as shipped: ResNet-18 backbone + ConvTranspose neck, head 256 -> 64
    (``name`` added: KM3DCore reads ``backbone['name']``, KM3D_core.py:17)."""
    det = km3d_cfg(obj_types, score_thr, output_w)
    det.backbone = EasyDict(name='resnet', depth=depth, pretrained=False, frozen_stages=-1, num_stages=4, out_indices=(3,),
                            norm_eval=False, dilations=(1, 1, 1, 1))
    det.head.layer_cfg.input_features = 256
    det.head.layer_cfg.head_features = 64
    return det


def scale_km3d_head(sd, gain):
    """Scale the last (1x1) conv of every KM3D head branch in place: with head_features < input_features the fan-out init of
    ``seeded_state_dict`` saturates the heat-map (sigmoid ~ 1 everywhere -> tied scores); parity cases want spread-out scores."""
    for k in sd:
        if k.startswith('bbox_head.head_layers.') and k.endswith('.2.weight'):
            sd[k] = sd[k] * gain
    return sd


# --------------------------------------------------------------------------------------------- priors
def write_synthetic_priors(preprocessed_path, obj_types, n_ratios, n_scales=16):
    """Write ``anchor_{mean,std}_{type}.npy`` under ``<preprocessed_path>/training``.

    z-mean decreases 60 -> 5 m over the scale index (bigger anchors are nearer), std 3; sin/cos priors
    around (0, 0.3) std 0.5; dims around (1.6, 1.5, 3.9) std 0.2.  A few (scale, ratio) cells are marked
    invalid (mean -100 / std 1e10) exactly as the reference's precompute does for under-populated cells.
    """
    save_dir = os.path.join(preprocessed_path, 'training')
    os.makedirs(save_dir, exist_ok=True)
    for t, name in enumerate(obj_types):
        mean = np.zeros((n_scales, n_ratios, 6), dtype=np.float64)
        std = np.zeros((n_scales, n_ratios, 6), dtype=np.float64)
        z = np.linspace(60.0, 5.0, n_scales) * (1.0 - 0.15 * t)
        for r in range(n_ratios):
            mean[:, r, 0] = z * (1.0 + 0.05 * r)
            std[:, r, 0] = 3.0
            mean[:, r, 1] = 0.0 + 0.02 * r
            mean[:, r, 2] = 0.3 - 0.05 * t
            std[:, r, 1:3] = 0.5
            mean[:, r, 3:6] = np.array([1.6, 1.5, 3.9]) * (1.0 - 0.4 * t)
            std[:, r, 3:6] = 0.2
        # under-populated cells -> filtered by "mean z > 0" (detection_3d_head.py:239)
        bad = [(0, n_ratios - 1), (1, n_ratios - 1), (n_scales - 1, 0)]
        if t > 0:
            bad += [(s, 0) for s in range(0, n_scales, 5)]
        for s, r in bad:
            mean[s, r, 0:3] = -100.0
            std[s, r, 0:3] = 1e10
        np.save(os.path.join(save_dir, 'anchor_mean_{}.npy'.format(name)), mean)
        np.save(os.path.join(save_dir, 'anchor_std_{}.npy'.format(name)), std)
    return preprocessed_path


# --------------------------------------------------------------------------------------------- weights
def seeded_state_dict(state_dict, seed=1, head_bias=-1.0, head_std=0.006):
    """Deterministic, non-degenerate values for every entry of ``state_dict`` (name + shape driven).

    Fresh reference heads emit no detections (last convs are zero-filled, detection_3d_head.py:66-67,81-82)
    and fresh BN has trivial running stats, so parity tests randomise both:
      conv weights ~ N(0, sqrt(2/(k*k*out)))  (the reference's own init rule, resnet.py:125-131)
      BN gamma ~ U(0.8, 1.2), beta ~ N(0, 0.05), running_mean ~ N(0, 0.1), running_var ~ U(0.5, 1.5)
      last cls/reg conv weight ~ N(0, head_std); last cls conv bias = head_bias; other conv bias ~ N(0, 0.01)
    Returns a new dict of CPU fp32/int64 tensors with identical keys/shapes.
    """
    import zlib
    keys = list(state_dict.keys())
    bn_prefixes = {k[:-len('running_mean')] for k in keys if k.endswith('running_mean')}
    out = {}
    for k in keys:
        v = state_dict[k]
        shape = tuple(v.shape)
        prefix = k[:k.rfind('.') + 1]
        leaf = k[k.rfind('.') + 1:]
        if leaf == 'num_batches_tracked':
            out[k] = torch.zeros(shape, dtype=v.dtype)
            continue
        rng = np.random.default_rng([seed, zlib.crc32(k.encode())])  # per-key stream: independent of key order
        if not torch.is_floating_point(v):
            out[k] = v.detach().clone().cpu()
            continue
        if prefix in bn_prefixes:
            if leaf == 'running_mean':
                a = rng.normal(0.0, 0.1, shape)
            elif leaf == 'running_var':
                a = rng.uniform(0.5, 1.5, shape)
            elif leaf == 'weight':
                # the BN that closes a residual branch gets a small gain so ~20 stacked blocks stay O(1)
                closes_branch = prefix.endswith('bn2.') or prefix.endswith('bn3.')
                a = rng.uniform(0.25, 0.45, shape) if closes_branch else rng.uniform(0.8, 1.2, shape)
            else:
                a = rng.normal(0.0, 0.05, shape)
        elif leaf == 'weight' and len(shape) >= 3:
            is_last = k.endswith('cls_feature_extraction.6.weight') or k.endswith('reg_feature_extraction.3.weight') \
                or k.endswith('reg_feature_extraction.6.weight') or k.endswith('reg_feature_extraction.7.weight')
            if is_last and 'bbox_head' in k:
                a = rng.normal(0.0, head_std, shape)
            else:
                fan = int(np.prod(shape[2:])) * shape[0]
                gain = 6.0 if ('head_layers.' in k and k.endswith('.2.weight')) else 1.0   # KM3D 1x1 output convs
                a = rng.normal(0.0, gain * np.sqrt(2.0 / fan), shape)
        elif leaf == 'bias' and len(shape) == 1:
            if k.endswith('cls_feature_extraction.6.bias'):
                a = np.full(shape, head_bias)
            elif k.endswith('head_layers.hm.2.bias') or k.endswith('head_layers.hm_hp.2.bias'):
                a = np.full(shape, -2.19)      # CenterNet-style heat-map prior (heads/km3d_head.py:146-148)
            else:
                a = rng.normal(0.0, 0.01, shape)
        elif leaf == 'alpha':  # LookGround.alpha is 0-initialised (look_ground.py:22) -> module is a no-op
            a = np.full(shape, 0.5)
        elif leaf in ('balance_weights', 'regression_weight', 'const'):
            out[k] = v.detach().clone().cpu().float()
            continue
        else:
            a = rng.normal(0.0, 0.05, shape)
        out[k] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape))
    return out


# --------------------------------------------------------------------------------------------- inputs
def kitti_calib(width=1280, crop_top=0, batch=1):
    """KITTI 000000 P2/P3 scaled to the network input width: ``[B,3,4]`` float32 tensors."""
    s = width / 1242.0
    P2 = np.array([[721.5377 * s, 0, 609.5593 * s, 44.85728 * s],
                   [0, 721.5377 * s, (172.854 - crop_top) * s, 0.2163791 * s],
                   [0, 0, 1, 0.002745884]], dtype=np.float32)
    P3 = P2.copy()
    P3[0, 3] = -339.5242 * s
    P2 = np.repeat(P2[None], batch, axis=0)
    P3 = np.repeat(P3[None], batch, axis=0)
    return torch.from_numpy(P2), torch.from_numpy(P3)


def stereo_pair(batch=1, height=384, width=1280, seed=0):
    """Normalised-RGB-statistics stereo pair: right = left shifted by a per-row disparity (+ noise), so
    the cost volumes are non-trivial.  ``[B,3,H,W]`` float32 tensors (the collate_fn layout,
    data/kitti/dataset/stereo_dataset.py:141-157)."""
    rng = np.random.default_rng(seed)
    left = rng.standard_normal((batch, 3, height, width)).astype(np.float32)
    # mild horizontal smoothing so neighbouring disparities correlate
    left = (left + np.roll(left, 1, axis=3) + np.roll(left, -1, axis=3)) / np.float32(1.7)
    right = np.empty_like(left)
    disp = (4 + 36 * (np.arange(height) / max(height - 1, 1))).astype(np.int64)  # 4..40 px, larger near the bottom
    for y in range(height):
        right[:, :, y, :] = np.roll(left[:, :, y, :], -int(disp[y]), axis=-1)
    right += 0.05 * rng.standard_normal(right.shape).astype(np.float32)
    return torch.from_numpy(left), torch.from_numpy(right.astype(np.float32))


def mono_image(batch=1, height=384, width=1280, seed=0):
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((batch, 3, height, width)).astype(np.float32)
    img = (img + np.roll(img, 1, axis=3) + np.roll(img, -1, axis=3)) / np.float32(1.7)
    return torch.from_numpy(img)
