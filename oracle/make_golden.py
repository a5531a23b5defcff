"""ORACLE tooling (build container only): run the REFERENCE ITSELF on CPU (through oracle/ref_shim.py) on
seeded synthetic inputs and commit its outputs as golden fixtures under tests/golden/.

    python -m oracle.make_golden            # regenerates every fixture (needs /root/reference)

The fixtures pin (a) the restatement in oracle/detector_oracle.py and (b) -- on the GPU box, where the
reference tree does not exist -- the HIP path, against the reference's own results.
Weights/inputs are regenerated from seeds by visualdet3d_amd.utils.synthetic, so only outputs are stored.
"""
import os
import sys
import tempfile

import numpy as np
import torch

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)

from oracle import ref_shim  # noqa: E402
from visualdet3d_amd.utils import synthetic as syn  # noqa: E402

GOLDEN_DIR = os.path.join(_REPO, 'tests', 'golden')

# name -> (detector, depth, H, W, batch-of-frames (each run at B=1 through the reference), weight seed, input seed, score_thr)
STEREO_CASES = {
    'stereo3d_r34_96x320': dict(depth=34, H=96, W=320, frames=2, wseed=1, iseed=3, score_thr=0.5, head_std=0.0005),
    'stereo3d_r34_384x1280': dict(depth=34, H=384, W=1280, frames=2, wseed=1, iseed=0, score_thr=0.75, head_std=0.00042),
    'stereo3d_r50_96x320': dict(depth=50, H=96, W=320, frames=1, wseed=5, iseed=8, score_thr=0.5, head_std=0.006),
    'stereo3d_r34_384x1280_thr06': dict(depth=34, H=384, W=1280, frames=1, wseed=2, iseed=5, score_thr=0.6, head_std=0.009),
    # BASELINE config 3 as specified: ResNet-50 stereo core + the BASE (DCNv2) head, 288 x 1280 (Stereo3D with
    # build_head overridden to AnchorBasedDetection3DHead); the reference's CUDA-only DCN is served by oracle/dcn_ref.py
    # config/Stereo3D_example:114-122 at its SHIPPED crop size (cropSize 288 x 1280, ResNet-34, StereoHead)
    'stereo3d_r34_288x1280': dict(depth=34, H=288, W=1280, frames=1, wseed=1, iseed=16, score_thr=0.75, head_std=0.0005),
    'stereo3d_r50_dcn_288x1280': dict(depth=50, H=288, W=1280, frames=1, wseed=6, iseed=9, score_thr=0.5, head_std=0.006, dcn_head=True),
}


def build_reference_stereo(case, tmp):
    DD = ref_shim.detector_dict()
    cfg = syn.stereo3d_cfg(tmp, depth=case['depth'], score_thr=case['score_thr'])
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    cls = DD['Stereo3D']
    if case.get('dcn_head'):
        import visualDet3D.networks.lib.ops.dcn.deform_conv as ref_dcn
        from visualDet3D.networks.heads.detection_3d_head import AnchorBasedDetection3DHead as RefBaseHead
        from oracle import dcn_ref
        ref_dcn.modulated_deform_conv = lambda x, off, m, w, b, s, p, d, g, dg: dcn_ref.deform_conv_forward(x, off, m, w, b, s, p, d, g, dg)

        class Stereo3DBaseHead(cls):          # yolostereo3d_detector.py:35-38 overridden, nothing else
            def build_head(self, network_cfg):
                self.bbox_head = RefBaseHead(**(network_cfg.head))
        cls = Stereo3DBaseHead
    model = cls(cfg).eval()
    sd = syn.seeded_state_dict(model.state_dict(), seed=case['wseed'], head_std=case['head_std'])
    model.load_state_dict(sd)
    return model, cfg, sd


def subsample(t, n=4096):
    """Deterministic strided sample of a tensor (keeps fixtures small)."""
    flat = t.reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n].clone()


def run_stereo_case(name, case):
    tmp = tempfile.mkdtemp()
    model, cfg, sd = build_reference_stereo(case, tmp)
    L, R = syn.stereo_pair(case['frames'], case['H'], case['W'], seed=case['iseed'])
    P2, P3 = syn.kitti_calib(case['W'], batch=case['frames'])
    out = {}
    with torch.no_grad():
        for f in range(case['frames']):
            l, r, p2, p3 = L[f:f + 1], R[f:f + 1], P2[f:f + 1].clone(), P3[f:f + 1].clone()
            # stage taps, straight from the reference modules
            core = model.core(torch.cat([l, r], dim=1))
            feats = core['features']
            cls_preds, reg_preds = model.bbox_head(dict(features=feats, P2=p2, image=l))
            model.bbox_head.anchors.P2 = None  # defeat the per-P2 cache so every frame recomputes its mask
            scores, boxes, labels = model([l, r, p2, p3])
            out['f%d_scores' % f] = scores.numpy()
            out['f%d_boxes' % f] = boxes.numpy()
            out['f%d_labels' % f] = labels.numpy()
            out['f%d_features_sub' % f] = subsample(feats).numpy()
            out['f%d_cls_sub' % f] = subsample(cls_preds).numpy()
            out['f%d_reg_sub' % f] = subsample(reg_preds).numpy()
            out['f%d_mask_sum' % f] = np.int64(model.bbox_head.anchors.useful_mask.sum().item())
            print(name, 'frame', f, 'detections', len(scores), 'labels', np.bincount(labels.numpy(), minlength=2))
    out['meta'] = np.array([case['depth'], case['H'], case['W'], case['frames'], case['wseed'], case['iseed']])
    out['score_thr'] = np.float32(case['score_thr'])
    out['head_std'] = np.float64(case['head_std'])
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + '.npz'), **out)
    return model, cfg, sd, (L, R, P2, P3), out


MONO_CASES = {
    'groundaware_r34_96x320': dict(name='GroundAwareYolo3D', depth=34, H=96, W=320, frames=2, wseed=3, iseed=4, score_thr=0.5, head_std=0.02),
    'groundaware_r34_384x1280': dict(name='GroundAwareYolo3D', depth=34, H=384, W=1280, frames=1, wseed=3, iseed=6, score_thr=0.75, head_std=0.02),
    'yolo3d_dcn_r34_96x320': dict(name='Yolo3D', depth=34, H=96, W=320, frames=2, wseed=4, iseed=7, score_thr=0.5, head_std=0.02),
    # config/Yolo3D_example:113-136 AS SHIPPED: GroundAwareYolo3D on ResNet-101, 288 x 1280 crop, score 0.75, nms 0.5, post_optimization on
    'groundaware_r101_288x1280_postopt': dict(name='GroundAwareYolo3D', depth=101, H=288, W=1280, frames=1, wseed=8, iseed=15, score_thr=0.75,
                                              head_std=0.03, post_optimization=True),
}


def build_reference_mono(case, tmp):
    DD = ref_shim.detector_dict()
    if case['name'] == 'Yolo3D':
        # the reference's DCN is CUDA-only: serve its forward with the oracle restatement (pinned to the reference's own
        # im2col device code through tests/golden/dcn_cases.npz)
        import visualDet3D.networks.lib.ops.dcn.deform_conv as ref_dcn
        from oracle import dcn_ref
        ref_dcn.modulated_deform_conv = lambda x, off, m, w, b, s, p, d, g, dg: dcn_ref.deform_conv_forward(x, off, m, w, b, s, p, d, g, dg)
    cfg = syn.mono3d_cfg(tmp, depth=case['depth'], score_thr=case['score_thr'], name=case['name'],
                         post_optimization=case.get('post_optimization', False))
    syn.write_synthetic_priors(tmp, cfg.obj_types, 2)
    model = DD[case['name']](cfg).eval()
    sd = syn.seeded_state_dict(model.state_dict(), seed=case['wseed'], head_std=case['head_std'])
    model.load_state_dict(sd)
    return model, cfg, sd


def run_mono_case(name, case):
    tmp = tempfile.mkdtemp()
    model, cfg, sd = build_reference_mono(case, tmp)
    img = syn.mono_image(case['frames'], case['H'], case['W'], seed=case['iseed'])
    P2, _ = syn.kitti_calib(case['W'], batch=case['frames'])
    out = {}
    with torch.no_grad():
        for f in range(case['frames']):
            im, p2 = img[f:f + 1], P2[f:f + 1].clone()
            feats = model.core(dict(image=im, P2=p2))
            cls_preds, reg_preds = model.bbox_head(dict(features=feats, P2=p2))
            model.bbox_head.anchors.P2 = None
            scores, boxes, labels = model([im, p2])
            out['f%d_scores' % f] = scores.numpy()
            out['f%d_boxes' % f] = boxes.numpy()
            out['f%d_labels' % f] = labels.numpy()
            out['f%d_features_sub' % f] = subsample(feats).numpy()
            out['f%d_cls_sub' % f] = subsample(cls_preds).numpy()
            out['f%d_reg_sub' % f] = subsample(reg_preds).numpy()
            print(name, 'frame', f, 'detections', len(scores), 'cls std', float(cls_preds.std()), 'reg std', float(reg_preds.std()))
    out['meta'] = np.array([case['depth'], case['H'], case['W'], case['frames'], case['wseed'], case['iseed']])
    out['score_thr'] = np.float32(case['score_thr'])
    out['head_std'] = np.float64(case['head_std'])
    out['post_optimization'] = np.int64(bool(case.get('post_optimization', False)))
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + '.npz'), **out)


KM3D_CASES = {
    'km3d_dla34_96x320': dict(H=96, W=320, frames=2, wseed=7, iseed=11, score_thr=0.3),
    'km3d_dla34_192x640': dict(H=192, W=640, frames=1, wseed=7, iseed=12, score_thr=0.3),
    'km3d_dla34_512x1760': dict(H=512, W=1760, frames=1, wseed=7, iseed=13, score_thr=0.3),      # BASELINE config 5 at size
    'km3d_res18_192x640': dict(H=192, W=640, frames=2, wseed=9, iseed=14, score_thr=0.3, resnet=18, head_gain=0.3),   # config/KM3D_example's core
}


def build_reference_km3d(case, tmp):
    DD = ref_shim.detector_dict()
    import visualDet3D.networks.lib.ops.dcn.deform_conv as ref_dcn
    import visualDet3D.networks.backbones.dla as ref_dla
    from oracle import dcn_ref
    ref_dcn.modulated_deform_conv = lambda x, off, m, w, b, s, p, d, g, dg: dcn_ref.deform_conv_forward(x, off, m, w, b, s, p, d, g, dg)
    ref_dla.DLA.load_pretrained_model = lambda self, *a, **k: None
    if case.get('resnet'):
        cfg = syn.km3d_resnet_cfg(score_thr=case['score_thr'], output_w=case['W'] // 4, depth=case['resnet'])
    else:
        cfg = syn.km3d_cfg(score_thr=case['score_thr'], output_w=case['W'] // 4)
    model = DD['KM3D'](cfg).eval()
    sd = syn.seeded_state_dict(model.state_dict(), seed=case['wseed'])
    if case.get('head_gain'):
        syn.scale_km3d_head(sd, case['head_gain'])
    model.load_state_dict(sd)
    return model, cfg, sd


def run_km3d_case(name, case):
    tmp = tempfile.mkdtemp()
    model, cfg, sd = build_reference_km3d(case, tmp)
    print(name, 'state_dict entries', len(sd), sum(v.numel() for v in sd.values()))
    img = syn.mono_image(case['frames'], case['H'], case['W'], seed=case['iseed'])
    P2, _ = syn.kitti_calib(case['W'], batch=case['frames'])
    out = {}
    torch.manual_seed(0)   # gen_position adds randn * 1e-8 before the 3x3 inverse (rtm3d_utils.py:447)
    with torch.no_grad():
        for f in range(case['frames']):
            im, p2 = img[f:f + 1], P2[f:f + 1].clone()
            feats = model.core(dict(image=im, P2=p2))
            maps = model.bbox_head(feats)
            for h, v in maps.items():
                out['f%d_%s_sub' % (f, h)] = subsample(v).numpy()
            scores, boxes, labels = model([im, p2])
            out['f%d_scores' % f] = scores.numpy()
            out['f%d_boxes' % f] = boxes.numpy()
            out['f%d_labels' % f] = labels.numpy()
            out['f%d_features_sub' % f] = subsample(feats).numpy()
            print(name, 'frame', f, 'detections', len(scores), 'hm max', float(torch.sigmoid(maps['hm']).max()), 'feat std', float(feats.std()))
    out['meta'] = np.array([case.get('resnet', 34), case['H'], case['W'], case['frames'], case['wseed'], case['iseed']])
    out['score_thr'] = np.float32(case['score_thr'])
    out['head_gain'] = np.float64(case.get('head_gain', 1.0))
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + '.npz'), **out)


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    only__ = sys.argv[1:] or None
    for name, case in KM3D_CASES.items():
        if only__ and name not in only__:
            continue
        run_km3d_case(name, case)
    only_ = sys.argv[1:] or None
    for name, case in MONO_CASES.items():
        if only_ and name not in only_:
            continue
        run_mono_case(name, case)
    only = sys.argv[1:] or None
    for name, case in STEREO_CASES.items():
        if only and name not in only:
            continue
        run_stereo_case(name, case)


if __name__ == '__main__':
    main()
