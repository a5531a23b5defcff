"""GPU: sweep the class-logit bias of bench.py's OTHER_CONFIGS workloads (C3: `head_bias`, C5: `hm_bias`) and print the detections per
frame each value gives, with the DCN sampling statistics of the scaled offset convs -- how the values in bench.py were chosen
(5-50 detections per frame, offsets <= 2 px rms, >= 95 % of the bilinear corners inside the image).

    python tools/calibrate_bench_workloads.py C3 head_bias -1.0 -1.5 -2.0
    python tools/calibrate_bench_workloads.py C5 head_gain 1.0 0.25 0.06
"""
import os
import sys
import tempfile

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    key, knob = sys.argv[1], sys.argv[2]            # knob: head_bias | hm_bias | head_gain | offset_scale
    values = [float(v) for v in sys.argv[3:]]
    c = dict([c for c in bench.OTHER_CONFIGS if c['key'] == key][0])
    from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT
    import visualdet3d_amd.networks.detectors  # noqa: F401
    from visualdet3d_amd.utils import synthetic as syn
    device = torch.device('cuda', 0)
    tmp = tempfile.mkdtemp()
    if c['kind'] == 'stereo':
        cfg = syn.stereo3d_cfg(tmp, depth=c['depth'], score_thr=c.get('score_thr', 0.75), nms_iou_thr=0.4)
        syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
        from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3DBaseHead
        m = Stereo3DBaseHead(cfg)
    else:
        cfg = syn.km3d_cfg(output_w=c['W'] // 4)
        m = DETECTOR_DICT[cfg.name](cfg)
    m = m.to(device).eval()
    m.compute_dtype = torch.float16 if c['dtype'] == 'fp16' else torch.bfloat16
    B, H, W = c['B'], c['H'], c['W']
    P2, _ = syn.kitti_calib(W, batch=B)
    if c['kind'] == 'stereo':
        L, R = syn.stereo_pair(B, H, W, seed=3)
        inputs = (L.to(device), R.to(device), P2.to(device))
    else:
        inputs = (syn.mono_image(B, H, W, seed=3).to(device), P2.to(device))
    for v in values:
        c[knob] = v
        bench.prepare_other_config(c, m, inputs)
        with torch.no_grad():
            out = m.forward_device(*inputs)
        counts = out[-1].cpu().reshape(-1).tolist()
        bench.profile_ops(m, inputs, reps=1)
        print('%s %s %+.4f: detections per frame %s (mean %.1f)  dcn %s' % (key, knob, v, [int(x) for x in counts], sum(counts) / len(counts), bench.profile_ops.dcn_sampling), flush=True)


if __name__ == '__main__':
    main()
