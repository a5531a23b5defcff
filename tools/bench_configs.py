"""Throughput of the other BASELINE.json configurations on one MI355X (hipGraph replay, bf16, synthetic inputs resident in HBM,
random-init weights): numbers for DESIGN.md, not the driver's bench line.
    python tools/bench_configs.py [names...]"""
import sys
import tempfile
import time

import torch

sys.path.insert(0, '.')
from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT  # noqa: E402
import visualdet3d_amd.networks.detectors  # noqa: E402,F401
from visualdet3d_amd.utils import synthetic as syn  # noqa: E402

CONFIGS = {
    'C1 GroundAwareYolo3D R34 384x1280 B=1': dict(kind='mono', name='GroundAwareYolo3D', depth=34, H=384, W=1280, B=1, gf=82.10),
    'C1 GroundAwareYolo3D R34 384x1280 B=16': dict(kind='mono', name='GroundAwareYolo3D', depth=34, H=384, W=1280, B=16, gf=82.10),
    'Yolo3D (DCN head) R34 384x1280 B=16': dict(kind='mono', name='Yolo3D', depth=34, H=384, W=1280, B=16, gf=None),
    'C3 Stereo3D R50 288x1280 B=16 (StereoHead)': dict(kind='stereo', depth=50, H=288, W=1280, B=16, gf=593.55),
    'C3 Stereo3D R50 + base DCNv2 head 288x1280 B=32 (as BASELINE states it)': dict(kind='stereo', depth=50, H=288, W=1280, B=32, gf=472.0,
                                                                                     dcn_head=True),
    'C2 Stereo3D R34 384x1280 B=8': dict(kind='stereo', depth=34, H=384, W=1280, B=8, gf=473.82),
    'C5 KM3D DLA-34 512x1760 B=16': dict(kind='km3d', H=512, W=1760, B=16, gf=326.18),
    'C5 KM3D DLA-34 512x1760 B=16 fp16 (as BASELINE states it)': dict(kind='km3d', H=512, W=1760, B=16, gf=326.18, dtype='fp16'),
}


def build(c):
    tmp = tempfile.mkdtemp()
    if c['kind'] == 'mono':
        cfg = syn.mono3d_cfg(tmp, depth=c['depth'], score_thr=0.75, name=c['name'])
        syn.write_synthetic_priors(tmp, cfg.obj_types, 2)
    elif c['kind'] == 'stereo':
        cfg = syn.stereo3d_cfg(tmp, depth=c['depth'], score_thr=0.75, nms_iou_thr=0.4)
        syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    else:
        cfg = syn.km3d_cfg(output_w=c['W'] // 4)
    if c.get('dcn_head'):
        from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3DBaseHead
        m = Stereo3DBaseHead(cfg)
    else:
        m = DETECTOR_DICT[cfg.name](cfg)
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), seed=1, head_std=0.0005))
    m = m.cuda().eval()
    m.compute_dtype = torch.float16 if c.get('dtype') == 'fp16' else torch.bfloat16
    B, H, W = c['B'], c['H'], c['W']
    P2, _ = syn.kitti_calib(W, batch=B)
    if c['kind'] == 'stereo':
        L, R = syn.stereo_pair(B, H, W, seed=3)
        inputs = (L.cuda(), R.cuda(), P2.cuda())
    else:
        inputs = (syn.mono_image(B, H, W, seed=3).cuda(), P2.cuda())
    return m, inputs


def main():
    eager = '--eager' in sys.argv          # no hipGraph: every dispatch visible to a rocprofv3 --pmc pass, 3 forwards
    only = [a for a in sys.argv[1:] if a != '--eager']
    for name, c in CONFIGS.items():
        if only and not any(o in name for o in only):
            continue
        m, inputs = build(c)
        with torch.no_grad():
            for _ in range(2):
                m.forward_device(*inputs)
            torch.cuda.synchronize()
            if eager:
                for _ in range(3):
                    m.forward_device(*inputs)
                torch.cuda.synchronize()
                print('%-76s eager: 3 forwards' % name, flush=True)
                continue
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                m.forward_device(*inputs)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = m.forward_device(*inputs)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        n, t0 = 10, time.perf_counter()
        for _ in range(n):
            g.replay()
            out[-1].cpu()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        rate = c['B'] / dt
        extra = '  %.0f TF/s whole path (%.1f %% of 2.5 PF)' % (rate * c['gf'] / 1e3, rate * c['gf'] / 25e3) if c['gf'] else ''
        print('%-76s %8.2f ms/step %9.1f img/s%s' % (name, dt * 1e3, rate, extra), flush=True)
        del m, g, out, inputs
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
