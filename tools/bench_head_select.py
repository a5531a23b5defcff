"""Micro-benchmark of the head's candidate selection (vd3d_head_select) and NMS stage on config 3's shape (32 frames x 69 120 anchors) and config 2's
(8 x 92 160) for different candidate densities:  python tools/bench_head_select.py"""
import sys
import tempfile

import torch

sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops  # noqa: E402
from visualdet3d_amd.networks.heads.anchors import Anchors  # noqa: E402
from visualdet3d_amd.utils import synthetic as syn  # noqa: E402

tmp = tempfile.mkdtemp()
cfg = syn.stereo3d_cfg(tmp, depth=34)
syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
anc = Anchors(preprocessed_path=cfg.head.preprocessed_path, readConfigFile=True, **cfg.head.anchors_cfg)
for B, H, W in ((32, 288, 1280), (8, 384, 1280), (1, 384, 1280)):
    anchors, prior, A = anc.device_tables((H, W), torch.device('cuda'))
    N = anchors.shape[0]
    P2, _ = syn.kitti_calib(W, batch=B)
    P2 = P2.cuda().contiguous()
    ws = torch.empty(ops._lib.lib().vd3d_head_workspace_bytes(B, 4096), dtype=torch.uint8, device='cuda')
    for bias in (-9.0, -3.0, -2.0, -1.0):
        g = torch.Generator(device='cuda').manual_seed(0)
        cls = torch.randn((B, N, 3), generator=g, device='cuda') + bias
        reg = torch.randn((B, N, 12), generator=g, device='cuda') * 0.1
        args = (anchors, prior, P2, A, 2, int(prior.shape[1]), (H, W), 0.5, 0.4)
        for use_filter in (True, False):
            kw = dict(use_filter=use_filter, max_cand=4096, workspace=ws)
            for _ in range(3):
                ops.head_select(cls, *args, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.head_select(cls, *args, **kw)
            e.record()
            torch.cuda.synchronize()
            out = ops.head_postprocess(cls, reg, *args, **kw)
            torch.cuda.synchronize()
            s2, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s2.record()
            for _ in range(10):
                ops.head_postprocess(cls, reg, *args, **kw)
            e2.record()
            torch.cuda.synchronize()
            cnt = out[4].tolist()
            print('B %2d N %6d bias %5.1f filter %d: select (+ zero_counts) %7.1f us, select + nms %7.1f us, detections/frame %s' %
                  (B, N, bias, use_filter, s.elapsed_time(e) / 20 * 1e3, s2.elapsed_time(e2) / 10 * 1e3, cnt[:4]))
