import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the read-only reference tree (build container only)")


def pytest_collection_modifyitems(config, items):
    import torch
    has_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(skip_gpu)


# BASELINE config 2 exactly as bench.py builds it (Stereo3D R34, score_thr 0.75, nms 0.4, seed-1 weights, 8 pairs of 384 x 1280)
# together with ONE run of the bf16-rounded oracle on it, with stage taps -- shared by the at-size parity tests (the oracle run is
# ~20 s of host time; the taps hold every activation of the network in fp32, ~3 GB of host memory).
_C2_CACHE = {}


def c2_bf16_case():
    """-> dict(model, sd, cfg, L, R, P2, ref (oracle detections), stages, taps).  Built on first use, kept for the session."""
    if _C2_CACHE:
        return _C2_CACHE
    import tempfile

    import torch

    from oracle import detector_oracle as orc
    from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3D
    from visualdet3d_amd.utils import synthetic as syn
    tmp = tempfile.mkdtemp()
    cfg = syn.stereo3d_cfg(tmp, depth=34, score_thr=0.75, nms_iou_thr=0.4)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    m = Stereo3D(cfg)
    sd = syn.seeded_state_dict(m.state_dict(), seed=1, head_std=0.00042)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = torch.bfloat16
    B, H, W = 8, 384, 1280
    L, R = syn.stereo_pair(B, H, W, seed=100)
    P2, _ = syn.kitti_calib(W, batch=B)
    torch.set_num_threads(min(64, torch.get_num_threads()))
    taps = []
    with torch.no_grad():
        ref, st = orc.stereo3d_forward(sd, cfg, L, R, P2, rnd=orc.bf16_round, return_stages=True, stage_taps=taps)
    _C2_CACHE.update(model=m, sd=sd, cfg=cfg, L=L, R=R, P2=P2, ref=ref, stages=st, taps=taps, B=B, H=H, W=W)
    return _C2_CACHE
