"""Race / determinism check on the GPU: the same batch through the whole Stereo3D, GroundAware and KM3D paths many times (eager and
hipGraph replay) must give bit-identical raw maps and detections every time.  Usage: python tools/stress.py [iters]"""
import sys
import tempfile

import torch

sys.path.insert(0, '.')
from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT  # noqa: E402
import visualdet3d_amd.networks.detectors  # noqa: E402,F401
from visualdet3d_amd.utils import synthetic as syn  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def flat(out, raw):
    """Raw head maps in full + the padded detections up to each sample's count (rows beyond it are uninitialised by design)."""
    count = out[-1]
    K = out[0].shape[1]
    valid = torch.arange(K, device=count.device)[None, :] < count[:, None]
    ts = [count.float()]
    for t in out[:-1]:
        m = valid if t.dim() == 2 else valid[..., None].expand_as(t)
        ts.append(torch.where(m, t.float(), torch.zeros_like(t, dtype=torch.float32)).reshape(-1))
    ts += [t.float().reshape(-1) for t in (raw.values() if isinstance(raw, dict) else raw)]
    return torch.cat(ts)


def run(name, m, inputs):
    with torch.no_grad():
        ref = None
        bad = 0
        for i in range(iters):
            out = m.forward_device(*inputs)
            cur = flat(out, m._last_raw)
            if ref is None:
                ref = cur.clone()
            elif not torch.equal(cur, ref):
                bad += 1
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.forward_device(*inputs)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            out = m.forward_device(*inputs)
        gbad = 0
        for i in range(iters):
            g.replay()
            cur = flat(out, m._last_raw)
            if not torch.equal(cur, ref):
                gbad += 1
    print('%-28s eager mismatches %d / %d   graph mismatches %d / %d   (%d values)' % (name, bad, iters - 1, gbad, iters, ref.numel()), flush=True)
    return bad + gbad


def main():
    tmp = tempfile.mkdtemp()
    total = 0
    cfg = syn.stereo3d_cfg(tmp, depth=34, score_thr=0.75, nms_iou_thr=0.4)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    m = DETECTOR_DICT[cfg.name](cfg)
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), seed=1, head_std=0.00042))
    m = m.cuda().eval()
    m.compute_dtype = torch.bfloat16
    L, R = syn.stereo_pair(8, 384, 1280, seed=3)
    P2, _ = syn.kitti_calib(1280, batch=8)
    total += run('Stereo3D R34 384x1280 B=8', m, (L.cuda(), R.cuda(), P2.cuda()))
    del m
    tmp2 = tempfile.mkdtemp()
    cfg = syn.mono3d_cfg(tmp2, depth=34, score_thr=0.75, name='Yolo3D')
    syn.write_synthetic_priors(tmp2, cfg.obj_types, 2)
    m = DETECTOR_DICT[cfg.name](cfg)
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), seed=2, head_std=0.0005))
    m = m.cuda().eval()
    m.compute_dtype = torch.bfloat16
    P2, _ = syn.kitti_calib(1280, batch=4)
    total += run('Yolo3D (DCN) 384x1280 B=4', m, (syn.mono_image(4, 384, 1280, seed=5).cuda(), P2.cuda()))
    del m
    cfg = syn.km3d_cfg(output_w=1280 // 4)
    m = DETECTOR_DICT[cfg.name](cfg)
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), seed=3, head_std=0.0005))
    m = m.cuda().eval()
    m.compute_dtype = torch.bfloat16
    total += run('KM3D DLA-34 384x1280 B=4', m, (syn.mono_image(4, 384, 1280, seed=6).cuda(), P2.cuda()))
    print('TOTAL mismatches', total)
    sys.exit(1 if total else 0)


if __name__ == '__main__':
    main()
