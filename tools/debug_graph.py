import sys, tempfile, torch
sys.path.insert(0, '.')
from visualdet3d_amd.networks.detectors.yolostereo3d_detector import Stereo3D
from visualdet3d_amd.utils import synthetic as syn
from visualdet3d_amd import hip_ops as ops
tmp = tempfile.mkdtemp(); cfg = syn.stereo3d_cfg(tmp); syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
m = Stereo3D(cfg); m.load_state_dict(syn.seeded_state_dict(m.state_dict(), seed=1, head_std=0.00042)); m = m.cuda().eval()
m.compute_dtype = torch.bfloat16
H, W, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L, R = syn.stereo_pair(B, H, W, seed=1); P2, _ = syn.kitti_calib(W, batch=B)
L, R, P2 = L.cuda(), R.cuda(), P2.cuda()
def capture(fn, name):
    with torch.no_grad():
        for _ in range(2): out = fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        torch.cuda.synchronize()
        print(name, 'captured', flush=True)
        import os
        for i in range(int(os.environ.get('REPLAYS', '3'))):
            g.replay()
            if os.environ.get('SYNC_EACH', '1') == '1': torch.cuda.synchronize()
            print(' replay', i, flush=True)
        torch.cuda.synchronize()
        print(name, 'replayed OK', flush=True)
    return out
which = sys.argv[4]
if which == 'backbone':
    capture(lambda: m.core.backbone.forward_nhwc(torch.cat([L, R], 0), torch.bfloat16), 'backbone')
elif which == 'stem':
    bb = m.core.backbone
    pc = ops.pack_stem_conv(bb.conv1.weight, (bb.bn1.weight, bb.bn1.bias, bb.bn1.running_mean, bb.bn1.running_var, 1e-5), torch.bfloat16)
    capture(lambda: ops.stem_conv(L, pc, torch.bfloat16), 'stem')
elif which == 'core':
    capture(lambda: m.core.forward_nhwc(L, R, torch.bfloat16), 'core')
elif which == 'all':
    capture(lambda: m.forward_device(L, R, P2), 'all')
