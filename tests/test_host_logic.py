"""CPU: host-side pieces of the boundary -- registry, config loader, anchors tables vs the oracle restatement,
checkpoint key layout, weight packing helpers (CPU-only parts)."""
import tempfile

import numpy as np
import pytest
import torch

from oracle import detector_oracle as orc
from visualdet3d_amd.utils import synthetic as syn


def test_registry_contract():
    from visualdet3d_amd.networks.utils.registry import DETECTOR_DICT, BACKBONE_DICT, Registry
    import visualdet3d_amd.networks  # noqa: F401
    for name in ('Stereo3D', 'Yolo3D', 'GroundAwareYolo3D'):
        assert name in DETECTOR_DICT
    assert 'resnet' in BACKBONE_DICT
    r = Registry('t')

    @r.register_module
    def f():
        pass
    assert r['f'] is f
    with pytest.raises(KeyError):
        r.register_module(f)
    r._register_module(f, force=True)
    with pytest.raises(TypeError):
        r._register_module(3)


def test_cfg_from_file(tmp_path):
    from visualdet3d_amd.utils import cfg_from_file, EasyDict
    p = tmp_path / 'cfg.py'
    p.write_text("from visualdet3d_amd.utils import EasyDict as edict\ncfg = edict()\ncfg.obj_types=['Car']\ncfg.detector = edict(name='Stereo3D', head=dict(a=1))\n")
    cfg = cfg_from_file(str(p))
    assert cfg.detector.name == 'Stereo3D' and cfg.detector.head.a == 1 and isinstance(cfg.detector.head, EasyDict)


@pytest.mark.parametrize('H,W', [(96, 320), (384, 1280), (288, 1280)])
def test_anchor_tables_match_oracle(H, W):
    from visualdet3d_amd.networks.heads.anchors import Anchors
    tmp = tempfile.mkdtemp()
    cfg = syn.stereo3d_cfg(tmp)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    a = Anchors(preprocessed_path=tmp, readConfigFile=True, **cfg.head.anchors_cfg)
    anchors, prior, A = a.device_tables((H, W), 'cpu')
    mean_npy, std_npy = orc.load_priors(tmp, cfg.obj_types)
    want_anchors, means, mean_std = orc.anchors_for_image(H, W, cfg.head.anchors_cfg, mean_npy, std_npy)
    assert A == 48 and torch.equal(anchors, want_anchors)
    N = anchors.shape[0]
    assert N == (H // 16) * (W // 16) * 48
    assert torch.equal(prior[torch.arange(N) % A], mean_std)
    # reference-compatible forward: same mask as the oracle
    img = torch.zeros(2, 3, H, W)
    P2, _ = syn.kitti_calib(W, batch=2)
    P2[1, 1, 2] += 11.0
    anc, mask, ms = a(img, P2, is_filtering=True)
    assert torch.equal(mask, orc.anchor_mask(want_anchors, means, P2))
    assert torch.equal(ms, mean_std) and anc.shape == (1, N, 4)


def test_checkpoint_key_layouts():
    from visualdet3d_amd.networks.detectors import GroundAwareYolo3D, Stereo3D, Yolo3D
    tmp = tempfile.mkdtemp()
    cfg = syn.stereo3d_cfg(tmp)
    syn.write_synthetic_priors(tmp, cfg.obj_types, 3)
    sd = Stereo3D(cfg).state_dict()
    assert len(sd) == 313 and sum(v.numel() for v in sd.values()) == 107605442   # reference Stereo3D-R34
    mcfg = syn.mono3d_cfg(tmp)
    syn.write_synthetic_priors(tmp, mcfg.obj_types, 2)
    keys = set(GroundAwareYolo3D(mcfg).state_dict())
    for k in ('bbox_head.reg_feature_extraction.0.disp_create.0.weight', 'bbox_head.reg_feature_extraction.0.extract.bias',
              'bbox_head.reg_feature_extraction.0.alpha', 'bbox_head.reg_feature_extraction.7.weight', 'core.backbone.layer3.5.bn2.running_var'):
        assert k in keys, k
    keys = set(Yolo3D(syn.mono3d_cfg(tmp, name='Yolo3D')).state_dict())
    for k in ('bbox_head.reg_feature_extraction.0.weight', 'bbox_head.reg_feature_extraction.0.conv_offset.bias',
              'bbox_head.reg_feature_extraction.1.running_mean', 'bbox_head.reg_feature_extraction.6.bias'):
        assert k in keys, k


def test_dcn_pack_legacy_offset_key_remap():
    # version < 2 checkpoints name the offset conv "<prefix>_offset.*" (deform_conv.py:468-489)
    from visualdet3d_amd.networks.lib.ops.dcn.deform_conv import _remap_legacy_offset_keys
    sd = {'m.conv_offset.weight': 1, 'm.conv_offset.bias': 2, 'm.conv.weight': 3}
    _remap_legacy_offset_keys(sd, 'm.conv.', {})
    assert sd == {'m.conv.conv_offset.weight': 1, 'm.conv.conv_offset.bias': 2, 'm.conv.weight': 3}
    sd2 = {'m.conv_offset.weight': 1}
    _remap_legacy_offset_keys(sd2, 'm.conv.', {'version': 2})
    assert sd2 == {'m.conv_offset.weight': 1}


def test_nms_oracle_semantics():
    from oracle.nms_ref import nms_numpy
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10]], dtype=np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.9], dtype=np.float32)
    assert nms_numpy(boxes, scores, 0.5).tolist() == [0, 2]          # tie -> lower index first, duplicates suppressed
    assert nms_numpy(boxes, scores, 0.99).tolist() == [0, 1, 2]      # identical box still suppressed (IoU 1 > .99)
    assert nms_numpy(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.5).shape == (0,)


def test_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under visualdet3d_amd/ may import, open or exec it (and bench.py only in its
    cpu_baseline leg, __graft_entry__ only in smoke())."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r'^\s*(from|import)\s+oracle\b|[\'"]oracle[/\'"]', re.M)
    for dp, _, fs in os.walk(os.path.join(root, 'visualdet3d_amd')):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dp, f), errors='ignore').read()
                assert not pat.search(src), os.path.join(dp, f)
    bench = open(os.path.join(root, 'bench.py')).read()
    uses = [m.start() for m in re.finditer(r'from oracle|import oracle', bench)]
    assert uses and all(bench.rfind('def ', 0, u) == bench.find('def cpu_baseline') for u in uses)


def test_weight_register_images_match_the_layout_in_vd3d_h():
    """Host-side packing of the MFMA register images the resident / point-wise / base-layer kernels load verbatim
    (``vd3d_conv_params.weight_frag`` and ``vd3d_image_conv7x7`` in include/vd3d.h): every element against the documented index map."""
    import random
    import torch
    from visualdet3d_amd import hip_ops as ops
    rnd = random.Random(3)
    g = torch.Generator().manual_seed(3)
    # 3x3 / s1 / p1: [Cout/32][Cin/64][tap*4 + ks][lane][8]  <-  w[32 nb + (l & 31)][tap][64 kc + (2 ks + (l >> 5)) 8 + e]
    w = torch.randn(64, 128, 3, 3, generator=g)
    pc = ops.pack_conv(w, None, None, torch.float16, 1, 1, 1)
    fr = pc.w_frag.reshape(2, 2, 36, 64, 8).float()
    wh = w.half().float()
    for _ in range(300):
        nb, kc, f, l, e = rnd.randrange(2), rnd.randrange(2), rnd.randrange(36), rnd.randrange(64), rnd.randrange(8)
        tap, ks = f // 4, f % 4
        c = 64 * kc + (2 * ks + (l >> 5)) * 8 + e
        assert fr[nb, kc, f, l, e] == wh[32 * nb + (l & 31), c, tap // 3, tap % 3]
    # 1x1 / s1 / p0 (point-wise streaming kernel): the same image with one tap
    w1 = torch.randn(256, 128, 1, 1, generator=g)
    pc1 = ops.pack_conv(w1, None, None, torch.bfloat16, 1, 0, 1)
    fr1 = pc1.w_frag.reshape(8, 2, 4, 64, 8).float()
    w1b = w1.bfloat16().float()
    for _ in range(300):
        nb, kc, ks, l, e = rnd.randrange(8), rnd.randrange(2), rnd.randrange(4), rnd.randrange(64), rnd.randrange(8)
        assert fr1[nb, kc, ks, l, e] == w1b[32 * nb + (l & 31), 64 * kc + (2 * ks + (l >> 5)) * 8 + e, 0, 0]
    assert ops.pack_conv(torch.randn(96, 64, 1, 1), None, None, torch.bfloat16, 1, 0, 1).w_frag is None      # Cout % 256 != 0: tile kernels
    # DLA base layer: [7][64][8]  <-  w[o = l & 15][c = e & 3][ky][kx = 2 (l >> 4) + (e >> 2)], zero for c = 3, kx = 7, o >= Cout
    w7 = torch.randn(12, 3, 7, 7, generator=g)
    pc7 = ops.pack_image_conv(w7, None, torch.float16, 1, 3)
    fr7 = pc7.w_frag7.reshape(7, 64, 8).float()
    w7h = w7.half().float()
    for ky in range(7):
        for l in range(64):
            for e in range(8):
                o, c, kx = l & 15, e & 3, 2 * (l >> 4) + (e >> 2)
                want = w7h[o, c, ky, kx] if (o < 12 and c < 3 and kx < 7) else 0.0
                assert fr7[ky, l, e] == want
    assert ops.pack_image_conv(w7, None, torch.float32, 1, 3).w_frag7 is None             # fp32: generic path


def test_grouped_tile_order_is_a_bijection():
    """The tile remap of ConvArgs::group_m (csrc/conv_igemm.hip: group_m pixel tiles x all N tiles per run) restated in Python: every
    (tile_m, tile_n) is produced exactly once for tile counts that are and are not multiples of the group, and a run of
    group_m * tiles_n consecutive indices covers all N tiles of group_m pixel tiles."""
    def remap(tile, tiles_m, tiles_n, group_m):
        per = group_m * tiles_n
        g, rr = divmod(tile, per)
        gm = min(group_m, tiles_m - g * group_m)
        tile_n = rr // gm
        return g * group_m + rr - tile_n * gm, tile_n

    for tiles_m, tiles_n, gm in [(180, 8, 4), (7, 3, 4), (1, 5, 4), (9, 1, 4), (10, 2, 3)]:
        seen = [remap(t, tiles_m, tiles_n, gm) for t in range(tiles_m * tiles_n)]
        assert len(set(seen)) == tiles_m * tiles_n
        assert all(0 <= m < tiles_m and 0 <= n < tiles_n for m, n in seen)
        first = seen[:gm * tiles_n] if tiles_m >= gm else seen
        assert {n for _, n in first} == set(range(tiles_n)) and len({m for m, _ in first}) == min(gm, tiles_m)


def test_xcd_tile_walk_is_a_partition():
    """The XCD-aware walk of the persistent stem / layer-1 / small-channel kernels (csrc/conv_resident.hip xcd_tile_walk, csrc/stem_pool.hip) restated in
    Python: over all workgroups every tile is visited exactly once, an XCD's workgroups (b, b + 8, ...) cover ONE contiguous run of the tile list,
    and in every round they take consecutive tiles of it; a grid that is not a multiple of 8 keeps the plain walk."""
    def walk(b, grid, ntiles):
        if grid % 8:
            return list(range(b, ntiles, grid))
        xcd, q, r = b & 7, ntiles >> 3, ntiles & 7
        start = xcd * q + min(xcd, r)
        end = start + q + (1 if xcd < r else 0)
        return list(range(start + (b >> 3), end, grid >> 3))

    for grid, ntiles in [(256, 3840), (256, 3841), (240, 240), (256, 487), (8, 3), (64, 10), (250, 1000), (256, 255)]:
        per_wg = [walk(b, grid, ntiles) for b in range(grid)]
        seen = sorted(t for w in per_wg for t in w)
        assert seen == list(range(ntiles)), (grid, ntiles)
        if grid % 8 == 0:
            for xcd in range(8):
                mine = sorted(t for b in range(xcd, grid, 8) for t in per_wg[b])
                assert mine == list(range(mine[0], mine[0] + len(mine))) if mine else True          # one contiguous run per XCD
                rounds = max((len(per_wg[b]) for b in range(xcd, grid, 8)), default=0)
                for k in range(rounds):                                                              # round k: consecutive tiles
                    row = [per_wg[b][k] for b in range(xcd, grid, 8) if len(per_wg[b]) > k]
                    assert row == list(range(row[0], row[0] + len(row)))


def test_unpad_own_copies_equal_the_plain_slices():
    """`unpad(..., own=True)` (what the detectors' test_forward returns: views of private copies made before the host sync) against the plain
    per-sample slices: same values, dtypes, shapes, contiguous -- both heads; counts through `lib/graphed.read_counts` (CPU tensors: .tolist())."""
    from visualdet3d_amd.networks.heads.detection_3d_head import AnchorBasedDetection3DHead as H
    from visualdet3d_amd.networks.heads.km3d_head import KM3DHead as K
    from visualdet3d_amd.networks.lib.graphed import read_counts
    g = torch.Generator().manual_seed(0)
    B, M = 3, 8
    sc, bx = torch.rand(B, M, generator=g), torch.rand(B, M, 11, generator=g)
    lb = torch.randint(0, 3, (B, M), generator=g, dtype=torch.int32)
    cnt = torch.tensor([2, 0, 5], dtype=torch.int32)
    assert read_counts(cnt) == [2, 0, 5]
    for plain, own in ((H.unpad((sc, bx, lb, None, cnt)), H.unpad((sc, bx, lb, None, cnt), own=True)),
                       (K.unpad((sc, bx, lb, cnt)), K.unpad((sc, bx, lb, cnt), own=True))):
        assert len(plain) == len(own) == B
        for x, y in zip(plain, own):
            for p_, q_ in zip(x, y):
                assert torch.equal(p_, q_) and p_.dtype == q_.dtype and p_.shape == q_.shape and q_.is_contiguous()
        assert own[2][2].dtype == torch.int64
        own[2][0].zero_()                                   # the caller's copy is private
        assert float(sc[2, :5].abs().sum()) > 0
    with pytest.raises(RuntimeError):
        H.unpad((sc, bx, lb, None, torch.tensor([1, -1, 0], dtype=torch.int32)), own=True)


def test_host_feed_frames_are_the_resident_images_as_bytes():
    """bench.py --feed host: the uploaded frames are the resident workload's images de-normalised and rounded to bytes (so that the host-fed step also
    decodes detections); slot 1 = slot 0 in another batch order."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from visualdet3d_amd.utils import synthetic as syn
    L, R = syn.stereo_pair(2, 32, 64, seed=5)
    f0, f1 = bench.HostFeed.slot_frames(L, R)
    assert f0.shape == (4, 32, 64, 3) and f0.dtype == torch.uint8 and f1.shape == f0.shape
    assert torch.equal(f1[1], f0[0]) and torch.equal(f1[0], f0[1]) and torch.equal(f1[3], f0[2])
    mean, std = torch.tensor((0.485, 0.456, 0.406)), torch.tensor((0.229, 0.224, 0.225))
    back = (f0[:2].float() / 255 - mean) / std               # [2, H, W, 3]
    want = L.permute(0, 2, 3, 1)
    inside = ((want * std + mean) >= 0) & ((want * std + mean) <= 1)
    assert float(((back - want).abs() * inside).max()) <= 0.5 / 255 / float(std.min()) + 1e-6
    assert float(inside.float().mean()) > 0.9


def test_bench_gpus_flag_is_honoured(tmp_path):
    """`bench.py --gpus N`: under a launcher N must equal WORLD_SIZE; without one and N > 1 the script re-executes itself under
    torch.distributed.run with N ranks; it never prints an `n_gpus: 1` line for a `--gpus 8` command."""
    import os
    import subprocess
    import sys

    import pytest
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    assert bench.resolve_world(None, {}, ['bench.py']) == ('run', 1)
    assert bench.resolve_world(1, {}, ['bench.py']) == ('run', 1)
    assert bench.resolve_world(None, {'WORLD_SIZE': '4'}, ['bench.py']) == ('run', 4)
    assert bench.resolve_world(8, {'WORLD_SIZE': '8'}, ['bench.py']) == ('run', 8)
    with pytest.raises(ValueError):
        bench.resolve_world(8, {'WORLD_SIZE': '1'}, ['bench.py'])
    with pytest.raises(ValueError):
        bench.resolve_world(2, {'WORLD_SIZE': '4'}, ['bench.py'])
    how, cmd = bench.resolve_world(8, {}, ['bench.py', '--gpus', '8', '--steps', '3'], executable='python')
    assert how == 'exec' and cmd[:3] == ['python', '-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-5:] == ['bench.py', '--gpus', '8', '--steps', '3']
    # the script itself: a mismatch exits non-zero before anything touches the GPU, and nothing is printed on stdout
    env = dict(os.environ, WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '8'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == '' and 'disagree' in r.stderr


def test_bench_dcn_sampling_stats_and_workload_weights():
    """bench.py's DCN workload statistics (CPU tensors): zero offsets on a 4 x 5 map with a 3 x 3 / pad 1 kernel -- every sample sits on a pixel centre, so the corners
    inside the image are exactly the in-image taps' own corners -- and a known shift; and the realistic-workload state dict scales every conv_offset entry and sets the
    KM3D heat-map biases while the legacy one leaves the seeded values."""
    import os
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    H, W, K = 4, 5, 9
    off = torch.zeros(1, H, W, 2 * K)
    sq, n, inside, total = bench.dcn_sampling_stats(off, 'nhwc', (3, 3), (1, 1), (1, 1), (1, 1), (H, W))
    assert sq == 0.0 and n == H * W * K * 2 and total == 4 * H * W * K
    want = 0
    for y in range(H):
        for x in range(W):
            for ky in range(3):
                for kx in range(3):
                    y0, x0 = y - 1 + ky, x - 1 + kx
                    want += sum((0 <= yy <= H - 1) and (0 <= xx <= W - 1) for yy in (y0, y0 + 1) for xx in (x0, x0 + 1))
    assert inside == want
    # NCHW layout, every sample moved 10 px down: nothing of the lower taps stays inside
    off2 = torch.zeros(1, 2 * K, H, W)
    off2[:, 0::2] = 10.0
    sq2, n2, inside2, total2 = bench.dcn_sampling_stats(off2, 'nchw', (3, 3), (1, 1), (1, 1), (1, 1), (H, W))
    assert abs(sq2 - 100.0 * H * W * K) < 1e-6 and inside2 == 0 and total2 == total

    class M:
        def state_dict(self):
            return {'core.up.conv.conv_offset.weight': torch.zeros(27, 64, 3, 3), 'core.up.conv.conv_offset.bias': torch.zeros(27),
                    'bbox_head.head_layers.hm.2.bias': torch.zeros(3), 'bbox_head.head_layers.hm_hp.2.bias': torch.zeros(9),
                    'bbox_head.head_layers.hm.2.weight': torch.zeros(3, 256, 1, 1), 'core.x.weight': torch.zeros(8, 8, 3, 3)}
    c = dict(kind='km3d', offset_scale=0.25, hm_bias=-2.0, hm_hp_bias=-3.5, head_gain=0.5)
    new, old = bench.other_config_state_dict(c, M()), bench.other_config_state_dict(c, M(), legacy=True)
    assert torch.allclose(new['core.up.conv.conv_offset.weight'], 0.25 * old['core.up.conv.conv_offset.weight'])
    assert torch.allclose(new['core.up.conv.conv_offset.bias'], 0.25 * old['core.up.conv.conv_offset.bias'])
    assert float(new['bbox_head.head_layers.hm.2.bias'][0]) == -2.0 and float(new['bbox_head.head_layers.hm_hp.2.bias'][0]) == -3.5
    assert torch.allclose(new['bbox_head.head_layers.hm.2.weight'], 0.5 * old['bbox_head.head_layers.hm.2.weight'])
    assert torch.equal(new['core.x.weight'], old['core.x.weight'])
