"""ORACLE tooling (build container): golden fixtures for the native ops, produced by the REFERENCE'S OWN device
functions compiled for the host (oracle/build_ref.sh -> oracle/_ref/*.so).

    bash oracle/build_ref.sh && python -m oracle.make_golden_native
"""
import os
import sys

import numpy as np

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)
from oracle import iou3d_ref  # noqa: E402

GOLDEN_DIR = os.path.join(_REPO, 'tests', 'golden')


def iou3d_boxes(n, seed):
    """BEV boxes (x1,y1,x2,y2,ry): random + the edge cases the op meets (identical, contained, disjoint, touching,
    axis-aligned, 90-degree rotations)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(0, 12, (n, 2))
    wh = rng.uniform(1, 6, (n, 2))
    ry = rng.uniform(-np.pi, np.pi, n)
    b = np.concatenate([c - wh / 2, c + wh / 2, ry[:, None]], axis=1).astype(np.float32)
    if n >= 8:
        b[1] = b[0]                                   # identical
        b[2, :4] = b[0, :4] + np.array([0.5, 0.5, -0.5, -0.5]); b[2, 4] = b[0, 4]   # contained, same angle
        b[3, :4] = b[0, :4] + 100.0                   # far away
        b[4] = [0, 0, 2, 2, 0]; b[5] = [2, 0, 4, 2, 0]  # touching, axis aligned
        b[6] = [0, 0, 4, 2, np.pi / 2]                # 90 degrees
        b[7] = [1, 1, 3, 3, 0.3]
    return b


DCN_CASES = {
    # name: B, C, H, W, O, k, stride, pad, dil, groups, dg, modulated
    'v2_3x3': (2, 16, 9, 13, 24, 3, 1, 1, 1, 1, 1, True),
    'v2_dg2_g2_s2': (1, 8, 12, 10, 128, 3, 2, 1, 1, 2, 2, True),
    'v1_3x3_d2': (2, 12, 8, 11, 20, 3, 1, 2, 2, 1, 1, False),
    'v2_1x1': (1, 20, 6, 7, 10, 1, 1, 0, 1, 1, 1, True),
}


def dcn_inputs(name):
    import torch
    B, C, H, W, O, k, st, pad, dil, groups, dg, mod = DCN_CASES[name]
    rng = np.random.default_rng(abs(hash(name)) % 1000 + 7) if False else np.random.default_rng(sum(map(ord, name)))
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // st + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // st + 1
    x = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32))
    off = torch.from_numpy((rng.standard_normal((B, dg * 2 * k * k, Ho, Wo)) * 2.5).astype(np.float32))  # far enough to leave the image
    off[:, :, 0, 0] = 0.0                               # exact-integer sample points too
    mask = torch.from_numpy(rng.uniform(0, 1, (B, dg * k * k, Ho, Wo)).astype(np.float32)) if mod else None
    w = torch.from_numpy((rng.standard_normal((O, C // groups, k, k)) * 0.2).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(O).astype(np.float32)) if mod else None
    return x, off, mask, w, bias, dict(stride=st, padding=pad, dilation=dil, groups=groups, deformable_groups=dg)


def dcn_golden():
    import torch
    from oracle import dcn_ref
    out = {}
    for name in DCN_CASES:
        x, off, mask, w, bias, kw = dcn_inputs(name)
        k = w.shape[2]
        cols = dcn_ref.native_im2col(x, off, mask, k, kw['stride'], kw['padding'], kw['dilation'], kw['deformable_groups'])
        # host side of the reference (deform_conv_cuda.cpp:531-569): per group W . columns (+ bias)
        B, C, K, Ho, Wo = cols.shape
        g = kw['groups']; O = w.shape[0]; Cg, Og = C // g, O // g
        y = torch.zeros(B, O, Ho, Wo)
        for gi in range(g):
            y[:, gi * Og:(gi + 1) * Og] = torch.matmul(w[gi * Og:(gi + 1) * Og].reshape(Og, Cg * K).double(),
                                                       cols[:, gi * Cg:(gi + 1) * Cg].reshape(B, Cg * K, Ho * Wo).double()).float().view(B, Og, Ho, Wo)
        if bias is not None:
            y = y + bias.view(1, -1, 1, 1)
        out[name + '_cols'] = cols.numpy()
        out[name + '_out'] = y.numpy()
        print('dcn golden', name, tuple(y.shape), float(y.abs().max()))
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'dcn_cases.npz'), **out)


def main():
    assert iou3d_ref.native_ref() is not None, 'run oracle/build_ref.sh first'
    dcn_golden()
    a, b = iou3d_boxes(24, 1), iou3d_boxes(17, 2)
    out = dict(boxes_a=a, boxes_b=b,
               overlap=iou3d_ref.native_pairwise('ref_box_overlap', a, b),
               iou_bev=iou3d_ref.native_pairwise('ref_iou_bev', a, b),
               iou_normal=iou3d_ref.native_pairwise('ref_iou_normal', a, b))
    # NMS: boxes sorted by a random score; keep lists from the mask+scan semantics driven by the NATIVE iou functions
    lib = iou3d_ref.native_ref()
    import ctypes
    nb = iou3d_boxes(150, 3)
    nb[:, :4] *= 0.6   # denser -> more suppression
    def nat(name):
        fn = getattr(lib, name)
        return lambda x, y: np.float32(fn(np.ascontiguousarray(x, dtype=np.float32).ctypes.data_as(ctypes.c_void_p),
                                          np.ascontiguousarray(y, dtype=np.float32).ctypes.data_as(ctypes.c_void_p)))
    out['nms_boxes'] = nb
    out['nms_keep_rot_03'] = iou3d_ref.nms(nb, 0.3, iou_fn=nat('ref_iou_bev'))
    out['nms_keep_norm_03'] = iou3d_ref.nms(nb, 0.3, iou_fn=nat('ref_iou_normal'))
    out['nms_keep_rot_01'] = iou3d_ref.nms(nb, 0.1, iou_fn=nat('ref_iou_bev'))
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'iou3d_cases.npz'), **out)
    print('iou3d golden: overlap max', out['overlap'].max(), 'keeps', len(out['nms_keep_rot_03']), len(out['nms_keep_norm_03']), len(out['nms_keep_rot_01']))


if __name__ == '__main__':
    main()
