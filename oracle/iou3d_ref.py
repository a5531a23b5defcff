"""ORACLE (test infrastructure only): CPU restatement of the reference's iou3d extension.

Follows ``visualDet3D/networks/lib/ops/iou3d/src/iou3d_kernel.cu``:
  * ``box_overlap``  :108-212  rotated-rectangle intersection area (edge x edge intersections :70-95, contained
                               corners :49-64 with MARGIN 1e-5, angular bubble sort :101-103,176-185, fan shoelace)
  * ``iou_bev``      :214-221  overlap / max(sa + sb - overlap, 1e-8)
  * ``iou_normal``   :295-303  axis-aligned IoU with the same EPS clamp
  * NMS              :250-292 (64x64 bitmask blocks) + host greedy scan ``src/iou3d.cpp:100-116``
  * 3D IoU assembly  ``lib/ops/iou3d/iou3d.py:8-21,37-69``
All arithmetic in numpy float32 scalars, in the reference's operation order.  Pinned against the reference's own
device functions compiled for the host (``oracle/build_ref.sh`` -> ``oracle/_ref/libiou3d_ref.so``,
``tests/test_oracle_native_ref.py``) and via the golden fixture ``tests/golden/iou3d_cases.npz`` generated from that
library (``oracle/make_golden_native.py``).  Pure-Python loops: small cases only.
"""
import ctypes
import math
import os

import numpy as np

f32 = np.float32
EPS = f32(1e-8)
MARGIN = f32(1e-5)


def _cross3(p1, p2, p0):
    return (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1])


def _intersection(p1, p0, q1, q0):
    """segment p0p1 x segment q0q1 (iou3d_kernel.cu:70-95); returns point or None."""
    if not (min(p0[0], p1[0]) <= max(q0[0], q1[0]) and min(q0[0], q1[0]) <= max(p0[0], p1[0]) and
            min(p0[1], p1[1]) <= max(q0[1], q1[1]) and min(q0[1], q1[1]) <= max(p0[1], p1[1])):
        return None
    s1 = _cross3(q0, p1, p0)
    s2 = _cross3(p1, q1, p0)
    s3 = _cross3(p0, q1, q0)
    s4 = _cross3(q1, p1, q0)
    if not (s1 * s2 > 0 and s3 * s4 > 0):
        return None
    s5 = _cross3(q1, p1, p0)
    if abs(s5 - s1) > EPS:
        return (f32((s5 * q0[0] - s1 * q1[0]) / (s5 - s1)), f32((s5 * q0[1] - s1 * q1[1]) / (s5 - s1)))
    a0, b0, c0 = p0[1] - p1[1], p1[0] - p0[0], p0[0] * p1[1] - p1[0] * p0[1]
    a1, b1, c1 = q0[1] - q1[1], q1[0] - q0[0], q0[0] * q1[1] - q1[0] * q0[1]
    D = a0 * b1 - a1 * b0
    return (f32((b0 * c1 - b1 * c0) / D), f32((a1 * c0 - a0 * c1) / D))


def _rotate(center, c, s, p):
    return (f32((p[0] - center[0]) * c + (p[1] - center[1]) * s + center[0]),
            f32(-(p[0] - center[0]) * s + (p[1] - center[1]) * c + center[1]))


def _in_box(box, p):
    cx, cy = (box[0] + box[2]) / f32(2), (box[1] + box[3]) / f32(2)
    c, s = f32(np.cos(f32(-box[4]))), f32(np.sin(f32(-box[4])))
    rx = (p[0] - cx) * c + (p[1] - cy) * s + cx
    ry = -(p[0] - cx) * s + (p[1] - cy) * c + cy
    return rx > box[0] - MARGIN and rx < box[2] + MARGIN and ry > box[1] - MARGIN and ry < box[3] + MARGIN


def box_overlap(a, b):
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=f32)
    ca = ((a[0] + a[2]) / f32(2), (a[1] + a[3]) / f32(2))
    cb = ((b[0] + b[2]) / f32(2), (b[1] + b[3]) / f32(2))
    A = [(a[0], a[1]), (a[2], a[1]), (a[2], a[3]), (a[0], a[3])]
    B = [(b[0], b[1]), (b[2], b[1]), (b[2], b[3]), (b[0], b[3])]
    cosa, sina = f32(np.cos(a[4])), f32(np.sin(a[4]))
    cosb, sinb = f32(np.cos(b[4])), f32(np.sin(b[4]))
    A = [_rotate(ca, cosa, sina, p) for p in A]
    B = [_rotate(cb, cosb, sinb, p) for p in B]
    A.append(A[0])
    B.append(B[0])
    pts = []
    for i in range(4):
        for j in range(4):
            p = _intersection(A[i + 1], A[i], B[j + 1], B[j])
            if p is not None:
                pts.append(p)
    for k in range(4):
        if _in_box(a, B[k]):
            pts.append(B[k])
        if _in_box(b, A[k]):
            pts.append(A[k])
    cnt = len(pts)
    if cnt == 0:
        return f32(0.0)  # the reference divides 0/0 here and the empty loops leave area = 0
    sx, sy = f32(0), f32(0)
    for p in pts:
        sx, sy = f32(sx + p[0]), f32(sy + p[1])
    center = (f32(sx / f32(cnt)), f32(sy / f32(cnt)))
    ang = lambda p: f32(np.arctan2(f32(p[1] - center[1]), f32(p[0] - center[0])))
    for j in range(cnt - 1):
        for i in range(cnt - j - 1):
            if ang(pts[i]) > ang(pts[i + 1]):
                pts[i], pts[i + 1] = pts[i + 1], pts[i]
    area = f32(0)
    for k in range(cnt - 1):
        u = (f32(pts[k][0] - pts[0][0]), f32(pts[k][1] - pts[0][1]))
        v = (f32(pts[k + 1][0] - pts[0][0]), f32(pts[k + 1][1] - pts[0][1]))
        area = f32(area + (u[0] * v[1] - u[1] * v[0]))
    return f32(abs(area) / f32(2.0))


def iou_bev(a, b):
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=f32)
    sa = (a[2] - a[0]) * (a[3] - a[1])
    sb = (b[2] - b[0]) * (b[3] - b[1])
    ov = box_overlap(a, b)
    return f32(ov / max(f32(sa + sb - ov), EPS))


def iou_normal(a, b):
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=f32)
    w = max(f32(min(a[2], b[2]) - max(a[0], b[0])), f32(0))
    h = max(f32(min(a[3], b[3]) - max(a[1], b[1])), f32(0))
    inter = f32(w * h)
    sa = (a[2] - a[0]) * (a[3] - a[1])
    sb = (b[2] - b[0]) * (b[3] - b[1])
    return f32(inter / max(f32(sa + sb - inter), EPS))


def pairwise(fn, boxes_a, boxes_b):
    out = np.zeros((len(boxes_a), len(boxes_b)), dtype=f32)
    for i, a in enumerate(boxes_a):
        for j, b in enumerate(boxes_b):
            out[i, j] = fn(a, b)
    return out


def nms(boxes, thr, normal=False, iou_fn=None):
    """Greedy NMS over boxes already sorted by score (mask semantics of :250-292 + scan iou3d.cpp:100-116):
    box i (kept) suppresses every j > i with iou(i, j) > thr.  Returns kept indices (int64)."""
    fn = iou_fn or (iou_normal if normal else iou_bev)
    n = len(boxes)
    removed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        for j in range(i + 1, n):
            if not removed[j] and fn(boxes[i], boxes[j]) > f32(thr):
                removed[j] = True
    return np.asarray(keep, dtype=np.int64)


def boxes3d_to_bev(boxes3d):
    """lib/ops/iou3d/iou3d.py:8-21: (N,7) [x,y,z,h,w,l,ry] -> (N,5) [x1,y1,x2,y2,ry]."""
    b = np.asarray(boxes3d, dtype=f32)
    out = np.zeros((b.shape[0], 5), dtype=f32)
    out[:, 0] = b[:, 0] - b[:, 5] / f32(2)
    out[:, 1] = b[:, 2] - b[:, 4] / f32(2)
    out[:, 2] = b[:, 0] + b[:, 5] / f32(2)
    out[:, 3] = b[:, 2] + b[:, 4] / f32(2)
    out[:, 4] = b[:, 6]
    return out


def boxes_iou3d(boxes_a, boxes_b, overlap_fn=None):
    """lib/ops/iou3d/iou3d.py:37-69."""
    a = np.asarray(boxes_a, dtype=f32)
    b = np.asarray(boxes_b, dtype=f32)
    ov = overlap_fn(boxes3d_to_bev(a), boxes3d_to_bev(b)) if overlap_fn else pairwise(box_overlap, boxes3d_to_bev(a), boxes3d_to_bev(b))
    hmin = np.maximum((a[:, 1] - a[:, 3])[:, None], (b[:, 1] - b[:, 3])[None, :])
    hmax = np.minimum(a[:, 1][:, None], b[:, 1][None, :])
    oh = np.clip(hmax - hmin, 0, None).astype(f32)
    o3 = (ov * oh).astype(f32)
    va = (a[:, 3] * a[:, 4] * a[:, 5])[:, None]
    vb = (b[:, 3] * b[:, 4] * b[:, 5])[None, :]
    return (o3 / np.clip(va + vb - o3, f32(1e-7), None)).astype(f32)


# ---- the reference's own device functions, compiled for the host (optional strengthening) -----------------------
_REF_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libiou3d_ref.so')


def native_ref():
    """ctypes handle of oracle/_ref/libiou3d_ref.so (built by oracle/build_ref.sh) or None."""
    if not os.path.exists(_REF_SO):
        return None
    lib = ctypes.CDLL(_REF_SO)
    for n in ('ref_box_overlap', 'ref_iou_bev', 'ref_iou_normal'):
        getattr(lib, n).restype = ctypes.c_float
        getattr(lib, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return lib


def native_pairwise(name, boxes_a, boxes_b):
    lib = native_ref()
    fn = getattr(lib, name)
    a = np.ascontiguousarray(boxes_a, dtype=f32)
    b = np.ascontiguousarray(boxes_b, dtype=f32)
    out = np.zeros((len(a), len(b)), dtype=f32)
    for i in range(len(a)):
        for j in range(len(b)):
            out[i, j] = fn(a[i].ctypes.data_as(ctypes.c_void_p), b[j].ctypes.data_as(ctypes.c_void_p))
    return out
