"""ORACLE (test infrastructure only): restatement of the rotated IoU of ``visualDet3D/evaluator/kitti/rotate_iou.py:15-258``
(numba.cuda device functions) in numpy float32 / float64 with numba's typing: fp32 everywhere except the "/ 2.0" of the
triangle area, the area sum and the final ratio (fp64).  At most 8 intersection points are kept (the reference overflows its
16-float local array beyond that: undefined behaviour).  Pinned by tests/golden/rotate_iou_cases.npz: the reference's own
device functions executed as plain Python through the numba stand-in (oracle/make_golden_rotate_iou.py)."""
import math

import numpy as np

f32 = np.float32


def _corners(r):
    a_cos, a_sin = f32(math.cos(float(r[4]))), f32(math.sin(float(r[4])))
    xs = [-r[2] / f32(2), -r[2] / f32(2), r[2] / f32(2), r[2] / f32(2)]
    ys = [-r[3] / f32(2), r[3] / f32(2), r[3] / f32(2), -r[3] / f32(2)]
    c = np.zeros(8, f32)
    for i in range(4):
        c[2 * i] = a_cos * xs[i] + a_sin * ys[i] + r[0]
        c[2 * i + 1] = -a_sin * xs[i] + a_cos * ys[i] + r[1]
    return c


def _in_quad(px, py, c):
    ab0, ab1 = c[2] - c[0], c[3] - c[1]
    ad0, ad1 = c[6] - c[0], c[7] - c[1]
    ap0, ap1 = px - c[0], py - c[1]
    abab, abap = ab0 * ab0 + ab1 * ab1, ab0 * ap0 + ab1 * ap1
    adad, adap = ad0 * ad0 + ad1 * ad1, ad0 * ap0 + ad1 * ap1
    return abab >= abap and abap >= 0 and adad >= adap and adap >= 0


def _seg(p1, p2, i, j):
    A0, A1, B0, B1 = p1[2 * i], p1[2 * i + 1], p1[2 * ((i + 1) % 4)], p1[2 * ((i + 1) % 4) + 1]
    C0, C1, D0, D1 = p2[2 * j], p2[2 * j + 1], p2[2 * ((j + 1) % 4)], p2[2 * ((j + 1) % 4) + 1]
    BA0, BA1, DA0, CA0, DA1, CA1 = B0 - A0, B1 - A1, D0 - A0, C0 - A0, D1 - A1, C1 - A1
    acd = DA1 * CA0 > CA1 * DA0
    bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0)
    if acd != bcd:
        abc = CA1 * BA0 > BA1 * CA0
        abd = DA1 * BA0 > BA1 * DA0
        if abc != abd:
            DC0, DC1 = D0 - C0, D1 - C1
            ABBA, CDDC = A0 * B1 - B0 * A1, C0 * D1 - D0 * C1
            DH = BA1 * DC0 - BA0 * DC1
            return (ABBA * DC0 - BA0 * CDDC) / DH, (ABBA * DC1 - BA1 * CDDC) / DH
    return None


def inter_area(r1, r2):
    c1, c2 = _corners(r1), _corners(r2)
    pts = []
    for i in range(4):
        if _in_quad(c1[2 * i], c1[2 * i + 1], c2):
            pts.append([c1[2 * i], c1[2 * i + 1]])
        if _in_quad(c2[2 * i], c2[2 * i + 1], c1):
            pts.append([c2[2 * i], c2[2 * i + 1]])
    for i in range(4):
        for j in range(4):
            t = _seg(c1, c2, i, j)
            if t is not None:
                pts.append([t[0], t[1]])
    pts = pts[:8]
    n = len(pts)
    if n > 0:
        cx, cy = f32(0), f32(0)
        for p in pts:
            cx, cy = cx + p[0], cy + p[1]
        cx, cy = cx / f32(n), cy / f32(n)
        vs = []
        for p in pts:
            v0, v1 = p[0] - cx, p[1] - cy
            d = f32(math.sqrt(float(v0 * v0 + v1 * v1)))
            v0, v1 = v0 / d, v1 / d
            if v1 < 0:
                v0 = f32(-2) - v0
            vs.append(v0)
        for i in range(1, n):
            if vs[i - 1] > vs[i]:
                temp, tp = vs[i], pts[i]
                j = i
                while j > 0 and vs[j - 1] > temp:
                    vs[j], pts[j] = vs[j - 1], pts[j - 1]
                    j -= 1
                vs[j], pts[j] = temp, tp
    area = 0.0
    for i in range(n - 2):
        a, b, c = pts[0], pts[i + 1], pts[i + 2]
        area += abs(float((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0)
    return area


def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    boxes = np.asarray(boxes, dtype=f32)
    query_boxes = np.asarray(query_boxes, dtype=f32)
    out = np.zeros((len(boxes), len(query_boxes)), dtype=f32)
    with np.errstate(all='ignore'):
        for n, b in enumerate(boxes):
            for k, q in enumerate(query_boxes):
                a1, a2 = q[2] * q[3], b[2] * b[3]          # rbox1 = query, rbox2 = box (rotate_iou.py:290-291)
                ai = inter_area(q, b)
                if criterion == -1:
                    v = ai / (float(a1 + a2) - ai)
                elif criterion == 0:
                    v = ai / float(a1)
                elif criterion == 1:
                    v = ai / float(a2)
                else:
                    v = ai
                out[n, k] = v
    return out
