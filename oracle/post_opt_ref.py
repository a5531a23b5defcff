"""ORACLE (test infrastructure only): fp64 restatement of the hill-climbing yaw post-optimisation.

Follows ``visualDet3D/networks/lib/fast_utils/hill_climbing.py`` (post_opt :7-23, post_optimization :25-51, hill_climb
:53-81, test_projection :84-122 -- including its hard-coded 1280 x 288 clamp), ``fast_utils/bbox3d.py:19-82`` (project_3d),
``fast_utils/bbox2d.py:39-66`` (iou_2d), ``utils/utils.py:30-45`` (alpha <-> rotation) and the caller
``heads/detection_3d_head.py:294-308`` (_post_process).  Pinned by ``tests/golden/post_opt_cases.npz``: outputs of the
reference's own functions imported through oracle/ref_shim.py (numba stub = plain Python), incl. the literal vector of
hill_climbing.py:125-141."""
import math

import numpy as np


def project_box(p2, x3d, y3d, z3d, w3d, h3d, l3d, ry):
    """8 projected corners (u, v) of the box (bbox3d.py:19-82)."""
    c, s = math.cos(ry), math.sin(ry)
    xs = [0.0, l3d, l3d, l3d, l3d, 0.0, 0.0, 0.0]
    ys = [0.0, 0.0, h3d, h3d, 0.0, 0.0, h3d, h3d]
    zs = [0.0, 0.0, 0.0, w3d, w3d, w3d, w3d, 0.0]
    out = []
    for i in range(8):
        x, y, z = xs[i] - l3d / 2, ys[i] - h3d / 2, zs[i] - w3d / 2
        X = c * x + s * z + x3d
        Y = y + y3d
        Z = -s * x + c * z + z3d
        u = p2[0, 0] * X + p2[0, 1] * Y + p2[0, 2] * Z + p2[0, 3]
        v = p2[1, 0] * X + p2[1, 1] * Y + p2[1, 2] * Z + p2[1, 3]
        wq = p2[2, 0] * X + p2[2, 1] * Y + p2[2, 2] * Z + p2[2, 3]
        out.append((u / wq, v / wq))
    return out


def test_projection(p2, p2_inv, box, cx, cy, z, w3d, h3d, l3d, ry):
    coord = p2_inv.dot(np.array([cx * z, cy * z, z, 1.0]))
    verts = project_box(p2, coord[0], coord[1], coord[2], w3d, h3d, l3d, ry)
    us, vs = [v[0] for v in verts], [v[1] for v in verts]
    xn, yn = max(0.0, min(us)), max(0.0, min(vs))
    x2n, y2n = min(max(us), 1280.0), min(max(vs), 288.0)      # hard-coded in the reference (hill_climbing.py:111-113)
    x1, x2 = max(box[0], xn), min(box[2], x2n)
    y1, y2 = max(box[1], yn), min(box[3], y2n)
    dx, dy = x2 - x1, y2 - y1
    if dx <= 0 or dy <= 0:
        return 0.0
    a0 = (box[2] - box[0]) * (box[3] - box[1])
    a1 = (x2n - xn) * (y2n - yn)
    return dx * dy / (a0 + a1 - dx * dy)


def hill_climb(p2, p2_inv, box, cx, cy, z, w, h, l, ry, step=0.4, r_lim=0.01):
    best = test_projection(p2, p2_inv, box, cx, cy, z, w, h, l, ry)
    while step > r_lim:
        neg = test_projection(p2, p2_inv, box, cx, cy, z, w, h, l, ry - step)
        pos = test_projection(p2, p2_inv, box, cx, cy, z, w, h, l, ry + step)
        if (pos - best) <= 0.0 and (neg - best) <= 0.0:
            step *= 0.5
        elif (pos - best) > 0.0 and pos > neg:
            ry += step
            best = pos
        elif (neg - best) > 0.0:
            ry -= step
            best = neg
        else:
            step *= 0.5
    while ry > 3.14:
        ry -= 3.14 * 2
    while ry < -3.14:
        ry += math.pi * 2
    return ry, best


def post_opt(box2d, state, P2, cx, cy):
    """box2d [4], state [x3d, y3d, z, w, h, l, alpha], P2 [3,4] -> [cx, cy, z, w, h, l, alpha'] (float64)."""
    P2 = np.asarray(P2, dtype=np.float64)
    p2 = np.eye(4)
    p2[0:3] = P2
    p2_inv = np.linalg.inv(p2)
    box = [float(v) for v in box2d]
    z, w, h, l, alpha = [float(v) for v in state[2:7]]
    cx, cy = float(cx), float(cy)
    theta = alpha + math.atan2(cx - P2[0, 2], P2[0, 0])
    if theta > math.pi:
        theta -= 2 * math.pi
    if theta <= -math.pi:
        theta += 2 * math.pi
    theta, _ = hill_climb(p2, p2_inv, box, cx, cy, z, w, h, l, theta)
    alpha = theta - math.atan2(cx - P2[0, 2], P2[0, 0])
    if alpha > math.pi:
        alpha -= 2 * math.pi
    if alpha <= -math.pi:
        alpha += 2 * math.pi
    return np.array([cx, cy, z, w, h, l, alpha])


def post_process(scores, bboxes, labels, P2):
    """heads/detection_3d_head.py:294-308 on numpy arrays: boxes [N,11] -> boxes with refined alpha for label-0 boxes
    whose depth is > 3 m."""
    b = np.array(bboxes, dtype=np.float32, copy=True)
    P2 = np.asarray(P2, dtype=np.float64)
    for i in range(len(b)):
        if b[i, 6] > 3 and int(labels[i]) == 0:
            z = float(b[i, 6])
            x3d = (float(b[i, 4]) * z - P2[0, 2] * z - P2[0, 3]) / P2[0, 0]
            y3d = (float(b[i, 5]) * z - P2[1, 2] * z - P2[1, 3]) / P2[1, 1]
            st = np.array([x3d, y3d, z, b[i, 7], b[i, 8], b[i, 9], b[i, 10]], dtype=np.float64)
            b[i, 4:] = post_opt(b[i, 0:4], st, P2, float(b[i, 4]), float(b[i, 5])).astype(np.float32)
    return b
