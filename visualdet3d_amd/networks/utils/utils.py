"""Small geometry helpers that sit at the edge of the hot path (networks/utils/utils.py:181-278).

``ClipBoxes`` is fused into the head post-processing kernel on the product path; the module is kept for API
compatibility.  ``BackProjection`` / ``BBox3dProjector`` run after the detector (pipelines/evaluators.py:112-131) on a
handful of boxes -- plain tensor math."""
import torch
import torch.nn as nn


class ClipBoxes(nn.Module):
    def forward(self, boxes, img):
        _, _, height, width = img.shape
        boxes[:, 0].clamp_(min=0)
        boxes[:, 1].clamp_(min=0)
        boxes[:, 2].clamp_(max=width)
        boxes[:, 3].clamp_(max=height)
        return boxes


class BackProjection(nn.Module):
    """[u, v, z, w, h, l, alpha] + P2[3,4] -> [x3d, y3d, z, w, h, l, alpha]."""

    def forward(self, bbox3d, p2):
        fx, fy, cx, cy, tx, ty = p2[0, 0], p2[1, 1], p2[0, 2], p2[1, 2], p2[0, 3], p2[1, 3]
        z = bbox3d[:, 2:3]
        x3d = (bbox3d[:, 0:1] * z - cx * z - tx) / fx
        y3d = (bbox3d[:, 1:2] * z - cy * z - ty) / fy
        return torch.cat([x3d, y3d, bbox3d[:, 2:]], dim=1)


def alpha2theta_3d(alpha, x, z, P2):
    """utils/utils.py (alpha2theta_3d): theta = alpha + atan2(x + tx/fx offset, z)."""
    offset = P2[0, 3] / P2[0, 0]
    return alpha + torch.atan2(x + offset, z)


class BBox3dProjector(nn.Module):
    """[x, y, z, w, h, l, alpha] -> (corners in camera frame [N,8,3], corners in image [N,8,3], theta [N])."""

    def __init__(self):
        super(BBox3dProjector, self).__init__()
        self.register_buffer('corner_matrix', torch.tensor(
            [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1], [-1, 1, -1]]).float())

    def forward(self, bbox_3d, tensor_p2):
        rel = 0.5 * self.corner_matrix * bbox_3d[:, 3:6].unsqueeze(1)
        thetas = alpha2theta_3d(bbox_3d[..., 6], bbox_3d[..., 0], bbox_3d[..., 2], tensor_p2)
        c, s = torch.cos(thetas).unsqueeze(1), torch.sin(thetas).unsqueeze(1)
        rx = rel[:, :, 2] * c + rel[:, :, 0] * s
        rz = -rel[:, :, 2] * s + rel[:, :, 0] * c
        rot = torch.stack([rx, rel[:, :, 1], rz], dim=-1)
        abs_corners = rot + bbox_3d[:, 0:3].unsqueeze(1)
        homo = torch.cat([abs_corners, abs_corners.new_ones([abs_corners.shape[0], 8, 1])], dim=-1).unsqueeze(3)
        cam = torch.matmul(tensor_p2, homo).squeeze(-1)
        return abs_corners, cam / (cam[:, :, 2:] + 1e-6), thetas
