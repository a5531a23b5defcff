from .iou3d import (boxes3d_to_bev_torch, boxes_iou3d_gpu, boxes_iou_bev, iou3d_hip, nms_gpu, nms_normal_gpu)  # noqa: F401
