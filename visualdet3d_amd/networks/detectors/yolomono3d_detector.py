"""``Yolo3D`` / ``GroundAwareYolo3D`` registered in ``DETECTOR_DICT`` with the reference's constructor and
``forward`` / ``test_forward`` signatures (detectors/yolomono3d_detector.py:12-138) on the HIP path.

``Yolo3D``: base anchor head, reg tower opens with DCNv2 (heads/detection_3d_head.py:69-79).
``GroundAwareYolo3D``: reg tower opens with LookGround."""
import torch
import torch.nn as nn

from ... import hip_ops as ops
from ..heads.detection_3d_head import AnchorBasedDetection3DHead, _conv_pack
from ..lib import fused
from ..lib.blocks import AnchorFlatten
from ..lib.graphed import GraphedForward
from ..lib.look_ground import LookGround
from ..utils.registry import DETECTOR_DICT
from .yolomono3d_core import YoloMono3DCore


class GroundAwareHead(AnchorBasedDetection3DHead):
    def init_layers(self, num_features_in, num_anchors: int, num_cls_output: int, num_reg_output: int,
                    cls_feature_size: int = 1024, reg_feature_size: int = 1024, **kwargs):
        self.cls_feature_extraction = self._cls_tower(num_features_in, cls_feature_size, num_anchors, num_cls_output)
        self.reg_feature_extraction = nn.Sequential(
            LookGround(reg_feature_size),
            nn.Conv2d(num_features_in, reg_feature_size, 3, padding=1),
            nn.BatchNorm2d(reg_feature_size),
            nn.ReLU(),
            nn.Conv2d(reg_feature_size, reg_feature_size, kernel_size=3, padding=1),
            nn.BatchNorm2d(reg_feature_size),
            nn.ReLU(inplace=True),
            nn.Conv2d(reg_feature_size, num_anchors * num_reg_output, kernel_size=3, padding=1),
            AnchorFlatten(num_reg_output),
        )
        self.reg_feature_extraction[-2].weight.data.fill_(0)
        self.reg_feature_extraction[-2].bias.data.fill_(0)

    def _reg_forward_nhwc(self, feat, inputs=None):
        t, dt = self.reg_feature_extraction, feat.dtype
        x = t[0].forward_nhwc(feat, inputs['P2'])
        x = ops.conv2d(x, _conv_pack(self._cache, 'reg1', t[1], t[2], dt), relu=True)
        x = ops.conv2d(x, _conv_pack(self._cache, 'reg4', t[4], t[5], dt), relu=True)
        x = ops.conv2d(x, _conv_pack(self._cache, 'reg7', t[7], None, dt), relu=False, out_f32=True)
        return t[8].forward_nhwc(x)

    def forward(self, inputs):
        d = dict(inputs)
        d['features'] = fused.to_nhwc(inputs['features'])
        return self.forward_nhwc(d)


@DETECTOR_DICT.register_module
class Yolo3D(GraphedForward, nn.Module):
    def __init__(self, network_cfg):
        super(Yolo3D, self).__init__()
        self.obj_types = network_cfg.obj_types
        self.build_head(network_cfg)
        self.build_core(network_cfg)
        self.network_cfg = network_cfg
        self.compute_dtype = None

    def build_core(self, network_cfg):
        self.core = YoloMono3DCore(network_cfg.backbone)

    def build_head(self, network_cfg):
        self.bbox_head = AnchorBasedDetection3DHead(**(network_cfg.head))

    def training_forward(self, img_batch, annotations, P2):
        raise NotImplementedError('training is out of scope of the MI355X inference path (SURVEY.md 2)')

    def forward_device(self, img_batch, P2):
        if not img_batch.is_cuda:
            raise RuntimeError('Yolo3D runs on the MI355X HIP path only: move the model and inputs to cuda')
        dtype = self.compute_dtype or fused.default_compute_dtype()
        feat = self.core.forward_nhwc(img_batch, dtype)
        cls_preds, reg_preds = self.bbox_head.forward_nhwc(dict(features=feat, P2=P2, image=img_batch))
        self._last_raw = (cls_preds, reg_preds)
        return self.bbox_head.get_bboxes_batched(cls_preds, reg_preds, P2, img_batch.shape[2:])

    def _overflow_retry(self, b, P2, img_hw):
        """sample b of the last forward had more candidates than the captured capacity: its logits (the forward's static outputs, still intact) once
        more through the post-processing, eagerly, with a capacity that fits (bbox_head.get_bboxes_unbounded)"""
        cls_preds, reg_preds = self._last_raw
        return self.bbox_head.get_bboxes_unbounded(cls_preds[b:b + 1], reg_preds[b:b + 1], P2[b:b + 1], img_hw)

    @torch.no_grad()
    def test_forward_batched(self, img_batch, P2):
        if not img_batch.is_cuda:
            raise RuntimeError('Yolo3D runs on the MI355X HIP path only: move the model and inputs to cuda')
        # the calibration in kernel form (contiguous fp32 on the device) BEFORE the graph cache: the graph's static input is then what the
        # kernels read, and a float64 / host / strided P2 of a later frame reaches them through the per-call copy into it
        P2 = torch.as_tensor(P2).to(device=img_batch.device, dtype=torch.float32).contiguous()
        # through the hipGraph cache (lib/graphed.py); post_optimization runs inside get_bboxes_batched, i.e. inside the graph
        return self.bbox_head.unpad(self._graphed(img_batch, P2), own=True, retry=lambda b: self._overflow_retry(b, P2, img_batch.shape[2:]))

    @torch.no_grad()
    def test_forward(self, img_batch, P2):
        assert img_batch.shape[0] == 1  # the reference's contract (yolomono3d_detector.py:111)
        return self.test_forward_batched(img_batch, P2)[0]

    def forward(self, inputs):
        if isinstance(inputs, list) and len(inputs) == 3:
            return self.training_forward(*inputs)
        img_batch, calib = inputs
        return self.test_forward(img_batch, calib)


@DETECTOR_DICT.register_module
class GroundAwareYolo3D(Yolo3D):
    def build_head(self, network_cfg):
        self.bbox_head = GroundAwareHead(**(network_cfg.head))
