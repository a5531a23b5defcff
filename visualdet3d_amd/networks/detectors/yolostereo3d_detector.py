"""``Stereo3D`` (YOLOStereo3D) registered in ``DETECTOR_DICT`` with the reference's constructor, ``forward`` dispatch
and ``test_forward`` signature (detectors/yolostereo3d_detector.py:16-103), running on the HIP path.

``test_forward`` keeps the reference's batch-1 contract.  ``test_forward_batched`` is the new entry for B >= 1:
per-sample results are identical to running each pair through ``test_forward`` alone (eval-mode BN, per-sample
post-processing), with a single device->host sync at the very end."""
import torch
import torch.nn as nn

from ..heads.detection_3d_head import AnchorBasedDetection3DHead, StereoHead
from ..lib import fused
from ..lib.graphed import GraphedForward
from ..utils.registry import DETECTOR_DICT
from .yolostereo3d_core import YoloStereo3DCore


@DETECTOR_DICT.register_module
class Stereo3D(GraphedForward, nn.Module):
    def __init__(self, network_cfg):
        super(Stereo3D, self).__init__()
        self.obj_types = network_cfg.obj_types
        self.build_head(network_cfg)
        self.build_core(network_cfg)
        self.network_cfg = network_cfg
        self.compute_dtype = None  # None -> fused.default_compute_dtype() (bf16); torch.float32 = validation mode

    def build_core(self, network_cfg):
        self.core = YoloStereo3DCore(network_cfg.backbone)

    def build_head(self, network_cfg):
        self.bbox_head = StereoHead(**(network_cfg.head))

    def train_forward(self, left_images, right_images, annotations, P2, P3, disparity=None):
        raise NotImplementedError('training is out of scope of the MI355X inference path (SURVEY.md 2)')

    # ---- device pipeline (no host sync) ---------------------------------------------------------------------
    def forward_device(self, left_images, right_images, P2):
        """Everything up to the padded detection tensors, on the current stream, without synchronising:
        returns (scores [B,K], boxes [B,K,11], labels [B,K] i32, anchor_idx [B,K] i32, count [B] i32)."""
        if not left_images.is_cuda:
            raise RuntimeError('Stereo3D runs on the MI355X HIP path only: move the model and inputs to cuda')
        dtype = self.compute_dtype or fused.default_compute_dtype()
        feat = self.core.forward_nhwc(left_images, right_images, dtype)
        cls_preds, reg_preds = self.bbox_head.forward_nhwc(dict(features=feat, P2=P2, image=left_images))
        self._last_raw = (cls_preds, reg_preds)
        return self.bbox_head.get_bboxes_batched(cls_preds, reg_preds, P2, left_images.shape[2:])

    def _overflow_retry(self, b, P2, img_hw):
        """sample b of the last forward had more candidates than the captured capacity: its logits (the forward's static outputs, still intact) once
        more through the post-processing, eagerly, with a capacity that fits (bbox_head.get_bboxes_unbounded)"""
        cls_preds, reg_preds = self._last_raw
        return self.bbox_head.get_bboxes_unbounded(cls_preds[b:b + 1], reg_preds[b:b + 1], P2[b:b + 1], img_hw)

    @torch.no_grad()
    def test_forward_batched(self, left_images, right_images, P2, P3=None):
        """B >= 1.  Returns a list of per-sample ``(scores[N], bboxes[N,11], cls_indexes[N] int64)``.  Runs through the hipGraph
        cache (lib/graphed.py): the first call for a shape captures ``forward_device``, later calls replay it."""
        if not left_images.is_cuda:
            raise RuntimeError('Stereo3D runs on the MI355X HIP path only: move the model and inputs to cuda')
        # the calibration in kernel form (contiguous fp32 on the device) BEFORE the graph cache: the graph's static input is then what the
        # kernels read, and a float64 / host / strided P2 of a later frame reaches them through the per-call copy into it
        P2 = torch.as_tensor(P2).to(device=left_images.device, dtype=torch.float32).contiguous()
        return self.bbox_head.unpad(self._graphed(left_images, right_images, P2), own=True,
                                    retry=lambda b: self._overflow_retry(b, P2, left_images.shape[2:]))

    @torch.no_grad()
    def test_forward(self, left_images, right_images, P2, P3):
        assert left_images.shape[0] == 1  # the reference's contract (yolostereo3d_detector.py:78)
        return self.test_forward_batched(left_images, right_images, P2, P3)[0]

    def forward(self, inputs):
        if isinstance(inputs, list) and len(inputs) >= 5:
            return self.train_forward(*inputs)
        return self.test_forward(*inputs)


class Stereo3DBaseHead(Stereo3D):
    """BASELINE config 3 ("YOLOStereo3D ResNet-50 + DCNv2 head"): the stereo core with the BASE anchor head, whose reg tower
    starts with a ModulatedDeformConvPack (heads/detection_3d_head.py:69-79).  Not a shipped / registered model of the
    reference (``StereoHead`` has no DCN); it is ``Stereo3D`` with ``build_head`` overridden
    (yolostereo3d_detector.py:35-38) and nothing else, so it is not added to ``DETECTOR_DICT`` either."""

    def build_head(self, network_cfg):
        self.bbox_head = AnchorBasedDetection3DHead(**(network_cfg.head))
