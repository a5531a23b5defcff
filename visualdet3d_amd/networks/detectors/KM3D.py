"""``KM3D`` registered in ``DETECTOR_DICT`` with the reference's constructor / ``forward`` / ``test_forward`` signatures
(detectors/KM3D.py:16-88) on the HIP path.  ``MonoFlex`` (a different head) is out of scope."""
import torch
import torch.nn as nn

from ..heads.km3d_head import KM3DHead
from ..lib import fused
from ..lib.graphed import GraphedForward
from ..utils.registry import DETECTOR_DICT
from .KM3D_core import KM3DCore


@DETECTOR_DICT.register_module
class KM3D(GraphedForward, nn.Module):
    def __init__(self, network_cfg):
        super(KM3D, self).__init__()
        self.obj_types = network_cfg.obj_types
        self.build_head(network_cfg)
        self.build_core(network_cfg)
        self.network_cfg = network_cfg
        self.compute_dtype = None

    def build_core(self, network_cfg):
        self.core = KM3DCore(network_cfg.backbone)

    def build_head(self, network_cfg):
        self.bbox_head = KM3DHead(**(network_cfg.head))

    def training_forward(self, img_batch, annotations, meta):
        raise NotImplementedError('training is out of scope of the MI355X inference path (SURVEY.md 2)')

    def forward_device(self, img_batch, P2):
        if not img_batch.is_cuda:
            raise RuntimeError('KM3D runs on the MI355X HIP path only: move the model and inputs to cuda')
        dtype = self.compute_dtype or fused.default_compute_dtype()
        feat = self.core.forward_nhwc(img_batch, dtype)
        maps = self.bbox_head.forward_nhwc(feat)
        self._last_raw = maps
        return self.bbox_head.get_bboxes_batched(maps, P2, img_batch.shape[2:])

    @torch.no_grad()
    def test_forward_batched(self, img_batch, P2):
        if not img_batch.is_cuda:
            raise RuntimeError('KM3D runs on the MI355X HIP path only: move the model and inputs to cuda')
        # the calibration in kernel form (contiguous fp32 on the device) BEFORE the graph cache: the graph's static input is then what the
        # kernels read, and a float64 / host / strided P2 of a later frame reaches them through the per-call copy into it
        P2 = torch.as_tensor(P2).to(device=img_batch.device, dtype=torch.float32).contiguous()
        return self.bbox_head.unpad(self._graphed(img_batch, P2), own=True,      # hipGraph cache, lib/graphed.py
                                    retry=lambda b: self.bbox_head.decode_unbounded({k: v[b:b + 1] for k, v in self._last_raw.items()}, P2[b:b + 1],
                                                                                    img_batch.shape[2:]))

    @torch.no_grad()
    def test_forward(self, img_batch, P2):
        assert img_batch.shape[0] == 1  # the reference's contract (KM3D.py:72)
        return self.test_forward_batched(img_batch, P2)[0]

    def forward(self, inputs):
        if isinstance(inputs, list) and len(inputs) == 3:
            return self.training_forward(*inputs)
        img_batch, calib = inputs
        return self.test_forward(img_batch, calib)
