"""``KM3DCore`` (detectors/KM3D_core.py:10-58): DLA-34 backbone + DLASegUpsample (16 DCNv2 layers) on HIP kernels.
The reference's alternative ResNet + dense ConvTranspose2d neck is not in BASELINE's configs and is not provided."""
import torch.nn as nn

from ..backbones import build_backbone
from ..backbones.dla import DLA
from ..backbones.dla_utils import DLASegUpsample
from ..lib import fused


class KM3DCore(nn.Module):
    def __init__(self, backbone_arguments=dict()):
        super(KM3DCore, self).__init__()
        self.backbone = build_backbone(backbone_arguments)
        if not isinstance(self.backbone, DLA):
            raise NotImplementedError('KM3DCore on the MI355X path supports the DLA-34 backbone (BASELINE config 5)')
        self.deconv_layers = DLASegUpsample(input_channels=[16, 32, 64, 128, 256, 512], down_ratio=4, final_kernel=1,
                                            last_level=5, out_channel=64)

    def forward_nhwc(self, image, dtype=None):
        return self.deconv_layers.forward_nhwc(self.backbone.forward_nhwc(image, dtype))

    def forward(self, x):
        return fused.to_nchw(self.forward_nhwc(x['image']))
