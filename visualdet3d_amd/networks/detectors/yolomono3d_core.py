"""``YoloMono3DCore`` (detectors/yolomono3d_core.py:9-18): ResNet backbone, single feature map."""
import torch.nn as nn

from ..backbones import resnet
from ..lib import fused


class YoloMono3DCore(nn.Module):
    def __init__(self, backbone_arguments=dict()):
        super(YoloMono3DCore, self).__init__()
        self.backbone = resnet(**backbone_arguments)

    def forward_nhwc(self, image, dtype=None):
        return self.backbone.forward_nhwc(image, dtype)[0]

    def forward(self, x):
        return fused.to_nchw(self.forward_nhwc(x['image']))
