"""``KM3DCore`` (detectors/KM3D_core.py:10-58) on HIP kernels, both necks of the reference:

  DLA-34 backbone  -> DLASegUpsample (16 DCNv2 layers), 64 feature channels            (KM3D_core.py:24-33)
  ResNet backbone  -> three ConvTranspose2d(4x4, stride 2, pad 1) + BN + ReLU, 256 ch  (KM3D_core.py:34-47, config/KM3D_example)

A 4x4 / stride-2 / pad-1 transposed convolution is four interleaved 2x2 convolutions (one per output-pixel parity); here they run as
ONE 3x3 implicit-GEMM launch with 4 x Cout output channels (the taps a parity does not use are zero), BN + ReLU folded into its
epilogue, followed by the pixel shuffle [B,H,W,(py,px,C)] -> [B,2H,2W,C]."""
import torch
import torch.nn as nn

from ... import hip_ops as ops
from ..backbones import build_backbone
from ..backbones.dla import DLA
from ..backbones.dla_utils import DLASegUpsample
from ..lib import fused


def deconv4x4s2_as_conv3x3(weight):
    """ConvTranspose2d weight [Cin, Cout, 4, 4] (stride 2, padding 1) -> Conv2d weight [4*Cout, Cin, 3, 3] (stride 1, padding 1) whose
    output channel (2*py + px)*Cout + o is the transposed convolution's output at pixel (2i + py, 2j + px):
    out[2i+py] = sum_ky x[(2i + py + 1 - ky) / 2] w[ky]  =>  tap ty = (py + 1 - ky)/2 + 1, i.e. ky = py + 3 - 2*ty."""
    cin, cout = weight.shape[:2]
    w3 = weight.new_zeros((2, 2, cout, cin, 3, 3))
    for py in range(2):
        for ty in range(3):
            ky = py + 3 - 2 * ty
            if not 0 <= ky <= 3:
                continue
            for px in range(2):
                for tx in range(3):
                    kx = px + 3 - 2 * tx
                    if 0 <= kx <= 3:
                        w3[py, px, :, :, ty, tx] = weight[:, :, ky, kx].t()
    return w3.reshape(4 * cout, cin, 3, 3)


class KM3DCore(nn.Module):
    def __init__(self, backbone_arguments=dict()):
        super(KM3DCore, self).__init__()
        self.backbone = build_backbone(backbone_arguments)
        self._cache = fused.PackCache()
        if isinstance(self.backbone, DLA):
            self.deconv_layers = DLASegUpsample(input_channels=[16, 32, 64, 128, 256, 512], down_ratio=4, final_kernel=1,
                                                last_level=5, out_channel=64)
            return
        depth = backbone_arguments.get('depth', 50)
        # (the reference writes 2024 for depth > 34, KM3D_core.py:20 -- a ResNet-50's last stage has 2048 channels, so that branch
        #  cannot run there; the bottleneck ResNets get their real width here)
        output_features = 512 if depth <= 34 else 2048
        feature_size = 256
        layers = []
        for i in range(3):
            layers += [nn.ConvTranspose2d(output_features if i == 0 else feature_size, feature_size, (4, 4), stride=(2, 2),
                                          padding=(1, 1), bias=False),
                       nn.BatchNorm2d(feature_size), nn.ReLU(inplace=True)]
        self.deconv_layers = nn.Sequential(*layers)
        for m in self.deconv_layers.modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)

    def _deconv_bn_relu(self, x, i):
        deconv, bn = self.deconv_layers[3 * i], self.deconv_layers[3 * i + 1]
        dt = x.dtype

        def build():
            s, b, m, v, eps = fused.bn_tuple(bn)
            rep = lambda t: t.detach().float().repeat(4)
            return ops.pack_conv(deconv4x4s2_as_conv3x3(deconv.weight.detach().float()), None, (rep(s), rep(b), rep(m), rep(v), eps),
                                 dt, 1, 1, 1)

        pc = self._cache.get(('deconv%d' % i, dt), [deconv.weight] + fused.bn_sources(bn), build)
        y = ops.conv2d(x, pc, relu=True)                                   # [B,H,W,(py,px,C)]
        B, H, W, _ = y.shape
        C = deconv.out_channels
        return y.view(B, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, C)

    def forward_nhwc(self, image, dtype=None):
        if isinstance(self.backbone, DLA):
            # DLA-Up reads levels first_level .. 5 only: level0's output never has to exist (level0 + level1 run as one launch)
            return self.deconv_layers.forward_nhwc(self.backbone.forward_nhwc(image, dtype, first_needed=self.deconv_layers.first_level))
        feats = self.backbone.forward_nhwc(image, dtype)
        x = feats[-1]
        for i in range(3):
            x = self._deconv_bn_relu(x, i)
        return x

    def forward(self, x):
        return fused.to_nchw(self.forward_nhwc(x['image']))
