"""Stereo cost-volume modules with the reference's names (lib/PSM_cost_volume.py:19-96) on HIP kernels.

``PSMCosineModule``: one kernel per pyramid level (LDS-staged right-feature window, all disparities at once) instead
of the reference's per-disparity Python loop.  ``CostVolume``: 1x1 down-sample as an implicit GEMM (L and R stacked
on the batch axis), concat-volume build, two fused Conv3d+BN3d+ReLU; the final reshape to ``[B, F*D, H, W]``
(channel = f*D + d) is folded into the second conv's store and lands directly in the caller's concat buffer."""
import torch
import torch.nn as nn

from ... import hip_ops as ops
from . import fused


class PSMCosineModule(nn.Module):
    def __init__(self, max_disp=192, downsample_scale=4, input_features=512):
        super(PSMCosineModule, self).__init__()
        self.max_disp = max_disp
        self.downsample_scale = downsample_scale
        self.depth_channel = int(self.max_disp / self.downsample_scale)

    def forward_nhwc(self, left, right, out=None):
        return ops.psm_cosine(left, right, self.depth_channel, out=out)

    def forward(self, left_features, right_features):
        dt = fused.default_compute_dtype()
        return fused.to_nchw(self.forward_nhwc(fused.to_nhwc(left_features, dt), fused.to_nhwc(right_features, dt)))


class CostVolume(nn.Module):
    def __init__(self, max_disp=192, downsample_scale=4, input_features=1024, PSM_features=64):
        super(CostVolume, self).__init__()
        self.max_disp = max_disp
        self.downsample_scale = downsample_scale
        self.depth_channel = int(self.max_disp / self.downsample_scale)
        self.down_sample = nn.Sequential(
            nn.Conv2d(input_features, PSM_features, 1),
            nn.BatchNorm2d(PSM_features),
            nn.ReLU(),
        )
        self.conv3d = nn.Sequential(
            nn.Conv3d(2 * PSM_features, PSM_features, 3, padding=1),
            nn.BatchNorm3d(PSM_features),
            nn.ReLU(),
            nn.Conv3d(PSM_features, PSM_features, 3, padding=1),
            nn.BatchNorm3d(PSM_features),
            nn.ReLU(),
        )
        self.output_channel = PSM_features * self.depth_channel
        self._cache = fused.PackCache()
        self.fuse_volume = True       # False: the three-launch path (A/B, per-stage parity taps)

    def forward_nhwc(self, feats_lr, batch, out=None):
        """feats_lr: NHWC [2B,H,W,C] with left images first (as the backbone produced them)."""
        dt = feats_lr.dtype
        conv, bn = self.down_sample[0], self.down_sample[1]
        pc = self._cache.get(('ds', dt), [conv.weight, conv.bias] + fused.bn_sources(bn),
                             lambda: ops.pack_conv(conv.weight, conv.bias, fused.bn_tuple(bn), dt, 1, 0, 1))
        small = ops.conv2d(feats_lr, pc, relu=True)  # [2B,H,W,F]
        c0, b0, c1, b1 = self.conv3d[0], self.conv3d[1], self.conv3d[3], self.conv3d[4]
        p0 = self._cache.get('c3d0', [c0.weight, c0.bias] + fused.bn_sources(b0), lambda: ops.pack_conv3d(c0.weight, c0.bias, fused.bn_tuple(b0)))
        p1 = self._cache.get('c3d1', [c1.weight, c1.bias] + fused.bn_sources(b1), lambda: ops.pack_conv3d(c1.weight, c1.bias, fused.bn_tuple(b1)))
        if self.fuse_volume and ops.cost_volume_fused_supported(small, self.depth_channel) and (p0.Cin, p1.Cin) == (16, 8):
            # bf16: concat volume + both Conv3d + BN3d + ReLU + reshape in ONE launch (the volume never reaches HBM)
            return ops.cost_volume_fused(small[:batch], small[batch:], p0, p1, self.depth_channel, out=out)
        vol = ops.costvol_build(small[:batch], small[batch:], self.depth_channel)
        mid = ops.conv3d_3x3x3(vol, p0, relu=True)
        B, D, H, W, _ = mid.shape
        if out is None:
            out = torch.empty((B, H, W, self.output_channel), dtype=dt, device=mid.device)
        ops.conv3d_3x3x3(mid, p1, relu=True, out_nhwc=out)
        return out

    def forward(self, left_features, right_features):
        dt = fused.default_compute_dtype()
        B = left_features.shape[0]
        x = fused.to_nhwc(torch.cat([left_features, right_features], dim=0), dt)
        return fused.to_nchw(self.forward_nhwc(x, B))
