"""YOLOStereo3D core (detectors/yolostereo3d_core.py:14-126): siamese ResNet -> three cost volumes -> ghost
pyramid, with the reference's module / parameter names, executed on HIP kernels.

Every ``torch.cat`` of the reference is fused away: each producer writes into its channel slice of the NHWC
concat buffer of its consumer:

    s4  buf72  = [ PSV_0 (24) | primary(24) | cheap(24) ]                       (ResGhost 24 -> 72)
    s8  buf288 = [ BasicBlock72 out (72) | PSV_1 (24) | primary(96) | cheap(96) ] (cat + ResGhost 96 -> 288)
    s16 buf1152= [ BasicBlock288 out (288) | CostVolume (96) | primary(384) | cheap(384) ]
    s16 feat   = [ left layer3 (256) | BasicBlock1152 out (1152) ]              (final features, 1408 ch)

The left/right images are never concatenated on the channel axis and re-split (detector :80 + core :113-116, two
full-image copies in the reference): they are stacked once on the batch axis, which is what the backbone consumes.
"""
import torch
import torch.nn as nn

from ... import hip_ops as ops
from ..backbones import resnet
from ..backbones.resnet import BasicBlock
from ..lib import fused
from ..lib.ghost_module import ResGhostModule
from ..lib.PSM_cost_volume import CostVolume, PSMCosineModule


class CostVolumePyramid(nn.Module):
    def __init__(self, depth_channel_4, depth_channel_8, depth_channel_16):
        super(CostVolumePyramid, self).__init__()
        self.depth_channel_4 = depth_channel_4    # 24
        self.depth_channel_8 = depth_channel_8    # 24
        self.depth_channel_16 = depth_channel_16  # 96
        c = depth_channel_4
        self.four_to_eight = nn.Sequential(ResGhostModule(c, 3 * c, 3, ratio=3), nn.AvgPool2d(2), BasicBlock(3 * c, 3 * c))
        c = 3 * c + depth_channel_8
        self.eight_to_sixteen = nn.Sequential(ResGhostModule(c, 3 * c, 3, ratio=3), nn.AvgPool2d(2), BasicBlock(3 * c, 3 * c))
        c = 3 * c + depth_channel_16
        self.depth_reason = nn.Sequential(ResGhostModule(c, 3 * c, kernel_size=3, ratio=3), BasicBlock(3 * c, 3 * c))
        self.output_channel_num = 3 * c
        oc = self.output_channel_num
        # training-only disparity branch; kept so checkpoints load with identical keys (never executed in eval,
        # yolostereo3d_core.py:69-71)
        self.depth_output = nn.Sequential(
            nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True),
            nn.Conv2d(oc, int(oc / 2), 3, padding=1),
            nn.BatchNorm2d(int(oc / 2)),
            nn.ReLU(),
            nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True),
            nn.Conv2d(int(oc / 2), int(oc / 4), 3, padding=1),
            nn.BatchNorm2d(int(oc / 4)),
            nn.ReLU(),
            nn.Conv2d(int(oc / 4), 96, 1),
        )


class StereoMerging(nn.Module):
    def __init__(self, base_features):
        super(StereoMerging, self).__init__()
        self.cost_volume_0 = PSMCosineModule(downsample_scale=4, max_disp=96, input_features=base_features)
        self.cost_volume_1 = PSMCosineModule(downsample_scale=8, max_disp=192, input_features=base_features * 2)
        self.cost_volume_2 = CostVolume(downsample_scale=16, max_disp=192, input_features=base_features * 4, PSM_features=8)
        self.depth_reasoning = CostVolumePyramid(self.cost_volume_0.depth_channel, self.cost_volume_1.depth_channel,
                                                 self.cost_volume_2.output_channel)
        self.final_channel = self.depth_reasoning.output_channel_num + base_features * 4

    # The neck in four parts so that each can start as soon as the backbone stage it depends on is done
    # (YoloStereo3DCore.forward_nhwc runs parts s4 / s8 on a side stream under backbone layer2 / layer3).
    def alloc(self, f4, B):
        dr = self.depth_reasoning
        dt, dev = f4.dtype, f4.device
        d4, d8, d16 = dr.depth_channel_4, dr.depth_channel_8, dr.depth_channel_16
        _, H4, W4, _ = f4.shape
        H8, W8 = (H4 - 1) // 2 + 1, (W4 - 1) // 2 + 1          # 3x3 / stride 2 / pad 1 stages
        H16, W16 = (H8 - 1) // 2 + 1, (W8 - 1) // 2 + 1
        c72, c288, c1152 = 3 * d4, 3 * (3 * d4 + d8), dr.output_channel_num
        return dict(B=B, d4=d4, d8=d8, d16=d16, c72=c72, c288=c288, c1152=c1152,
                    buf72=torch.empty((B, H4, W4, c72), dtype=dt, device=dev),
                    buf288=torch.empty((B, H8, W8, c288), dtype=dt, device=dev),
                    buf1152=torch.empty((B, H16, W16, c1152), dtype=dt, device=dev), hw16=(H16, W16))

    def part_s4(self, f4, st):
        dr, B, d4, c72 = self.depth_reasoning, st['B'], st['d4'], st['c72']
        self.cost_volume_0.forward_nhwc(f4[:B], f4[B:], out=st['buf72'][..., :d4])
        x = dr.four_to_eight[0].forward_nhwc(st['buf72'][..., :d4], out=st['buf72'])
        x = ops.avgpool2x2(x)
        dr.four_to_eight[2].forward_nhwc(x, out=st['buf288'][..., :c72])

    def part_s8(self, f8, st):
        dr, B, d8, c72, c288 = self.depth_reasoning, st['B'], st['d8'], st['c72'], st['c288']
        self.cost_volume_1.forward_nhwc(f8[:B], f8[B:], out=st['buf288'][..., c72:c72 + d8])
        x = dr.eight_to_sixteen[0].forward_nhwc(st['buf288'][..., :c72 + d8], out=st['buf288'])
        x = ops.avgpool2x2(x)
        dr.eight_to_sixteen[2].forward_nhwc(x, out=st['buf1152'][..., :c288])

    def part_s16(self, f16, st):
        B, d16, c288 = st['B'], st['d16'], st['c288']
        _, H16, W16, C16 = f16.shape
        assert (H16, W16) == st['hw16']
        st['feat'] = torch.empty((B, H16, W16, C16 + st['c1152']), dtype=f16.dtype, device=f16.device)
        self.cost_volume_2.forward_nhwc(f16, B, out=st['buf1152'][..., c288:c288 + d16])
        ops.copy_channels(f16[:B], st['feat'][..., :C16])

    def part_merge(self, st):
        dr, d16, c288 = self.depth_reasoning, st['d16'], st['c288']
        feat = st['feat']
        C16 = feat.shape[3] - st['c1152']
        x = dr.depth_reason[0].forward_nhwc(st['buf1152'][..., :c288 + d16], out=st['buf1152'])
        dr.depth_reason[1].forward_nhwc(x, out=feat[..., C16:])
        return feat

    def forward_nhwc(self, feats, batch):
        """feats: [s4, s8, s16] NHWC tensors of the stacked [left; right] batch.  Returns features [B,H16,W16,C]."""
        f4, f8, f16 = feats
        st = self.alloc(f4, batch)
        self.part_s4(f4, st)
        self.part_s8(f8, st)
        self.part_s16(f16, st)
        return self.part_merge(st)


class YoloStereo3DCore(nn.Module):
    """Left and right images go through the backbone as one batch (yolostereo3d_core.py:96-126)."""

    def __init__(self, backbone_arguments):
        super(YoloStereo3DCore, self).__init__()
        self.backbone = resnet(**backbone_arguments)
        base_features = 256 if backbone_arguments['depth'] > 34 else 64
        self.neck = StereoMerging(base_features)
        self.overlap_neck = True     # False: everything on one stream (per-kernel profiling)
        self._side_streams = {}

    def forward_nhwc(self, left_images, right_images, dtype=None):
        """The s4 / s8 parts of the neck (PSM cost volumes, ghost pyramid: ~15 small launches) only depend on backbone
        layer1 / layer2, so they run on a side HIP stream underneath layer2 / layer3 instead of after layer3: their
        launch gaps and partially-filled rounds disappear from the critical path (fork/join is captured into the hipGraph)."""
        B = left_images.shape[0]
        images = (left_images, right_images)     # stacked on the batch axis by the stem's image pack (no cat copy)
        if not self.overlap_neck or not left_images.is_cuda:
            feats = self.backbone.forward_nhwc(images, dtype)
            return self.neck.forward_nhwc(feats, B)
        main = torch.cuda.current_stream()
        dev = left_images.device
        side = self._side_streams.get(dev)
        if side is None:
            side = self._side_streams[dev] = torch.cuda.Stream(device=dev)
        st = {}
        keep = []

        def on_stage(i, x):
            if i == 0:
                st.update(self.neck.alloc(x, B))
            if i in (0, 1):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    (self.neck.part_s4 if i == 0 else self.neck.part_s8)(x, st)
                x.record_stream(side)
                keep.append(x)
            elif i == 2:
                self.neck.part_s16(x, st)

        feats = self.backbone.forward_nhwc(images, dtype, on_stage=on_stage)
        assert len(feats) == 3, 'stereo neck needs the s4, s8 and s16 stages'
        main.wait_stream(side)
        for k in ('buf72', 'buf288', 'buf1152'):
            st[k].record_stream(side)
        return self.neck.part_merge(st)

    def forward(self, images):
        """Reference signature: ``images`` = [B,6,H,W] (left | right on the channel axis)."""
        feat = self.forward_nhwc(images[:, 0:3].contiguous(), images[:, 3:].contiguous())
        B, _, H, W = images.shape
        return dict(features=fused.to_nchw(feat), depth_output=torch.zeros([B, 1, H // 4, W // 4]))
