"""``KM3DHead`` (heads/km3d_head.py:22-357) -- inference path -- with the reference's parameter names.

Forward: the nine heads' 3x3 convs share one input, so they run as ONE implicit GEMM 64 -> 9*256 (+bias, ReLU); the nine
1x1 output convs read their 256-channel slice of that buffer and write fp32 maps ``[B,H,W,n]`` (NHWC).
Decode (``_decode`` :155-252, ``get_bboxes`` :255-314, rtm3d_utils ``_nms/_topk/_topk_channel/gen_position``): three
kernels (csrc/km3d_decode.hip) -- peak extraction (sigmoid + 3x3 max test), per-channel top-K, and one workgroup per sample
doing class merge, gathers, keypoint association, the fp64 16x3 least squares, re-projection, clip, score mask and NMS.
The only host sync is reading the detection counts."""
import ctypes as C

import torch
import torch.nn as nn

from ... import _lib
from ... import hip_ops as ops
from ...utils.config import EasyDict
from ..lib import fused
from ..utils.utils import ClipBoxes


class _PositionLossStub(nn.Module):
    """Holds the ``const`` buffer the reference's Position_loss registers (rtm3d_utils.py:234-237): checkpoint-key parity."""

    def __init__(self):
        super(_PositionLossStub, self).__init__()
        self.register_buffer('const', torch.Tensor([[-1, 0], [0, -1]] * 8).unsqueeze(0).unsqueeze(0))

    def forward(self, *a, **k):
        raise NotImplementedError('training losses are out of scope of the inference path')


class KM3DHead(nn.Module):
    TOPK = 100

    def __init__(self, num_classes: int = 3, num_joints: int = 9, max_objects: int = 32, layer_cfg=EasyDict(),
                 loss_cfg=EasyDict(), test_cfg=EasyDict()):
        super(KM3DHead, self).__init__()
        self._init_layers(**layer_cfg)
        self.position_loss = _PositionLossStub()
        self.rampup_length = getattr(loss_cfg, 'rampup_length', 100)
        self.test_cfg = test_cfg
        const = torch.Tensor([[-1, 0], [0, -1]] * 8).unsqueeze(0).unsqueeze(0)
        self.register_buffer('const', const)
        self.num_classes, self.num_joints, self.max_objects = num_classes, num_joints, max_objects
        self.clipper = ClipBoxes()
        self._cache = fused.PackCache()
        # per (sample, heat-map channel) capacity of the peak list INSIDE the captured forward (<= 8192: sorted in LDS).  Not a limit of the head: a frame with
        # more local maxima above a threshold (a plateau makes every pixel one) is decoded again off-graph with a capacity that fits (decode_unbounded)
        self.max_peaks = 8192
        self._workspaces = {}
        self._workspace = None
        self.fuse_head = True        # bf16: the nine head branches in one launch (vd3d_km3d_head_fused)

    def _init_layers(self, input_features=256, head_features=64, head_dict=dict(), **kwargs):
        self.head_layers = nn.ModuleDict()
        for head_name, num_output in head_dict.items():
            self.head_layers[head_name] = nn.Sequential(
                nn.Conv2d(input_features, head_features, 3, padding=1, bias=True),
                nn.ReLU(inplace=True),
                nn.Conv2d(head_features, num_output, 1))
            out = self.head_layers[head_name][-1]
            if 'hm' in head_name:
                nn.init.constant_(out.bias, -2.19)
            else:
                nn.init.normal_(out.weight, std=0.001)
                nn.init.constant_(out.bias, 0)

    # ---- forward -------------------------------------------------------------------------------------------------
    def forward_nhwc(self, x):
        dt = x.dtype
        names = list(self.head_layers.keys())
        firsts = [self.head_layers[n][0] for n in names]

        def build_first():
            w = torch.cat([c.weight.detach().float() for c in firsts], dim=0)
            b = torch.cat([c.bias.detach().float() for c in firsts], dim=0)
            return ops.pack_conv(w, b, None, dt, 1, 1, 1)

        srcs = [t for c in firsts for t in (c.weight, c.bias)]
        pc = self._cache.get(('first', dt), srcs, build_first)
        F_ = firsts[0].out_channels
        lasts = [self.head_layers[n][2] for n in names]
        if (self.fuse_head and ops.is16(dt) and F_ == 256 and len(names) <= 9 and x.shape[3] % 64 == 0
                and all(c.out_channels <= 32 for c in lasts)):
            # one launch: the nine 3x3 convs as a single GEMM whose epilogue applies ReLU and the head's 1x1 conv; the
            # [B,H,W,9*256] intermediate (4 GB at 16 x 128 x 440) is never written
            def build_second():
                w2 = torch.zeros((len(names), 32, F_), dtype=torch.float32, device=x.device)
                b2 = torch.zeros((len(names), 32), dtype=torch.float32, device=x.device)
                for i, c in enumerate(lasts):
                    w2[i, :c.out_channels] = c.weight.detach().float().reshape(c.out_channels, F_)
                    b2[i, :c.out_channels] = c.bias.detach().float()
                return w2.to(dt).contiguous(), b2.contiguous()

            w2, b2 = self._cache.get(('second_fused', dt), [t for c in lasts for t in (c.weight, c.bias)], build_second)
            outs = ops.km3d_head_fused(x, pc, w2, b2, [c.out_channels for c in lasts])
            return dict(zip(names, outs))
        mid = ops.conv2d(x, pc, relu=True)                    # [B,H,W,9*F]
        ret = {}
        for i, n in enumerate(names):
            conv = self.head_layers[n][2]
            pco = self._cache.get((n, dt), [conv.weight, conv.bias], lambda conv=conv: ops.pack_conv(conv.weight, conv.bias, None, dt, 1, 0, 1))
            ret[n] = ops.conv2d(mid[..., i * F_:(i + 1) * F_], pco, relu=False, out_f32=True)   # fp32 [B,H,W,n]
        return ret

    def forward(self, x):
        """Reference signature: NCHW fp32 features -> dict of NCHW fp32 maps."""
        return {k: v.permute(0, 3, 1, 2).contiguous() for k, v in self.forward_nhwc(fused.to_nhwc(x)).items()}

    # ---- decode --------------------------------------------------------------------------------------------------
    def decode_unbounded(self, maps, P2, img_hw):
        """ONE frame (maps [1, H, W, n]) whose peak list did not fit ``max_peaks``: the same kernels, eagerly, with a doubled capacity until the frame
        fits -- at most every pixel of the map (beyond 8192 the lists are sorted in global memory, csrc/km3d_decode.hip).  The reference's ``_topk`` has no cap
        (rtm3d_utils.py:201-228).  -> (scores [n], boxes [n, 11], cls [n, 1] int64), private."""
        assert maps['hm'].shape[0] == 1
        H, W = maps['hm'].shape[1:3]
        limit = 1 << max(H * W - 1, 1).bit_length()
        cap = int(self.max_peaks)
        while True:
            cap = min(cap * 2, limit)
            scores, boxes, cls, count = self.get_bboxes_batched({k: v.contiguous() for k, v in maps.items()}, P2, img_hw, max_peaks=cap)
            k = int(count.item())
            if k >= 0:
                return scores[0, :k].clone(), boxes[0, :k].clone(), cls[0, :k].long().unsqueeze(-1)
            assert cap < limit, 'a capacity of every pixel of the map cannot overflow'

    def get_bboxes_batched(self, maps, P2, img_hw, max_peaks=None):
        """maps: dict of fp32 NHWC tensors.  Returns padded device tensors (scores [B,K], boxes [B,K,11], cls [B,K] i32,
        count [B] i32) -- no host sync.  ``max_peaks``: a capacity other than the head's (decode_unbounded): a private scratch buffer."""
        from ..._lib import Km3dParams
        hm = maps['hm']
        B, H, W, ncls = hm.shape
        dev = hm.device
        K = self.TOPK
        J = self.num_joints
        cap = int(max_peaks or self.max_peaks)
        # one scratch buffer per (batch, capacities, device), kept for the life of the head: hipGraphs bake its address, so it is never
        # replaced by a larger one (a later B = 16 call must not free the buffer a cached B = 1 graph writes on every replay)
        wkey = (B, ncls, J, cap, K, dev)
        ws = self._workspaces.get(wkey) if max_peaks is None else None
        if ws is None:
            need = _lib.lib().vd3d_km3d_workspace_bytes(B, ncls, J, cap, K)
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            if max_peaks is None and not torch.cuda.is_current_stream_capturing():     # (first seen inside a capture: private to that graph's pool)
                self._workspaces[wkey] = ws
        self._workspace = ws
        scores = torch.empty((B, K), dtype=torch.float32, device=dev)
        boxes = torch.empty((B, K, 11), dtype=torch.float32, device=dev)
        cls = torch.empty((B, K), dtype=torch.int32, device=dev)
        count = torch.empty((B,), dtype=torch.int32, device=dev)
        P2 = P2.to(device=dev, dtype=torch.float32).contiguous()
        p = Km3dParams()
        for name in ('hm', 'wh', 'hps', 'rot', 'dim', 'prob', 'reg', 'hm_hp', 'hp_offset'):
            t = maps[name]
            assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[:3] == (B, H, W)
            setattr(p, name, t.data_ptr())
        assert maps['hps'].shape[3] == 2 * J and maps['hm_hp'].shape[3] == J and maps['rot'].shape[3] == 8
        p.P2 = P2.data_ptr()
        kconst = self.const.detach().to(device=dev, dtype=torch.float32).contiguous()
        p.kconst = kconst.data_ptr()
        p.B, p.H, p.W, p.n_cls, p.n_joints, p.K, p.max_peaks = B, H, W, ncls, J, K, cap
        p.img_h, p.img_w = int(img_hw[0]), int(img_hw[1])
        p.score_thr = float(getattr(self.test_cfg, 'score_thr', 0.1))
        p.nms_iou_thr = float(getattr(self.test_cfg, 'nms_iou_thr', 0.5))
        p.workspace = self._workspace.data_ptr()
        p.out_scores, p.out_boxes, p.out_cls, p.out_count = scores.data_ptr(), boxes.data_ptr(), cls.data_ptr(), count.data_ptr()
        _lib.check(_lib.lib().vd3d_km3d_decode(C.byref(p), ops._stream()), 'vd3d_km3d_decode')
        return scores, boxes, cls, count

    @staticmethod
    def unpad(padded, own=False, retry=None):
        """``own=True``: views of private copies made BEFORE the host sync (see AnchorBasedDetection3DHead.unpad).  ``retry(b)``: called for a sample whose
        peak list overflowed the captured capacity; returns that sample's tuple (``decode_unbounded``); without it an overflow raises."""
        scores, boxes, cls, count = padded
        outs = []
        from ..lib.graphed import COPY_AFTER_SYNC, read_counts
        early = own and not COPY_AFTER_SYNC
        if early:
            scores, boxes, cls = scores.clone(), boxes.clone(), cls.long().unsqueeze(-1)
            rows = list(zip(scores.unbind(0), boxes.unbind(0), cls.unbind(0)))
        for b, k in enumerate(read_counts(count)):
            if k < 0:
                if retry is None:
                    raise RuntimeError('sample %d: more heat-map peaks than KM3DHead.max_peaks' % b)
                outs.append(retry(b))
                continue
            if early:
                s, bx, c = rows[b]
                outs.append((s[:k], bx[:k], c[:k]))
            else:
                outs.append((scores[b, :k], boxes[b, :k], cls[b, :k].long().unsqueeze(1)))
        if own and not early:
            outs = [(s.clone(), bx.clone(), l) for s, bx, l in outs]
        return outs

    def get_bboxes(self, output: dict, P2, img_batch=None):
        """Reference signature: ``output`` = dict of NCHW fp32 maps (as returned by ``forward``), batch 1."""
        maps = {k: v.permute(0, 2, 3, 1).contiguous().float() for k, v in output.items()}
        assert img_batch is not None
        return self.unpad(self.get_bboxes_batched(maps, P2, img_batch.shape[2:]), retry=lambda b: self.decode_unbounded(maps, P2, img_batch.shape[2:]))[0]
