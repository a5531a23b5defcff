"""Anchor-based 3D detection heads with the reference's constructor and parameter names
(heads/detection_3d_head.py:21-100,218-263,310-400,500-533), inference path only.

Towers are fused implicit-GEMM convs on NHWC; the last conv of each tower writes fp32 ``[B,H,W,A*C]`` which *is* the
``AnchorFlatten`` layout.  ``get_bboxes`` runs entirely on the device for the whole batch
(``vd3d_head_postprocess``: ground filter, sigmoid, threshold, argmax, decode, clip, z-prior mask, NMS) and
synchronises with the host exactly once, to learn the detection counts.

Training-side members (``_assign``, ``_encode``, ``_sample``, ``loss``) are out of scope; the loss buffers
are registered so reference checkpoints load with identical keys."""
import numpy as np
import torch
import torch.nn as nn

from ... import hip_ops as ops
from ...utils.config import EasyDict
from ..backbones.resnet import BasicBlock
from ..lib import fused
from ..lib.blocks import AnchorFlatten, ConvBnReLU
from ..utils.utils import BackProjection, ClipBoxes
from .anchors import Anchors


class _LossStub(nn.Module):
    """Holds the buffers the reference's SigmoidFocalLoss registers (heads/losses.py) -- checkpoint-key parity only."""

    def __init__(self, balance_weights=None):
        super(_LossStub, self).__init__()
        if balance_weights is not None:
            self.register_buffer('balance_weights', balance_weights)

    def forward(self, *a, **k):
        raise NotImplementedError('training losses are out of scope of the inference path')


def _conv_pack(cache, key, conv, bn, dtype):
    srcs = [conv.weight, conv.bias] + (fused.bn_sources(bn) if bn is not None else [])
    return cache.get((key, dtype), srcs,
                     lambda: ops.pack_conv(conv.weight, conv.bias, fused.bn_tuple(bn) if bn is not None else None, dtype,
                                           conv.stride[0], conv.padding[0], conv.dilation[0]))


class AnchorBasedDetection3DHead(nn.Module):
    def __init__(self, num_features_in: int = 1024, num_classes: int = 3, num_regression_loss_terms=12,
                 preprocessed_path: str = '', anchors_cfg=EasyDict(), layer_cfg=EasyDict(), loss_cfg=EasyDict(),
                 test_cfg=EasyDict(), read_precompute_anchor: bool = True):
        super(AnchorBasedDetection3DHead, self).__init__()
        self.anchors = Anchors(preprocessed_path=preprocessed_path, readConfigFile=read_precompute_anchor, **anchors_cfg)
        self.num_classes = num_classes
        self.num_regression_loss_terms = num_regression_loss_terms
        self.decode_before_loss = getattr(loss_cfg, 'decode_before_loss', False)
        self.loss_cfg = loss_cfg
        self.test_cfg = test_cfg
        self.build_loss(**loss_cfg)
        self.backprojector = BackProjection()
        self.clipper = ClipBoxes()
        if getattr(layer_cfg, 'num_anchors', None) is None:
            layer_cfg['num_anchors'] = self.anchors.num_anchors
        self.init_layers(**layer_cfg)
        self._cache = fused.PackCache()
        self._side_streams = {}
        self.overlap_towers = True   # False: run the towers back to back on one stream (per-kernel profiling)
        # The overlap needs a LONG branch to pay: a forked launch inside a hipGraph costs ~13.6 us (profiles/r06_b1_l2_warm_experiment.txt).  With the mono head's
        # two towers of similar, short length (256-wide features) it loses 4.5 % at batch 1 - 4 (0.531 -> 0.507 ms per call), ties at 8, wins 2.5 % at 16
        # (tools/ab_overlap_b1.py); the stereo towers (1408 / 2176-wide reg tower against a 256-wide cls tower) win at every batch.  Below this many input
        # values (pixels x channels of the head's input) the towers run on one stream:
        self.overlap_min_elems = 2_500_000      # (mono 384x1280: batches 1 - 5 on one stream; the stereo head's 1408-wide input: every batch overlaps)
        # per-sample capacity of the device candidate list INSIDE the captured forward (a power of two; <= 8192 keeps the lists in LDS).  Not a
        # limit of the head: a frame with more candidates is re-run off-graph with a doubled capacity (get_bboxes_unbounded) -- the reference's
        # list has no cap (detection_3d_head.py:341-400)
        self.max_candidates = 4096
        self._workspaces = {}        # candidate scratch per (batch size, capacity, device): never reallocated (hipGraphs bake its address)
        self._workspace = None       # the one the last call used
        self.overlap_select = True   # candidate selection (needs only the cls logits) on the cls tower's side stream, under the reg tower
        self._preselected = None

    # ---- layers ---------------------------------------------------------------------------------------------
    def _cls_tower(self, num_features_in, cls_feature_size, num_anchors, num_cls_output):
        tower = nn.Sequential(
            nn.Conv2d(num_features_in, cls_feature_size, kernel_size=3, padding=1),
            nn.Dropout2d(0.3),
            nn.ReLU(inplace=True),
            nn.Conv2d(cls_feature_size, cls_feature_size, kernel_size=3, padding=1),
            nn.Dropout2d(0.3),
            nn.ReLU(inplace=True),
            nn.Conv2d(cls_feature_size, num_anchors * num_cls_output, kernel_size=3, padding=1),
            AnchorFlatten(num_cls_output),
        )
        tower[-2].weight.data.fill_(0)
        tower[-2].bias.data.fill_(0)
        return tower

    def init_layers(self, num_features_in, num_anchors: int, num_cls_output: int, num_reg_output: int,
                    cls_feature_size: int = 1024, reg_feature_size: int = 1024, **kwargs):
        """Base head: reg tower opens with DCNv2 (detection_3d_head.py:69-79)."""
        from ..lib.ops import ModulatedDeformConvPack
        self.cls_feature_extraction = self._cls_tower(num_features_in, cls_feature_size, num_anchors, num_cls_output)
        self.reg_feature_extraction = nn.Sequential(
            ModulatedDeformConvPack(num_features_in, reg_feature_size, 3, padding=1),
            nn.BatchNorm2d(reg_feature_size),
            nn.ReLU(inplace=True),
            nn.Conv2d(reg_feature_size, reg_feature_size, kernel_size=3, padding=1),
            nn.BatchNorm2d(reg_feature_size),
            nn.ReLU(inplace=True),
            nn.Conv2d(reg_feature_size, num_anchors * num_reg_output, kernel_size=3, padding=1),
            AnchorFlatten(num_reg_output),
        )
        self.reg_feature_extraction[-2].weight.data.fill_(0)
        self.reg_feature_extraction[-2].bias.data.fill_(0)

    def build_loss(self, focal_loss_gamma=0.0, balance_weight=[0], L1_regression_alpha=9, **kwargs):
        self.focal_loss_gamma = focal_loss_gamma
        self.register_buffer('balance_weights', torch.tensor(balance_weight, dtype=torch.float32))
        self.loss_cls = _LossStub(self.balance_weights)
        regression_weight = kwargs.get('regression_weight', [1 for _ in range(self.num_regression_loss_terms)])
        self.register_buffer('regression_weight', torch.tensor(regression_weight, dtype=torch.float))

    # ---- towers ---------------------------------------------------------------------------------------------
    def _cls_forward_nhwc(self, feat):
        t, dt = self.cls_feature_extraction, feat.dtype
        x = ops.conv2d(feat, _conv_pack(self._cache, 'cls0', t[0], None, dt), relu=True)
        x = ops.conv2d(x, _conv_pack(self._cache, 'cls3', t[3], None, dt), relu=True)
        x = ops.conv2d(x, _conv_pack(self._cache, 'cls6', t[6], None, dt), relu=False, out_f32=True)
        return t[7].forward_nhwc(x)

    def _reg_forward_nhwc(self, feat, inputs=None):
        t, dt = self.reg_feature_extraction, feat.dtype
        x = t[0].forward_nhwc(feat, bn=t[1], relu=True)     # DCNv2 + BN + ReLU
        x = ops.conv2d(x, _conv_pack(self._cache, 'reg3', t[3], t[4], dt), relu=True)
        x = ops.conv2d(x, _conv_pack(self._cache, 'reg6', t[6], None, dt), relu=False, out_f32=True)
        return t[7].forward_nhwc(x)

    def forward_nhwc(self, inputs):
        """The two towers are independent: the (small) cls tower runs on a side HIP stream so its workgroups fill the
        CUs the reg tower's partially-filled last tile rounds leave idle (fork/join is captured into the hipGraph)."""
        feat = inputs['features']
        self._p2_conv = None         # a calibration conversion is shared by the stages of ONE forward only (see _device_p2)
        if not self.overlap_towers or feat.numel() < self.overlap_min_elems:
            return self._cls_forward_nhwc(feat), self._reg_forward_nhwc(feat, inputs)
        main = torch.cuda.current_stream()
        side = self._side_streams.get(feat.device)
        if side is None:
            side = self._side_streams[feat.device] = torch.cuda.Stream(device=feat.device)
        self._preselected = None
        early = self.overlap_select and not self.training and inputs.get('P2') is not None and inputs.get('image') is not None
        if early:
            # on the MAIN stream, before the fork: the calibration tensor in its device fp32 form and the candidate workspace -- a
            # tensor allocated while the side stream is current would belong to that stream's pool, and a host / non-fp32 P2 would be
            # converted (a synchronous H2D copy) inside the overlapped region and never match the key of the later NMS stage
            self._device_p2(inputs['P2'], feat.device)
            self._ensure_workspace(feat.shape[0], feat.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            cls_preds = self._cls_forward_nhwc(feat)
            if early:
                # stage 1 of get_bboxes (ground filter + sigmoid + threshold -> candidate lists) only reads the class logits: it runs
                # here, behind the cls tower, while the reg tower still computes on the main stream (-20 us on the critical path)
                self._preselected = self._select(cls_preds, inputs['P2'], inputs['image'].shape[2:], clip=True)
        reg_preds = self._reg_forward_nhwc(feat, inputs)
        main.wait_stream(side)
        cls_preds.record_stream(main)
        feat.record_stream(side)
        if self._workspace is not None:
            self._workspace.record_stream(side)      # allocated on the main stream (above), also used on the side stream
        return cls_preds, reg_preds

    def forward(self, inputs):
        """Reference signature: ``inputs['features']`` is NCHW fp32."""
        d = dict(inputs)
        d['features'] = fused.to_nhwc(inputs['features'])
        return self.forward_nhwc(d)

    # ---- anchors ----------------------------------------------------------------------------------------------
    def _is_filtering(self):
        is_filtering = getattr(self.loss_cfg, 'filter_anchor', True)
        if not self.training:
            is_filtering = getattr(self.test_cfg, 'filter_anchor', is_filtering)
        return is_filtering

    def get_anchor(self, img_batch, P2):
        self._last_img_hw = (int(img_batch.shape[2]), int(img_batch.shape[3]))
        anchors, useful_mask, anchor_mean_std = self.anchors(img_batch, P2, is_filtering=self._is_filtering())
        return dict(anchors=anchors, mask=useful_mask, anchor_mean_std_3d=anchor_mean_std)

    # ---- post-processing ------------------------------------------------------------------------------------
    def get_bboxes_batched(self, cls_preds, reg_preds, P2s, img_hw, clip=True):
        """Device-side ``get_bboxes`` for B samples.  Returns padded device tensors
        (scores [B,K], boxes [B,K,11], labels [B,K] int32, anchor_idx [B,K] int32, count [B] int32); no host sync.
        ``clip=False``: skip ClipBoxes (the reference's ``img_batch is None``, detection_3d_head.py:375-376)."""
        if not getattr(self.test_cfg, 'cls_agnositc', True):     # (sic) the reference's key, detection_3d_head.py:381
            # the reference's per-class NMS branch cannot run either (`label.float().unsqueeze()` without a dim, :388-389)
            raise NotImplementedError('test_cfg.cls_agnositc=False (per-class NMS) is not implemented: the reference branch '
                                      'itself raises TypeError (detection_3d_head.py:388); only class-agnostic NMS is on the path')
        self._last_img_hw = (int(img_hw[0]), int(img_hw[1]))
        args, kw = self._post_args(cls_preds, P2s, img_hw, clip)
        pre = self._preselected
        self._preselected = None
        preselected = pre is not None and pre == self._select_key(cls_preds, args, kw)
        padded = ops.head_postprocess(cls_preds, reg_preds, *args, preselected=preselected, **kw)
        if getattr(self.test_cfg, 'post_optimization', False):
            # detection_3d_head.py:396-398 -> _post_process, here on the padded batch (count < 0 rows are skipped)
            from ..lib.fast_utils.hill_climbing import post_opt_batch
            post_opt_batch(padded[1], padded[2], P2s, counts=padded[4])
        return padded

    def _post_args(self, cls_preds, P2s, img_hw, clip):
        """positional (anchors, prior, P2, A, n_cls, n_types, img_hw, score_thr, nms_iou_thr) + keyword arguments shared by
        ops.head_select / ops.head_postprocess; (re)allocates the candidate workspace."""
        dev = cls_preds.device
        anchors, prior, A = self.anchors.device_tables(img_hw, dev)
        self._ensure_workspace(cls_preds.shape[0], dev)
        lo, hi = self.anchors.filter_y_threshold_min_max
        P2s = self._device_p2(P2s, dev)
        args = (anchors, prior, P2s, A, self.num_classes, len(self.anchors.obj_types), tuple(int(v) for v in img_hw) if clip else (0, 0),
                getattr(self.test_cfg, 'score_thr', 0.5), getattr(self.test_cfg, 'nms_iou_thr', 0.5))
        kw = dict(use_filter=bool(self._is_filtering() and self.anchors.readConfigFile), y_min_max=(lo, hi),
                  x_max=self.anchors.filter_x_threshold, max_cand=self.max_candidates, workspace=self._workspace)
        return args, kw

    def _ensure_workspace(self, B, dev):
        """The candidate scratch of a B-frame call.  One buffer per (B, capacity, device), kept for the life of the head: a captured
        hipGraph bakes the address into its launches, so a buffer that a later, larger call replaced (and handed back to the caching
        allocator) would be written by every replay of the older graph.  A few hundred KB per batch size."""
        key = (int(B), int(self.max_candidates), dev)
        ws = self._workspaces.get(key)
        if ws is None:
            need = ops._lib.lib().vd3d_head_workspace_bytes(B, self.max_candidates)
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            if not torch.cuda.is_current_stream_capturing():     # (first seen inside a capture: it lives in that graph's private pool
                self._workspaces[key] = ws                       #  like the activations do, and is not shared with anything else)
        self._workspace = ws
        return ws

    def _device_p2(self, P2s, dev):
        """The calibration matrices as a contiguous fp32 device tensor.  Already in that form: returned as is.  Otherwise converted once
        per FORWARD (``forward_nhwc`` clears the slot), so that the early selection and the NMS stage of one forward see the very same
        tensor (the preselection key holds its data_ptr).  The slot never outlives a forward and never crosses a capture boundary: a
        conversion cached during the eager warm-up passes and then HIT inside the capture would leave the conversion out of the graph,
        and every replay would decode with the warm-up frame's calibration."""
        if P2s.device == dev and P2s.dtype == torch.float32 and P2s.is_contiguous():
            return P2s
        capturing = torch.cuda.is_current_stream_capturing() if P2s.is_cuda or dev.type == 'cuda' else False
        hit = getattr(self, '_p2_conv', None)
        if hit is not None and hit[0] is P2s and hit[1] == P2s._version and hit[2].device == dev and hit[3] == capturing:
            return hit[2]
        conv = P2s.to(device=dev, dtype=torch.float32).contiguous()
        self._p2_conv = (P2s, P2s._version, conv, capturing)
        return conv

    @staticmethod
    def _select_key(cls_preds, args, kw):
        """what a preselection is valid for: the very logits tensor, calibration tensor, tables and thresholds"""
        anchors, prior, P2s = args[0], args[1], args[2]
        return (cls_preds.data_ptr(), tuple(cls_preds.shape), anchors.data_ptr(), prior.data_ptr(), P2s.data_ptr(), args[3:],
                kw['use_filter'], kw['y_min_max'], kw['x_max'], kw['max_cand'], kw['workspace'].data_ptr())

    def _select(self, cls_preds, P2s, img_hw, clip):
        args, kw = self._post_args(cls_preds, P2s, img_hw, clip)
        self._select_p2 = args[2]                    # keep the (possibly converted) calibration tensor alive until the NMS stage
        ops.head_select(cls_preds, *args, **kw)
        return self._select_key(cls_preds, args, kw)

    def get_bboxes_unbounded(self, cls_preds, reg_preds, P2s, img_hw, clip=True):
        """ONE frame ([1, N, ...] logits) whose candidate list (or detection list) did not fit ``max_candidates``: the same two kernels, launched
        eagerly with a doubled capacity until the frame fits -- at most every anchor of the frame (capacities beyond 8192 sort in global memory,
        csrc/postprocess.hip).  The reference has no cap (boolean indexing, then nms: detection_3d_head.py:341-400), so neither has this head;
        the capacity only bounds what a captured hipGraph handles without this detour.  -> (scores [n], boxes [n, 11], labels [n] int64), private."""
        assert cls_preds.shape[0] == 1
        N = int(cls_preds.shape[1])
        dev = cls_preds.device
        anchors, prior, A = self.anchors.device_tables(img_hw, dev)
        lo, hi = self.anchors.filter_y_threshold_min_max
        P2s = P2s.to(device=dev, dtype=torch.float32).contiguous()
        limit = 1 << max(N - 1, 1).bit_length()
        cap = int(self.max_candidates)
        while True:
            cap = min(cap * 2, limit)
            padded = ops.head_postprocess(cls_preds.float().contiguous(), reg_preds.float().contiguous(), anchors, prior, P2s, A, self.num_classes,
                                          len(self.anchors.obj_types), tuple(int(v) for v in img_hw) if clip else (0, 0),
                                          getattr(self.test_cfg, 'score_thr', 0.5), getattr(self.test_cfg, 'nms_iou_thr', 0.5),
                                          use_filter=bool(self._is_filtering() and self.anchors.readConfigFile), y_min_max=(lo, hi),
                                          x_max=self.anchors.filter_x_threshold, max_cand=cap, workspace=None)
            if getattr(self.test_cfg, 'post_optimization', False):
                from ..lib.fast_utils.hill_climbing import post_opt_batch
                post_opt_batch(padded[1], padded[2], P2s, counts=padded[4])
            k = int(padded[4].item())
            if k >= 0:
                return padded[0][0, :k].clone(), padded[1][0, :k].clone(), padded[2][0, :k].long()
            assert cap < limit, 'a capacity of every anchor of the frame cannot overflow'

    @staticmethod
    def unpad(padded, own=False, retry=None):
        """One host sync: slice the padded batch results into per-sample (scores, boxes, labels int64) tuples.
        ``own=True`` (the detectors' ``test_forward``: the padded tensors are a hipGraph's static outputs, overwritten by the next replay): the
        results are views of PRIVATE copies of the padded arrays, made BEFORE the sync -- three launches per call that the host issues while the
        GPU still works on the forward, instead of three per SAMPLE after the sync (slices, ``.long()``, clones).  NB: the per-sample results of
        one call are then VIEWS of one batch-wide copy (``[B, max_det, ...]``): holding one sample keeps the batch's copy alive, and
        ``untyped_storage()`` of a sample is the whole batch -- ``.clone()`` a sample that outlives the call or crosses a process boundary.
        ``retry(b)``: called for a sample whose candidate list overflowed the captured capacity (count < 0); returns that sample's tuple
        (``get_bboxes_unbounded``).  Without it an overflow raises."""
        scores, boxes, labels, aidx, count = padded
        from ..lib.graphed import COPY_AFTER_SYNC, read_counts
        early = own and not COPY_AFTER_SYNC
        if early:
            scores, boxes, labels = scores.clone(), boxes.clone(), labels.long()
            rows = list(zip(scores.unbind(0), boxes.unbind(0), labels.unbind(0)))      # per-sample rows, still before the sync
        counts = read_counts(count)
        outs = []
        for b, k in enumerate(counts):
            if k < 0:
                if retry is None:
                    raise RuntimeError('sample %d: more candidates than max_candidates (or detections than max_det); '
                                       'raise AnchorBasedDetection3DHead.max_candidates' % b)
                outs.append(retry(b))
                continue
            if early:                            # after the sync: three slices per sample, nothing else (this is the call's critical path)
                s, bx, l = rows[b]
                outs.append((s[:k], bx[:k], l[:k]))
            else:
                outs.append((scores[b, :k], boxes[b, :k], labels[b, :k].long()))
        if own and not early:
            outs = [(s.clone(), bx.clone(), l) for s, bx, l in outs]
        return outs

    def get_bboxes(self, cls_scores, reg_preds, anchors, P2s, img_batch=None):
        """Reference signature (batch 1, detection_3d_head.py:341).  Limits relative to the reference, all explicit:
          * ``anchors`` (the ``get_anchor`` dict) is accepted for compatibility but NOT read: the kernel uses the cached anchor /
            prior tables of the image shape and evaluates the ground filter from ``P2s`` itself, so a caller-modified
            ``anchors['mask']`` has no effect (the reference's only caller passes ``get_anchor``'s output unchanged);
          * ``img_batch=None`` skips ClipBoxes like the reference; the image shape (needed for the anchor grid) is then the one
            of the last ``get_anchor`` / ``get_bboxes`` call;
          * ``test_cfg.cls_agnositc=False`` raises NotImplementedError (see get_bboxes_batched);
          * single pyramid level (every shipped config uses ``pyramid_levels=[4]``; ``Anchors.device_tables`` asserts it)."""
        assert cls_scores.shape[0] == 1
        if img_batch is not None:
            img_hw = img_batch.shape[2:]
        else:
            img_hw = getattr(self, '_last_img_hw', None)
            if img_hw is None:
                raise RuntimeError('get_bboxes(img_batch=None): image shape unknown -- call get_anchor(img_batch, P2) first '
                                   '(the reference flow, yolostereo3d_detector.py:90-93)')
        cls_scores, reg_preds = cls_scores.float().contiguous(), reg_preds.float().contiguous()
        padded = self.get_bboxes_batched(cls_scores, reg_preds, P2s, img_hw, clip=img_batch is not None)
        return self.unpad(padded, retry=lambda b: self.get_bboxes_unbounded(cls_scores, reg_preds, P2s, img_hw, clip=img_batch is not None))[0]

    def _post_process(self, scores, bboxes, labels, P2s):
        """heads/detection_3d_head.py:294-308: hill-climb the yaw of every label-0 box deeper than 3 m so that its
        projection matches the 2D box.  One device launch for all boxes instead of the reference's per-box host loop."""
        from ..lib.fast_utils.hill_climbing import post_opt_batch
        bboxes = post_opt_batch(bboxes.float().contiguous(), labels, P2s[0:1])
        return scores, bboxes, labels


class StereoHead(AnchorBasedDetection3DHead):
    """heads/detection_3d_head.py:500-533: reg tower = ConvBnReLU -> BasicBlock -> ReLU -> conv."""

    def init_layers(self, num_features_in, num_anchors: int, num_cls_output: int, num_reg_output: int,
                    cls_feature_size: int = 1024, reg_feature_size: int = 1024, **kwargs):
        self.cls_feature_extraction = self._cls_tower(num_features_in, cls_feature_size, num_anchors, num_cls_output)
        self.reg_feature_extraction = nn.Sequential(
            ConvBnReLU(num_features_in, reg_feature_size, (3, 3)),
            BasicBlock(reg_feature_size, reg_feature_size),
            nn.ReLU(),
            nn.Conv2d(reg_feature_size, num_anchors * num_reg_output, kernel_size=3, padding=1),
            AnchorFlatten(num_reg_output),
        )
        self.reg_feature_extraction[-2].weight.data.fill_(0)
        self.reg_feature_extraction[-2].bias.data.fill_(0)

    def _reg_forward_nhwc(self, feat, inputs=None):
        t, dt = self.reg_feature_extraction, feat.dtype
        x = t[0].forward_nhwc(feat)
        x = t[1].forward_nhwc(x)      # ends in ReLU; the extra nn.ReLU (t[2]) is a no-op
        x = ops.conv2d(x, _conv_pack(self._cache, 'reg3', t[3], None, dt), relu=False, out_f32=True)
        return t[4].forward_nhwc(x)
