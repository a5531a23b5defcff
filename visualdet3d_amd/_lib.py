"""ctypes binding of libvd3d_hip.so (the C-ABI declared in include/vd3d.h).

The product path has NO fallback: if the shared library is missing or an entry point fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VD3D_TUNING_LIB=1 (tools/ only): the -DVD3D_TUNING build with the experimental tiles / timing ablations; a value ending in
# .so names an A/B build of the library NEXT TO THIS FILE (only the base name is used: a path in the environment must not be
# able to load an arbitrary shared object).  Loading anything but the product library is announced on stderr.
_T = os.environ.get('VD3D_TUNING_LIB', '')
if _T.endswith('.so'):
    if os.path.basename(_T) != _T:
        raise ImportError('VD3D_TUNING_LIB=%r: only the name of a library next to %s is accepted, not a path' % (_T, _HERE))
    _NAME = _T
else:
    _NAME = 'libvd3d_hip_tuning.so' if _T else 'libvd3d_hip.so'
LIB_PATH = os.path.join(_HERE, _NAME)
if _NAME != 'libvd3d_hip.so':
    import sys as _sys
    print('[visualdet3d_amd] WARNING: loading %s instead of the product library (VD3D_TUNING_LIB=%s): tuning builds contain '
          'timing ablations whose results are wrong by construction' % (_NAME, _T), file=_sys.stderr)

VD3D_BF16 = 0
VD3D_F32 = 1
VD3D_F16 = 2
ABI_VERSION = 6

c_void_p, c_int, c_int64, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_float


class ConvParams(C.Structure):
    _fields_ = [
        ('in_', c_void_p), ('weight', c_void_p), ('scale', c_void_p), ('shift', c_void_p),
        ('residual', c_void_p), ('out', c_void_p),
        ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('Cin', C.c_int32),
        ('in_pix_stride', C.c_int32), ('in_row_stride', C.c_int32),
        ('in_batch_stride', C.c_int64), ('in_bytes', C.c_int64),
        ('Ho', C.c_int32), ('Wo', C.c_int32), ('Cout', C.c_int32),
        ('out_pix_stride', C.c_int32), ('res_pix_stride', C.c_int32),
        ('kh', C.c_int32), ('kw', C.c_int32), ('stride', C.c_int32), ('pad', C.c_int32), ('dil', C.c_int32),
        ('Kpad', C.c_int32), ('CoutPad', C.c_int32), ('relu', C.c_int32), ('dtype', C.c_int32), ('out_f32', C.c_int32),
        ('weight_frag', c_void_p),
        ('splitk_ws', c_void_p), ('splitk_ws_bytes', C.c_int64),
    ]


class DcnParams(C.Structure):
    _fields_ = [
        ('in_', c_void_p), ('weight', c_void_p), ('bias', c_void_p), ('scale', c_void_p), ('shift', c_void_p),
        ('offset', c_void_p), ('mask', c_void_p), ('out', c_void_p),
        ('B', C.c_int32), ('C', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('O', C.c_int32),
        ('kh', C.c_int32), ('kw', C.c_int32), ('stride_h', C.c_int32), ('stride_w', C.c_int32),
        ('pad_h', C.c_int32), ('pad_w', C.c_int32), ('dil_h', C.c_int32), ('dil_w', C.c_int32),
        ('groups', C.c_int32), ('deformable_groups', C.c_int32), ('Kpad', C.c_int32), ('dtype', C.c_int32),
        ('mask_sigmoid', C.c_int32), ('relu', C.c_int32),
        ('in_strides', C.c_int64 * 4), ('offset_strides', C.c_int64 * 4), ('mask_strides', C.c_int64 * 4),
        ('out_strides', C.c_int64 * 4),
    ]


class Km3dParams(C.Structure):
    _fields_ = [(n, c_void_p) for n in ('hm', 'wh', 'hps', 'rot', 'dim', 'prob', 'reg', 'hm_hp', 'hp_offset', 'P2', 'kconst')] + [
        ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('n_cls', C.c_int32), ('n_joints', C.c_int32), ('K', C.c_int32),
        ('max_peaks', C.c_int32), ('img_h', C.c_int32), ('img_w', C.c_int32), ('score_thr', c_float), ('nms_iou_thr', c_float),
        ('workspace', c_void_p), ('out_scores', c_void_p), ('out_boxes', c_void_p), ('out_cls', c_void_p), ('out_count', c_void_p)]


class HeadParams(C.Structure):
    _fields_ = [
        ('cls', c_void_p), ('reg', c_void_p), ('anchors', c_void_p), ('prior_mean_std', c_void_p), ('P2', c_void_p),
        ('B', C.c_int32), ('N', C.c_int32), ('A', C.c_int32), ('n_cls', C.c_int32), ('n_types', C.c_int32),
        ('img_h', C.c_int32), ('img_w', C.c_int32),
        ('score_thr', c_float), ('nms_iou_thr', c_float),
        ('filter_y_min', c_float), ('filter_y_max', c_float), ('filter_x_max', c_float),
        ('use_filter', C.c_int32), ('max_cand', C.c_int32), ('max_det', C.c_int32),
        ('workspace', c_void_p), ('out_scores', c_void_p), ('out_boxes', c_void_p), ('out_labels', c_void_p),
        ('out_anchor', c_void_p), ('out_count', c_void_p),
    ]


# name -> (restype, argtypes); must list every symbol include/vd3d.h declares (tests/test_abi.py checks)
SIGNATURES = {
    'vd3d_abi_version': (c_int, []),
    'vd3d_last_error': (C.c_char_p, []),
    'vd3d_source_hash': (C.c_char_p, []),
    'vd3d_conv2d_igemm': (c_int, [C.POINTER(ConvParams), c_void_p]),
    'vd3d_conv2d_production_tiles': (c_int, [c_void_p, c_int]),
    'vd3d_conv2d_workspace_bytes': (c_int64, [C.POINTER(ConvParams)]),
    'vd3d_pack_image_nhwc4': (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    'vd3d_maxpool3x3s2': (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    'vd3d_avgpool2x2': (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    'vd3d_dwconv3x3': (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    'vd3d_copy_channels': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    'vd3d_nhwc_to_nchw_f32': (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    'vd3d_nchw_f32_to_nhwc': (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    'vd3d_psm_cosine': (c_int, [c_void_p] * 3 + [c_int] * 8 + [c_void_p]),
    'vd3d_costvol_build': (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p]),
    'vd3d_conv3d_3x3x3': (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_void_p]),
    'vd3d_cost_volume_fused': (c_int, [c_void_p] * 9 + [c_int] * 8 + [c_void_p]),
    'vd3d_head_workspace_bytes': (c_int64, [c_int, c_int]),
    'vd3d_head_postprocess': (c_int, [C.POINTER(HeadParams), c_void_p]),
    'vd3d_head_select': (c_int, [C.POINTER(HeadParams), c_void_p]),
    'vd3d_head_nms': (c_int, [C.POINTER(HeadParams), c_void_p]),
    'vd3d_pack_detections': (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_void_p, c_void_p]),
    'vd3d_nms': (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'vd3d_nms_workspace_bytes': (c_int64, [c_int]),
    'vd3d_boxes_overlap_bev': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'vd3d_boxes_iou_bev': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'vd3d_nms_bev': (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'vd3d_nms_bev_workspace_bytes': (c_int64, [c_int]),
    'vd3d_deform_conv_workspace_bytes': (c_int64, [c_int] * 5),
    'vd3d_deform_conv_forward': (c_int, [c_void_p] * 7 + [c_int] * 15 + [c_void_p]),
    'vd3d_dcn_pack_weight': (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    'vd3d_deform_conv': (c_int, [C.POINTER(DcnParams), c_void_p]),
    'vd3d_deform_columns': (c_int, [C.POINTER(DcnParams), c_void_p, c_void_p]),
    'vd3d_look_ground_sample': (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_float, c_float, c_int, c_void_p]),
    'vd3d_pack_image_nhwc': (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    'vd3d_image_conv7x7': (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    'vd3d_maxpool2x2': (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    'vd3d_dwconv_transpose': (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p]),
    'vd3d_km3d_workspace_bytes': (c_int64, [c_int] * 5),
    'vd3d_km3d_decode': (c_int, [C.POINTER(Km3dParams), c_void_p]),
    'vd3d_stem_conv_pool': (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    'vd3d_stem_conv_pool_f32': (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    'vd3d_kitti_postpath': (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    'vd3d_preprocess_image': (c_int, [c_void_p] + [c_int] * 5 + [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'vd3d_rotate_iou_eval': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vd3d_conv2d_pair': (c_int, [C.POINTER(ConvParams), C.POINTER(ConvParams), c_void_p]),
    'vd3d_conv2d_bottleneck': (c_int, [C.POINTER(ConvParams), C.POINTER(ConvParams), C.POINTER(ConvParams), C.POINTER(ConvParams), c_void_p]),
    'vd3d_km3d_head_fused': (c_int, [C.POINTER(ConvParams), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'vd3d_post_opt': (c_int, [c_void_p] * 4 + [c_int, c_int] + [c_float] * 3 + [c_int, c_void_p]),
}
# TEST hooks (csrc/test_hooks.h): exported by the library, NOT part of the drop-in ABI of include/vd3d.h
TEST_HOOKS = {
    'vd3d_test_force_conv_tile': (c_int, [c_int]),
    'vd3d_test_set_switch': (c_int, [C.c_char_p, c_int]),
    'vd3d_test_get_switch': (c_int, [C.c_char_p]),
}
# declared in include/vd3d.h, implemented later this round (moved into SIGNATURES as they land)
PENDING = {
}

_lib = None


class Vd3dError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Fails loudly when the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Vd3dError('%s not found -- run `python -m visualdet3d_amd.build` (hipcc, gfx950). '
                            'There is no CPU / PyTorch fallback for the HIP path.' % LIB_PATH)
        # torch bundles its own libamdhip64 (soname libamdhip64.so.7, NEEDED by torch as "libamdhip64.so").  Import torch
        # FIRST so the HIP runtime that owns torch's device context is the one this library binds to; loading
        # /opt/rocm's copy first would leave two HIP runtimes in the process ("no ROCm-capable device").
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(TEST_HOOKS.items()):
            fn = getattr(h, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        v = h.vd3d_abi_version()
        if v != ABI_VERSION:
            raise Vd3dError('libvd3d_hip.so ABI %d != expected %d; rebuild' % (v, ABI_VERSION))
        # the test hooks change which kernels the following launches run: every call bumps an epoch that the detectors' hipGraph
        # cache (networks/lib/graphed.py) keys on, so a captured graph never outlives the setting it was captured under
        for name in TEST_HOOKS:
            if name.startswith('vd3d_test_set') or name.startswith('vd3d_test_force'):
                setattr(h, name, _bumping(getattr(h, name)))
        _lib = h
    return _lib


_HOOK_EPOCH = 0


def hook_epoch():
    return _HOOK_EPOCH


def _bumping(fn):
    def call(*a):
        global _HOOK_EPOCH
        _HOOK_EPOCH += 1
        return fn(*a)
    return call


def check(rc, what):
    if rc != 0:
        msg = lib().vd3d_last_error()
        raise Vd3dError('%s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else ''))


class test_switch:
    """``with test_switch('VD3D_NO_LINE_STORE'):`` -- flip one of the library's A/B switches for the block; the
    library reads the environment only once per process, so tests go through the hook of csrc/test_hooks.h."""

    def __init__(self, name, on=True):
        self.name, self.on = name.encode(), int(bool(on))

    def __enter__(self):
        self.prev = lib().vd3d_test_get_switch(self.name)
        if self.prev < 0:
            raise Vd3dError('unknown switch %r' % self.name)
        check(lib().vd3d_test_set_switch(self.name, self.on), 'vd3d_test_set_switch')

    def __exit__(self, *exc):
        check(lib().vd3d_test_set_switch(self.name, self.prev), 'vd3d_test_set_switch')   # restore what was found, not 1 - on
