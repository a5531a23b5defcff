"""ORACLE (test infrastructure only).  Import the *reference itself* on CPU in the build container.

Follows SURVEY.md Appendix B: the reference (``/root/reference``, read-only) hard-codes ``.cuda()`` and
depends on packages that are not installed here, so it is imported through stand-in packages
(``oracle/stubs``) plus a handful of monkey patches.  This is used ONLY to (a) validate the restatement in
``oracle/*.py`` and (b) generate the golden fixtures under ``tests/golden`` (``oracle/make_golden.py``).
``/root/reference`` does not exist on the GPU box, so nothing at test/bench run time may call this.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VD3D_REFERENCE_ROOT", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))
_loaded = False


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "visualDet3D"))


def load():
    """Make ``import visualDet3D`` work on CPU.  Returns the ``visualDet3D.networks`` module."""
    global _loaded
    import torch

    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    if not _loaded:
        stubs = os.path.join(_HERE, "stubs")
        repo = os.path.dirname(_HERE)
        for p in (REFERENCE_ROOT, stubs, repo):
            if p not in sys.path:
                sys.path.insert(0, p)
        # visualDet3D/networks/lib/ops/dcn/deform_conv.py:10 string-compares torch.version.cuda
        torch.version.cuda = "11.0"
        tb = types.ModuleType("torch.utils.tensorboard")
        tb.SummaryWriter = object
        sys.modules["torch.utils.tensorboard"] = tb
        for name, attrs in [
            ("visualDet3D.networks.lib.ops.dcn.deform_conv_ext", []),
            ("visualDet3D.networks.lib.ops.iou3d.iou3d_cuda",
             ["boxes_iou_bev_gpu", "boxes_overlap_bev_gpu", "nms_normal_gpu", "nms_gpu"]),
        ]:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, None)
            sys.modules[name] = m
        import numba  # the stub

        sys.modules["numba.cuda"] = numba.cuda
        # hard-coded .cuda() in PSM_cost_volume.py:56,88, anchors.py:113, testers.py:25,39
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        torch.cuda.synchronize = lambda *a, **k: None  # utils/timer.py:10,13
        _loaded = True
    import warnings

    warnings.filterwarnings("ignore", message=".*volatile was removed.*")
    import visualDet3D.networks as nets  # noqa: E402

    return nets


def detector_dict():
    load()
    from visualDet3D.networks.utils.registry import DETECTOR_DICT

    return DETECTOR_DICT
