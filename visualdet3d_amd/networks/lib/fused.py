"""Shared machinery for the HIP-backed modules: lazily packed weights keyed on parameter identity/version, and
the NCHW-fp32 <-> NHWC boundary helpers used when a module is called stand-alone (module-level parity tests).

The product path keeps activations NHWC in the compute dtype from the stem to the head; only the detector's
inputs (NCHW fp32 images) and outputs (boxes) are in the reference's formats."""
import torch

from ... import hip_ops as ops

_DEFAULT_DTYPE = torch.bfloat16


def set_default_compute_dtype(dtype):
    """bf16 (default; BASELINE configs 2-4), fp16 (BASELINE config 5: KM3D and the mono detectors; the stereo-only kernels --
    PSM cosine volume, 3-D cost-volume convs -- are bf16 / fp32) or fp32 (strict-tolerance validation mode)."""
    global _DEFAULT_DTYPE
    assert dtype in (torch.bfloat16, torch.float16, torch.float32)
    _DEFAULT_DTYPE = dtype


def default_compute_dtype():
    return _DEFAULT_DTYPE


def _sig(*tensors):
    return tuple((t.data_ptr(), t._version, t.device, t.dtype) if t is not None else None for t in tensors)


class PackCache:
    """Re-packs when any source tensor was replaced / modified in place (load_state_dict, .cuda(), optimizers)."""

    def __init__(self):
        self._store = {}

    def get(self, key, sources, builder):
        sig = _sig(*sources)
        hit = self._store.get(key)
        if hit is None or hit[0] != sig:
            hit = (sig, builder())
            self._store[key] = hit
        return hit[1]


def bn_tuple(bn):
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


def bn_sources(bn):
    return [bn.weight, bn.bias, bn.running_mean, bn.running_var]


def to_nhwc(x, dtype=None):
    """NCHW fp32 -> NHWC compute dtype (boundary only)."""
    return ops.nchw_f32_to_nhwc(x, dtype or _DEFAULT_DTYPE)


def to_nchw(x):
    return ops.nhwc_to_nchw_f32(x)
