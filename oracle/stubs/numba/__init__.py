"""numba stand-in: jit == identity (the reference's numba kernels are fp64 numpy code)."""


def jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


njit = jit


import numpy as _np


class _Arrays:
    """cuda.local.array / cuda.shared.array -> zero-initialised numpy arrays (the device functions of
    evaluator/kitti/rotate_iou.py then run as plain Python on numpy float32 storage)."""

    @staticmethod
    def array(shape, dtype=None):
        return _np.zeros(shape, dtype=dtype or _np.float32)


class _Cuda:
    jit = staticmethod(jit)
    local = _Arrays()
    shared = _Arrays()

    def __getattr__(self, n):
        return None


cuda = _Cuda()
float32, float64, int32, int64 = _np.float32, _np.float64, _np.int32, _np.int64
prange = range
