"""CPU: the C-ABI shared library loads and exports exactly the symbols include/vd3d.h declares (no compute calls)."""
import os
import re
import subprocess

from visualdet3d_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, 'include', 'vd3d.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return set(re.findall(r'\b(vd3d_[a-z0-9_]+)\s*\(', text))


def test_header_symbols_are_exported_and_bound():
    declared = _declared()
    assert len(declared) >= 25
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r' T (vd3d_[a-z0-9_]+)', out))
    # test hooks (csrc/test_hooks.h) are exported but deliberately NOT part of the drop-in ABI
    hooks = {e for e in exported if e.startswith('vd3d_test_')}
    hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(REPO, 'visualdet3d_amd', 'csrc', 'test_hooks.h')).read(), flags=re.S)
    assert hooks == set(re.findall(r'\b(vd3d_test_[a-z0-9_]+)\s*\(', hdr)) == set(_lib.TEST_HOOKS)
    assert not any(d.startswith('vd3d_test_') for d in declared), 'test hooks must stay out of include/vd3d.h'
    exported -= hooks
    assert declared <= exported, 'declared but not exported: %s' % sorted(declared - exported)
    assert exported <= declared, 'exported but not declared in include/vd3d.h: %s' % sorted(exported - declared)
    assert set(_lib.SIGNATURES) == declared, sorted(set(_lib.SIGNATURES) ^ declared)
    assert not _lib.PENDING


def test_library_loads_and_reports_version():
    h = _lib.lib()
    assert h.vd3d_abi_version() == _lib.ABI_VERSION
    assert h.vd3d_last_error() is not None
    assert h.vd3d_head_workspace_bytes(8, 4096) > 8 * 4096 * 60
    assert h.vd3d_nms_bev_workspace_bytes(1000) >= 1000 * 16 * 8


def test_cpu_tensors_fail_loudly():
    import pytest
    import torch
    from visualdet3d_amd import hip_ops as ops
    with pytest.raises(_lib.Vd3dError):
        ops.maxpool3x3s2(torch.zeros(1, 8, 8, 8, dtype=torch.bfloat16))


def test_production_tile_list_is_the_one_the_tile_tests_force():
    """tests/test_conv_tiles_gpu.py forces every id of this list against the oracle; ids outside it are rejected by the
    product library (timing ablations / experimental tiles only exist in the -DVD3D_TUNING build)."""
    import ctypes as C
    ids = (C.c_int32 * 64)()
    n = _lib.lib().vd3d_conv2d_production_tiles(ids, 64)
    from tests.test_conv_tiles_gpu import PRODUCTION_TILES
    assert sorted(ids[i] for i in range(n)) == sorted(PRODUCTION_TILES)
    out = subprocess.run(['nm', '-C', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    # no ablation instantiation (template argument ABL != 0) in the product library: <..., (int)16, (int)0, (int)N, ...>
    assert not re.search(r'conv_igemm_dma_kernel<short, \d+, \d+, \d+, \d+, true, 16, 0, [1-9]', out)
