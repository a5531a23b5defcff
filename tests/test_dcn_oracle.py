"""CPU: the DCN restatement (oracle/dcn_ref.py) against golden columns / outputs produced by the reference's own
im2col device code compiled for the host (oracle/build_ref.sh, oracle/make_golden_native.py)."""
import numpy as np
import pytest

from oracle import dcn_ref
from oracle.make_golden_native import DCN_CASES, dcn_inputs
from tests.common import load_golden


@pytest.mark.parametrize('name', list(DCN_CASES))
def test_dcn_restatement_matches_reference_native_golden(name):
    g = load_golden('dcn_cases')
    x, off, mask, w, bias, kw = dcn_inputs(name)
    cols = dcn_ref.im2col(x, off, mask, w.shape[2], kw['stride'], kw['padding'], kw['dilation'], kw['deformable_groups'])
    assert np.allclose(cols.numpy(), g[name + '_cols'], rtol=1e-5, atol=1e-6)
    out = dcn_ref.deform_conv_forward(x, off, mask, w, bias, **kw)
    assert np.allclose(out.numpy(), g[name + '_out'], rtol=1e-4, atol=1e-5)
