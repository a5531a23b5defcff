"""GPU: the XCD-aware tile walk of the persistent kernels (csrc/conv_resident.hip `xcd_tile_walk`, csrc/stem_pool.hip) only permutes WHICH
workgroup computes which tile: flipped in-process through the library's switch (`VD3D_PLAIN_TILE_WALK`, `vd3d_test_set_switch`), the plain
walk must give bit-identical results for the resident 64 -> 64 kernel, the small-channel streaming kernel and the fused stem, at sizes
with several rounds of a grid that is a multiple of 8 (the walk is only active then).  The partition property itself is a CPU test
(tests/test_host_logic.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ab(fn):
    from visualdet3d_amd import _lib
    a = fn()
    with _lib.test_switch('VD3D_PLAIN_TILE_WALK'):
        b = fn()
    c = fn()
    torch.cuda.synchronize()
    return a, b, c


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_resident64_and_conv_small_plain_walk_is_bit_identical(dtype):
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    for (B, H, W, Cin, Cout) in [(4, 96, 320, 64, 64),          # resident64: ResNet-34 layer1 at batch 2 (L and R stacked)
                                 (2, 128, 440, 16, 16),         # conv_small: DLA level 0
                                 (2, 100, 333, 32, 32)]:        # conv_small, ragged
        x = torch.randn(B, H, W, Cin, generator=g).cuda().to(dtype)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda()
        res = torch.randn(B, H, W, Cout, generator=g).cuda().to(dtype)
        pc = ops.pack_conv(w, None, None, dtype, 1, 1, 1)
        a, b, c = _ab(lambda: ops.conv2d(x, pc, residual=res, relu=True))
        assert torch.equal(a, b) and torch.equal(a, c), (B, H, W, Cin, Cout)
        assert float(a.float().abs().sum()) > 0


def test_stem_pool_plain_walk_is_bit_identical():
    from visualdet3d_amd import hip_ops as ops
    g = torch.Generator().manual_seed(6)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5, 1e-5)
    pc = ops.pack_stem_conv(w.cuda(), tuple(t.cuda() if torch.is_tensor(t) else t for t in bn), torch.bfloat16)
    for (B0, B1, H, W) in [(4, 4, 384, 1280), (3, 0, 96, 320)]:
        imgs = [torch.randn(B0, 3, H, W, generator=g).cuda()] + ([torch.randn(B1, 3, H, W, generator=g).cuda()] if B1 else [])
        a, b, c = _ab(lambda: ops.stem_conv_pool(imgs, pc, torch.bfloat16))
        assert torch.equal(a, b) and torch.equal(a, c), (B0, B1, H, W)
