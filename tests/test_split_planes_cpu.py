"""CPU checks of the host side of the two-plane (half precision) operand form used by the Winograd GEMMs and the direct
convolutions on the 16-bit matrix cores (engine.Engine.split_planes / winograd_weights_split; csrc/spa_gemm16.hip):
the planes reconstruct the scaled weights to 2^-22, the scales are exact powers of two, the layout is the kernel's."""
import importlib

import numpy as np
import pytest

torch = pytest.importorskip('torch')


def _engine():
    return importlib.import_module('superpixel-align_amd.engine').Engine


def test_split_planes_reconstruct_and_layout():
    Engine = _engine()
    g = torch.Generator().manual_seed(11)
    wt = torch.randn((64, 9, 96), generator=g) * 0.03
    wt2, inv_t = Engine.split_planes(wt)
    assert wt2.dtype == torch.float16 and tuple(wt2.shape) == (64, 9, 3, 2, 32)
    t = 1.0 / inv_t
    assert t == 2.0 ** round(np.log2(t))                         # a power of two
    top = float((wt.abs().max() * t))
    assert 2.0 ** 14 <= top < 2.0 ** 15
    h = wt2[:, :, :, 0, :].reshape(64, 9, 96).double()
    l = wt2[:, :, :, 1, :].reshape(64, 9, 96).double()
    exact = wt.double() * t
    # h is the nearest half-precision number, l the nearest to the remainder: 22 significand bits together
    assert torch.equal(h.float().half(), (wt.double() * t).float().half())
    err = (h + l - exact).abs()
    # ... for values whose remainder is a normal half-precision number; below, the remainder lands on the subnormal grid:
    # at most 2^-24 absolute = 2^-38 of the largest weight
    big = exact.abs() >= 1.0
    assert float((err[big] / exact.abs()[big]).max()) <= 2.0 ** -21
    assert float(err[~big].max()) <= 2.0 ** -24
    assert float(err.max()) <= 2.0 ** -21 * top


def test_winograd_weights_split_scales():
    Engine = _engine()
    g = torch.Generator().manual_seed(12)
    w = torch.randn((128, 64, 3, 3), generator=g) * 0.05
    u = Engine.winograd_weights(w, 4)
    u2, cs = Engine.winograd_weights_split(w)
    assert tuple(u2.shape) == (36, 128, 2, 2, 32) and cs.dtype == np.float32 and cs.shape == (36,)
    p = np.array([4, 4, 4, 3, 3, 4])
    for z in range(36):
        t = 2.0 ** float(p[z // 6] + p[z % 6]) / float(cs[z])
        assert t == 2.0 ** round(np.log2(t))
        top = float(u[z].abs().max()) * t
        assert 2.0 ** 14 <= top < 2.0 ** 15
        rec = (u2[z, :, :, 0, :].double() + u2[z, :, :, 1, :].double()).reshape(128, 64)
        assert float((rec - u[z].double() * t).abs().max()) <= 2.0 ** -21 * top
