"""The kernels' own source, run lane by lane on the CPU (tests/simt: a SIMT interpreter, TEST INFRASTRUCTURE), against numpy
restatements of what each entry point of include/diamond_hip.h promises.  This checks index arithmetic, LDS layouts,
synchronisation structure and MFMA operand layouts without a GPU; the `-m gpu` tests remain the parity tests proper (the
interpreter's MFMA accumulation order is not the hardware's, and dmd_conv_f16ws.hip is not part of the host build)."""
import os

import numpy as np
import pytest

from tests.simt import loader as S


def test_product_refuses_the_host_build(monkeypatch):
    """diamond_amd has no CPU path: pointing it at the interpreter build must fail loudly."""
    import importlib
    S.lib()
    import diamond_amd.native as nv
    monkeypatch.setattr(nv, "LIB_PATH", S.LIB_PATH)
    monkeypatch.setattr(nv, "_lib", None)
    with pytest.raises(nv.NativeLibraryMissing, match="no CPU path"):
        nv.lib()


def _group_sums(x, hv=None, wv=None):
    n, h, w, c = x.shape
    v = x[:, :hv, :wv].astype(np.float64).reshape(n, -1, c // 32, 32)
    return np.stack([v.sum(axis=(1, 3)), (v * v).sum(axis=(1, 3))], axis=-1)  # (N, G, 2)


@pytest.mark.parametrize("shape", [(2, 8, 8, 64), (1, 5, 7, 32), (3, 16, 4, 96)])
def test_gn_stats(shape):
    rng = np.random.default_rng(0)
    x = rng.standard_normal(shape).astype(np.float32)
    n, h, w, c = shape
    stats = np.full((n, c // 32, 2), np.nan)
    S.check(S.lib().dmd_gn_stats(S.ptr(x), S.ptr(stats), n, h * w, c, None), "dmd_gn_stats")
    np.testing.assert_allclose(stats, _group_sums(x), rtol=1e-12, atol=1e-10)


def test_gn_stats_valid_extent():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 16, 16, 64)).astype(np.float32)
    stats = np.full((2, 2, 2), np.nan)
    S.check(S.lib().dmd_gn_stats_valid(S.ptr(x), S.ptr(stats), 2, 16, 16, 9, 13, 64, None), "dmd_gn_stats_valid")
    np.testing.assert_allclose(stats, _group_sums(x, 9, 13), rtol=1e-12, atol=1e-10)
    assert S.lib().dmd_gn_stats_valid(S.ptr(x), S.ptr(stats), 2, 16, 16, 17, 13, 64, None) != 0  # extent beyond the buffer
