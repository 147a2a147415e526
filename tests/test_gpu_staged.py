"""Kernels that are written and checked on the SIMT interpreter (tests/test_simt_kernels.py) but NOT yet measured on the GPU:
they are selected by environment variables only, the defaults do not reach them.  These tests run them on the GPU against the
same float64 truths as the shipping kernels; they are skipped unless DIAMOND_STAGED_TESTS=1 (tools/gpu/staged_wgrad.sh sets it),
so the regular `-m gpu` run covers exactly what ships."""
import os

import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("DIAMOND_STAGED_TESTS") != "1", reason="staged kernels: DIAMOND_STAGED_TESTS=1 runs them")]

PLANS = [dict(DIAMOND_WGRAD_MODE="2"), dict(DIAMOND_WGRAD_MODE="3"), dict(DIAMOND_WGRAD_MAX_WG="256"),
         dict(DIAMOND_WGRAD_MODE="3", DIAMOND_WGRAD_MAX_WG="7")]


@pytest.mark.parametrize("plan", PLANS, ids=lambda p: ",".join(f"{k[8:]}={v}" for k, v in p.items()))
def test_wgrad_staged_modes(plan, monkeypatch):
    """dmd_conv2d_wgrad's 32-pixel MFMA kernel (mode 2), its prefetching form (mode 3) and smaller workgroup plans."""
    from tests import test_gpu_kernels as K

    for k, v in plan.items():
        monkeypatch.setenv(k, v)
    for case in K.WGRAD_CASES:
        K.test_conv_wgrad(case)


@pytest.mark.parametrize("plan", PLANS[:2] + PLANS[3:], ids=lambda p: ",".join(f"{k[8:]}={v}" for k, v in p.items()))
def test_denoiser_training_step_staged_modes(plan, monkeypatch):
    """the denoiser training step's loss and all gradient tensors against the reference golden, weight gradients on the staged kernels"""
    from tests import test_gpu_models as M

    for k, v in plan.items():
        monkeypatch.setenv(k, v)
    fn = getattr(M, "test_denoiser_training_step_vs_reference_golden", None)
    assert fn is not None, "the golden training-step test moved: update tests/test_gpu_staged.py"
    fn()
