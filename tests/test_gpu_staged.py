"""Kernels that are written and checked on the SIMT interpreter (tests/test_simt_kernels.py) but NOT yet measured on the GPU:
they are selected by environment variables only, the defaults do not reach them.  These tests run them on the GPU against the
same float64 truths as the shipping kernels; they are skipped unless DIAMOND_STAGED_TESTS=1 (tools/gpu/staged_wgrad.sh sets it),
so the regular `-m gpu` run covers exactly what ships."""
import os

import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("DIAMOND_STAGED_TESTS") != "1", reason="staged kernels: DIAMOND_STAGED_TESTS=1 runs them")]

PLANS = [dict(DIAMOND_WGRAD_MODE="2"), dict(DIAMOND_WGRAD_MODE="3"), dict(DIAMOND_WGRAD_MAX_WG="256"),
         dict(DIAMOND_WGRAD_MODE="3", DIAMOND_WGRAD_MAX_WG="7"),
         dict(DIAMOND_WGRAD_MODE="3", DIAMOND_WGRAD_MAX_WG="256", DIAMOND_WGRAD_SINGLE_REDUCE="256", DIAMOND_GN_BWD_FOLD="1"),
         dict(DIAMOND_WGRAD_MODE="3", DIAMOND_WGRAD_MAX_WG="256", DIAMOND_WGRAD_SINGLE_REDUCE="256", DIAMOND_GN_BWD_FOLD="1",
              DIAMOND_CONV_LATENCY_TILES="64")]


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


# ---- conv_lat_kernel (dmd_conv_lat.hip): the few-tile 3x3 for play.py's B = 1 sampler, routed by DIAMOND_CONV_LATENCY_TILES ------
def test_latency_conv_route_vs_reference_goldens(monkeypatch):
    """The denoiser at batch 1-2 with every eligible 3x3 on conv_lat_kernel: model output and quantised frames against the
    reference's goldens (same bars as the shipping route), the 3-step / Heun samplers teacher-forced, and the graphed B = 1
    sampler bitwise against the eager one."""
    from tests import test_gpu_env as EV, test_gpu_models as M

    monkeypatch.setenv("DIAMOND_CONV_LATENCY_TILES", "64")
    M.test_denoiser_vs_reference_golden("default", (0, 0, 0, 0), 2)
    M.test_denoiser_vs_reference_golden("attn0011", (0, 0, 1, 1), 1)
    ag = M.make_agent()
    M.test_sampler_teacher_forced_vs_golden(ag)
    M.test_denoiser_deterministic(ag)
    M.test_rew_end_model_vs_golden(ag)  # its 32-channel 3x3s take the route as well
    M.test_actor_critic_vs_golden(ag)
    M.test_full_window_vs_reference_golden()  # batch 4: the whole rollout on the route, integer trajectories bit-exact
    for name in dir(EV):
        if name.startswith("test_") and "graph" in name:
            getattr(EV, name)()


def test_latency_conv_route_is_taken_and_close_to_the_throughput_kernel(monkeypatch):
    """same conv, both routes: the kernel-name query reports the route, outputs agree to split-fp16 rounding, statistics too"""
    import torch

    from diamond_amd import engine as E, native as nv
    from tests.test_gpu_kernels import DEV, make_act

    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 32, 32, generator=g, dtype=torch.float64)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(DEV)
    b = torch.randn(64, generator=g).to(DEV)
    outs = {}
    for cap in ("0", "64"):
        monkeypatch.setenv("DIAMOND_CONV_LATENCY_TILES", cap)
        xa = make_act(x)
        y = E.conv2d([(xa, nv.PROLOGUE_NORM_SILU, E.NormSpec(mul=None, add=None))], nv.pack_conv_weight(w), b, 64,
                     w_f16=nv.pack_conv_weight_f16x2(w))
        torch.cuda.synchronize()
        outs[cap] = (y.t.clone(), y.stats.sum(2).clone())
    err = float((outs["0"][0] - outs["64"][0]).abs().max() / outs["0"][0].abs().max())
    assert 0 < err < 1e-5, err  # different summation order (not bitwise), same arithmetic
    assert torch.allclose(outs["0"][1], outs["64"][1], rtol=1e-5)
