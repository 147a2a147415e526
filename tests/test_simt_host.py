"""The product's HOST-side orchestration on the CPU: engine.py / blocks.py / inner_model.py / denoiser.py / unet_train.py drive
the SIMT-interpreter build of the kernels (tests/simt: TEST INFRASTRUCTURE) on CPU tensors, and whole-network results are held
against the REFERENCE's goldens -- the same fixtures and the same bars as the `-m gpu` tests, whose functions are reused with
their device patched to "cpu".  This covers what the kernel-level interpreter tests cannot: parameter packing through
PackCache / dmd_pack_jobs, the batched FiLM table and its strides, statistics plumbing between producers and consumers, the
fused-projection and fused-8x8-level routing, the recorded-tape backward.  The product itself cannot run this way
(tests/simt/host_harness.py patches the two guards that prevent it, for the duration of a test)."""
import os

import pytest
import torch

from tests.simt.host_harness import engine_on_interpreter


class _LaunchCounter:
    """stands in for native.PROFILER: counts launches per kernel key (no timing)"""

    def __init__(self):
        self.n = {}
        self._pending = None

    def annotate(self, key, flops, nbytes):
        self._pending = key

    def call(self, name, fn, args):
        key, self._pending = self._pending or name, None
        self.n[key] = self.n.get(key, 0) + 1
        return fn(*args)


@pytest.fixture
def models(monkeypatch):
    from diamond_amd import native as nv
    from tests import test_gpu_models as M

    monkeypatch.setattr(M, "DEV", "cpu")
    # quantised frames: the interpreter's exp2f / division / MFMA summation order are the host's, not the device's, and a value
    # within ~1e-6 of a rounding boundary lands on the other side: 3 of 24,576 pixels at one of the four sigmas where the
    # device has at most 2 (the contract's 1e-4).  The fp32 bars (model output, gradients: 1e-4) are NOT relaxed.
    strict = M.check_quantised
    monkeypatch.setattr(M, "check_quantised", lambda a, b, max_frac, what="": strict(a, b, max_frac=2 * max_frac, what=what))
    counter = _LaunchCounter()
    monkeypatch.setattr(nv, "PROFILER", counter)
    with engine_on_interpreter():
        yield M, counter


def test_denoiser_vs_reference_golden_on_the_interpreter(models, monkeypatch):
    M, counter = models
    M.test_denoiser_vs_reference_golden("attn0011", (0, 0, 1, 1), 1)
    keys = "\n".join(counter.n)
    assert "conv_f16ws_kernel<WsGeom<false, 2, 9>>" in keys and "conv_f16ws_kernel<WsGeomProj>" in keys, keys


@pytest.mark.parametrize("env", [{}, {"DIAMOND_WGRAD_MAX_WG": "7"}], ids=["shipping", "few-workgroups"])
def test_denoiser_training_step_vs_reference_golden_on_the_interpreter(models, dmd_env, env):
    """loss and all 236 gradient tensors of Denoiser.forward + backward (split-fp16 arithmetic): the shipping plan, and 7
    workgroups walking many tiles each"""
    if env and os.environ.get("DIAMOND_SLOW_CPU_TESTS") != "1":
        pytest.skip("a second 55 s end-to-end run: DIAMOND_SLOW_CPU_TESTS=1 runs it; the kernels' plans are in test_simt_kernels.py")
    M, counter = models
    dmd_env(**env)
    errs = M.denoiser_training_step_errors(("f16x2",))["f16x2"]
    bad = {k: v for k, v in errs.items() if v >= 1e-4}
    assert not bad, bad
    assert sum(v for k, v in counter.n.items() if k.startswith("wgrad_ps_kernel<")) > 100 and counter.n.get("dmd_gn_silu_bwd", 0) > 50, counter.n
    # (the reductions of those weight gradients: deferred to one table per backward, three launches of <= 32 jobs each)
    assert 1 <= counter.n.get("dmd_wgrad_reduce_jobs", 0) <= 4, counter.n


def test_deferred_wgrad_reductions_are_bitwise_on_the_interpreter(models, monkeypatch):
    M, counter = models
    M.test_deferred_wgrad_reductions_leave_every_gradient_bitwise()
    M.test_deferred_wgrad_reductions_leave_the_actor_critic_gradients_bitwise(M.make_agent(), monkeypatch)
    assert counter.n.get("dmd_wgrad_reduce_jobs", 0) >= 1, counter.n


def test_film_tables_of_a_frame_computed_together_on_the_interpreter(models, monkeypatch):
    M, _ = models
    M.test_film_tables_of_a_frame_computed_together_are_bitwise_the_per_step_ones(M.make_agent(), monkeypatch, 3, 1, 0.0, 2)


def test_rew_end_model_and_actor_critic_vs_goldens_on_the_interpreter(models):
    """reward / end model (32-channel AdaGN encoder, fused 8x8 tail, LSTM, head) and the actor-critic (forward + every gradient)
    against the reference-generated goldens"""
    M, counter = models
    ag = M.make_agent()
    M.test_rew_end_model_vs_golden(ag)
    M.test_actor_critic_vs_golden(ag)
    assert counter.n.get("dmd_lowres_chain32", 0) >= 2 and counter.n.get("dmd_lstm_pointwise_bwd", 0) >= 1, counter.n


@pytest.mark.skipif(os.environ.get("DIAMOND_SLOW_CPU_TESTS") != "1", reason="2-4 minutes on 8 cores: DIAMOND_SLOW_CPU_TESTS=1 runs it")
def test_full_window_vs_reference_golden_on_the_interpreter(models):
    """The whole north-star path on the CPU: two BPTT windows of ActorCritic.forward() + backward through WorldModelEnv /
    env_loop / DiffusionSampler / reward-end model with resets and burn-in, the reference's RNG order -- sampled actions,
    rewards, ends and truncations BIT-exact against the reference-generated golden, frames on its uint8 levels."""
    M, counter = models
    M.test_full_window_vs_reference_golden()
    assert counter.n.get("dmd_categorical_sample", 0) >= 12 and counter.n.get("lowres_chain_kernel", 0) >= 36, counter.n


@pytest.mark.skipif(os.environ.get("DIAMOND_SLOW_CPU_TESTS") != "1", reason="several minutes: DIAMOND_SLOW_CPU_TESTS=1 runs it")
def test_slots_window_is_bitwise_the_sequential_one_on_the_interpreter(models, monkeypatch):
    """env_loop's default form through the REAL WorldModelEnv (deaths resolved into reset slots by dmd_resolve_deaths / dmd_reset_slots,
    the host one step behind, windows repeated from their snapshot after a slot overflow) against the sequential order, on the CPU"""
    M, counter = models
    M.test_slots_window_is_bitwise_the_sequential_one(monkeypatch, 3, 4, 0.15, True, 0.9)
    stats = dict(M.LAST_SLOTS_STATS)
    assert counter.n.get("dmd_resolve_deaths", 0) >= 9 and counter.n.get("dmd_reset_slots", 0) >= 9 and stats["dead_rows"] > 0, (counter.n, stats)


def test_weight_audit_on_the_interpreter(monkeypatch):
    """engine.WeightAudit / dmd_checksums (the always-on guard against silent parameter writes) on the CPU: the GPU tests' own
    functions with their device patched"""
    from tests import test_gpu_pack_jobs as P

    monkeypatch.setattr(P, "DEV", "cpu")
    with engine_on_interpreter():
        P.test_a_silent_parameter_write_is_detected_by_the_audit("conv.weight", "data.copy_")
        P.test_a_silent_parameter_write_is_detected_by_the_audit("lstm.weight_ih", "data.mul_")
        P.test_a_silent_parameter_write_is_detected_by_the_audit("norm.linear.weight", "set_ under no_grad via .data view")
        P.test_the_audit_raises_no_false_alarm(monkeypatch)


def test_fused_burn_in_on_the_interpreter(models):
    """lstm_native.LstmBurnInFn (one autograd node for the policy-side burn-in of a reset) against chained per-frame calls"""
    M, counter = models
    M.test_fused_burn_in_is_bitwise_the_frame_by_frame_one()


def test_torch_compile_of_the_env_callables_is_a_no_op(models):
    """trainer.py:182-184 under the reference's DEFAULT configuration (config/trainer.yaml:60, compile_wm: True) re-assigns
    rl_env.predict_next_obs / predict_rew_end with torch.compile(..., mode="reduce-overhead") objects.  They are HIP launch
    sequences: the env undoes the wrapper on assignment -- dynamo never sees a frame of the host code (it used to trace it for a
    minute, hit its recompilation limit and move the initial-noise draw into a compiled region), results and the random stream
    are bitwise those of the unwrapped env.  Plain wrappers (spies) are kept and called, like an attribute of the reference's env."""
    import torch._dynamo.utils as U

    import diamond_amd as D

    M, counter = models
    ag = M.make_agent()

    def rollout(compiled):
        env = D.WorldModelEnv(ag.denoiser, ag.rew_end_model, M._Loader(1, 21, 64),
                              D.WorldModelEnvConfig(horizon=2, num_batches_to_preload=1, diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=1)))
        env.sampler.noise_fn = lambda shape, dev: torch.randn(*shape)
        calls = []
        if compiled == "torch.compile":
            env.predict_next_obs = torch.compile(env.predict_next_obs, mode="reduce-overhead")
            env.predict_rew_end = torch.compile(env.predict_rew_end, mode="reduce-overhead")
            assert not hasattr(env.predict_next_obs, "_torchdynamo_orig_callable") and not hasattr(env.predict_rew_end, "_torchdynamo_orig_callable")
        elif compiled == "spy":
            inner_obs, inner_re = env.predict_next_obs, env.predict_rew_end
            env.predict_next_obs = lambda: (calls.append("obs"), inner_obs())[1]
            env.predict_rew_end = lambda *a, **k: (calls.append("rew_end"), inner_re(*a, **k))[1]
        torch.manual_seed(5)
        env.reset()
        out = []
        for _ in range(2 if os.environ.get("DIAMOND_SLOW_CPU_TESTS") == "1" else 1):  # (horizon 2: a second step ends with a truncation reset)
            obs, rew, end, trunc, _ = env.step(torch.randint(0, 4, (1,)))
            out += [obs, rew, end, trunc]
        return out + [torch.rand(1)], calls  # (+ where the CPU generator stands afterwards)

    plain, _ = rollout(None)
    frames = dict(U.counters["frames"])
    wrapped, _ = rollout("torch.compile")
    assert dict(U.counters["frames"]) == frames, "dynamo traced host code of the env"
    for a, b in zip(plain, wrapped):
        assert torch.equal(a, b)
    if os.environ.get("DIAMOND_SLOW_CPU_TESTS") == "1":  # (plain wrappers are kept and called: also tests/test_boundary.py, test_gpu_env.py)
        spied, calls = rollout("spy")
        assert all(torch.equal(a, c) for a, c in zip(plain, spied)) and calls == ["obs", "rew_end"] * 2, calls  # (two steps in this mode)


def test_trainer_usage_patterns_of_the_denoiser_on_the_interpreter():
    """What an UNCHANGED src/trainer.py does with a world-model component besides `loss.backward()`: `test_component` runs
    `model(batch)` in eval mode under torch.no_grad() (trainer.py:391-399), `train_component` accumulates `grad_acc_steps`
    backward passes before the optimizer step (:363-375), and under DDP the model is `DistributedDataParallel(model)` (:110).  All
    three on a small denoiser against its own plain training step (which tests/test_wide_configs.py pins to the reference): same
    loss, no graph under no_grad; two accumulated backward passes = exactly twice the gradients; DDP at world size 1 (gloo) =
    bitwise the gradients without it."""
    import os
    from types import SimpleNamespace

    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    import diamond_amd as D
    from diamond_amd.inner_model import InnerModelConfig
    from diamond_amd.testing import fill_module_, synthetic_actions, synthetic_frames
    from tests import wide_configs as W

    cfg = dict(W.DENOISER, depths=[1, 1], channels=[64, 96], attn_depths=[0, 1])  # (two levels at 16 x 16: seconds per step)
    den = D.Denoiser(D.DenoiserConfig(inner_model=InnerModelConfig(**cfg), sigma_data=0.5, sigma_offset_noise=0.3))
    fill_module_(den, W.WEIGHT_SEED)
    den.setup_training(D.SigmaDistributionConfig(**W.SIGMA_DIST))
    den.randn_fn = lambda shape: torch.randn(*shape)
    g = torch.Generator().manual_seed(31)
    batch = SimpleNamespace(obs=synthetic_frames(g, 1, 5, 3, 16, 16), act=synthetic_actions(g, 4, 1, 5), mask_padding=torch.ones(1, 5, dtype=torch.bool))

    def step(model, times=1):
        den.zero_grad()
        for _ in range(times):
            torch.manual_seed(77)
            loss, _ = model(batch)
            loss.backward()
        return loss.detach().clone(), {k: p.grad.clone() for k, p in den.named_parameters()}

    with engine_on_interpreter():
        den.train()
        loss, grads = step(den)
        _, twice = step(den, times=2)
        den.eval()
        with torch.no_grad():
            torch.manual_seed(77)
            loss_eval, metrics = den(batch)
        assert not loss_eval.requires_grad and torch.equal(loss_eval, loss) and "loss_denoising" in metrics
        for k in grads:
            assert torch.equal(twice[k], 2 * grads[k]), k
        den.train()
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        try:
            loss_ddp, grads_ddp = step(DDP(den))
        finally:
            dist.destroy_process_group()
        assert torch.equal(loss_ddp, loss)
        for k in grads:
            assert torch.equal(grads_ddp[k], grads[k]), k


def test_uint8_pool_keeps_the_loaders_zero_padded_frames(models, monkeypatch):
    """The reference's segments are ZERO-padded in front of an episode's first step (data/utils.py:18-41; the actor-critic's batch
    sampler allows padding before the start, data/batch_sampler.py:63-68) and its WorldModelEnv uses those frames as they are.
    0.0 is not a uint8 level, so real data used to send the whole initial-condition pool to its fp32 fallback; frames the
    loader marks as padding (mask_padding False) now travel as a stand-in level and come back as exact zeros: the uint8 pool
    (and with it the one-launch reset) serves real data, bitwise the fp32 pool's frames."""
    from types import SimpleNamespace

    import diamond_amd as D
    from diamond_amd.testing import initial_condition_batches
    from diamond_amd.world_model_env import InitialConditionPool

    M, counter = models
    ag = M.make_agent()

    class PaddedLoader:
        batch_sampler = SimpleNamespace(batch_size=3)

        def __iter__(self):
            for obs, act in initial_condition_batches(21, 3, 4):
                mask = torch.ones(3, 4, dtype=torch.bool)
                mask[0, :2] = False  # row 0 starts two steps before its episode, row 2 one step
                mask[2, :1] = False
                obs, act = obs.clone(), act.clone()
                obs[~mask] = 0.0
                act[~mask] = 0
                yield SimpleNamespace(obs=obs, act=act, mask_padding=mask)

    def run(pad_aware):
        monkeypatch.setattr(InitialConditionPool, "PAD_AWARE", pad_aware)
        monkeypatch.setattr(InitialConditionPool, "_warned_fp32", True)  # (the fallback's warning is not the subject)
        env = D.WorldModelEnv(ag.denoiser, ag.rew_end_model, PaddedLoader(),
                              D.WorldModelEnvConfig(horizon=1, num_batches_to_preload=2, diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=1)))
        obs0, _ = env.reset()
        out = [obs0, env.obs_buffer.clone(), env.act_buffer.clone()]
        if os.environ.get("DIAMOND_SLOW_CPU_TESTS") == "1":  # a whole step whose truncations reset every env from the pool
            env.sampler.noise_fn = lambda shape, dev: torch.randn(*shape)
            torch.manual_seed(5)
            obs, rew, end, trunc, info = env.step(torch.tensor([1, 2, 3]))
            out += [obs, rew, end, trunc, info["burnin_obs"], info["final_observation"]]
            # ... and the same step with its deaths resolved on the device (env_loop's default form: dmd_reset_slots dequantises
            # the padded frames itself): the policy's next input, the rings and the reward/end state are those of the step above
            env2 = D.WorldModelEnv(ag.denoiser, ag.rew_end_model, PaddedLoader(),
                                   D.WorldModelEnvConfig(horizon=1, num_batches_to_preload=2, diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=1)))
            env2.reset()
            env2.sampler.noise_fn = lambda shape, dev: torch.randn(*shape)
            torch.manual_seed(5)
            env2.step_begin(torch.tensor([1, 2, 3]))
            ext, rew2, end2, trunc2, slots, _ = env2.step_end_slots()
            env2.slots_finish()
            k = slots.K
            assert k == 3 and torch.equal(slots.slot_row, torch.arange(3))
            burn2 = ext[3 + k:].reshape(3, k, *ext.shape[1:]).transpose(0, 1)
            for a_, b_ in ((ext[:3], obs), (rew2, rew), (end2, end), (trunc2, trunc), (ext[3:3 + k], info["final_observation"]), (burn2, info["burnin_obs"]),
                           (env2.obs_buffer, env.obs_buffer), (env2.act_buffer, env.act_buffer), (env2.hx_rew_end, env.hx_rew_end), (env2.ep_len, env.ep_len)):
                assert torch.equal(a_, b_)
        else:  # the reset itself (what step_end does for dead rows), without the sampler in front of it
            env._head = 2  # (a ring that has advanced: the padded frames have to land in the right slots)
            env._reset_rows(torch.tensor([2, 0]), env.pool.take(2))
        out += [env.obs_buffer.clone(), env.act_buffer.clone(), env.hx_rew_end.clone(), env.cx_rew_end.clone(), env.ep_len.clone(),
                env.pool.gather_frames(torch.tensor([0, 2, 5]))]
        return out, env.pool

    aware, pool = run(True)
    assert pool.frames_u8 is not None and pool.pad is not None and int(pool.pad.sum()) == 2 * 3, "the padded pool did not stay uint8"
    assert float(aware[1][0, :2].abs().max()) == 0 and float(aware[1][0, 2:].abs().min()) > 0  # zeros exactly where the loader padded
    plain, pool32 = run(False)
    assert pool32.frames_u8 is None and pool32.frames_f32 is not None
    for a, b in zip(aware, plain):
        assert torch.equal(a, b)
    assert counter.n.get("dmd_reset_state", 0) >= 1, counter.n  # (the fused reset needs the uint8 pool)
