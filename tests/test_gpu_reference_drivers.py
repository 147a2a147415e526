"""Drop-in, proven with the REFERENCE'S OWN DRIVER CODE on the GPU: its `coroutines/env_loop.py:12-74` generator, the attribute
re-assignment of `trainer.py:182-184`, and the loop body of `trainer.py:363-382` run over `diamond_amd.WorldModelEnv` /
`diamond_amd.ActorCritic` -- not mirrors of them.  The reference travels to the GPU box as bytecode (oracle/_ref, built by
oracle/make_ref.py in the build container; TEST INFRASTRUCTURE, the product never imports it); where neither the bytecode nor
/root/reference exists the tests skip."""
import json
import os
import random
import subprocess
import sys

import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


def _reference():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import reference_window as RW
    finally:
        sys.path.pop(0)
    path, what = RW.reference_location()
    if path is None:
        pytest.skip(f"no importable reference: {what}")
    RW._install(path)
    return what


def _window_setup(gold, size=64):
    import diamond_amd as D
    import tests.test_gpu_models as M
    from tests.test_gpu_models import _Loader, make_agent

    M.DEV = DEV  # (DIAMOND_TESTS_ON_INTERPRETER=1 re-targets the collected modules only)
    ag = make_agent(img_size=size)
    env = D.WorldModelEnv(ag.denoiser, ag.rew_end_model, _Loader(gold["b"], gold["pool_seed"], size),
                          D.WorldModelEnvConfig(horizon=gold["horizon"], num_batches_to_preload=gold["preload"],
                                                diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=3)))
    expo = lambda logits: torch.empty(logits.shape, dtype=torch.float32).exponential_(1)
    env.expo_fn = expo
    env.sampler.noise_fn = lambda shape, dev: torch.randn(*shape).to(dev)
    return ag, env


def _cpu_stream_categorical(monkeypatch):
    """The reference samples actions with Categorical(logits).sample() on the tensors' device; the golden was produced on the CPU
    default generator.  Same distribution object, same call: only the exponential draws behind torch.multinomial are made on the
    CPU generator (the stream the golden consumed: argmax(probs / E))."""
    from torch.distributions.categorical import Categorical

    import diamond_amd.env_loop as EL

    def sample(self, sample_shape=torch.Size()):
        assert sample_shape == torch.Size()
        e = torch.empty(self.logits.shape, dtype=torch.float32).exponential_(1)
        return EL.sample_categorical(self.logits, e)

    monkeypatch.setattr(Categorical, "sample", sample)


def test_reference_env_loop_drives_our_env_and_actor_critic_bit_exact_vs_golden(monkeypatch):
    """the reference's make_env_loop (its control flow, its boolean-mask resets, its in-place burn-in) over diamond_amd's
    WorldModelEnv.step / ActorCritic.predict_act_value: two windows with truncations and sampled ends against window.pt"""
    _reference()
    from coroutines.env_loop import make_env_loop  # the reference's

    gold = load_golden("window.pt")
    ag, env = _window_setup(gold)
    _cpu_stream_categorical(monkeypatch)
    torch.manual_seed(gold["rng_seed"])
    random.seed(0)
    loop = make_env_loop(env, ag.actor_critic)
    from tests.test_gpu_models import check_quantised, rel_err, u8

    for w in gold["windows"]:
        all_obs, act, rew, end, trunc, logits_act, val, vb, infos = loop.send(gold["backup_every"])
        assert torch.equal(act.cpu(), w["act"]) and torch.equal(rew.cpu(), w["rew"])
        assert torch.equal(end.cpu(), w["end"]) and torch.equal(trunc.cpu(), w["trunc"])
        check_quantised(u8(all_obs), w["obs_u8"], max_frac=2e-3)
        assert rel_err(logits_act.detach(), w["logits_act"]) < 1e-2 and rel_err(val.detach(), w["val"]) < 1e-2
        (logits_act.square().mean() + val.mean()).backward()  # the reference's in-place burn-in graph differentiates on our Functions
    assert int(sum(w["end"].sum() + w["trunc"].sum() for w in gold["windows"])) > 0


def test_reference_style_reassignment_of_predict_next_obs_and_predict_rew_end(monkeypatch):
    """trainer.py:182-184 replaces rl_env.predict_next_obs / predict_rew_end by wrappers (torch.compile objects) AFTER the env was
    built: the env -- including the slots loop's step_begin / step_end_slots -- must call through the attributes"""
    import diamond_amd as D
    from diamond_amd.actor_critic import actor_critic_loss

    gold = load_golden("window.pt")
    ag, env = _window_setup(gold)
    calls = {"next_obs": 0, "rew_end": 0}

    def wrap(fn, key):  # what torch.compile returns: a callable that forwards *args / **kwargs
        def inner(*a, **k):
            calls[key] += 1
            return fn(*a, **k)
        return inner

    env.predict_next_obs = wrap(env.predict_next_obs, "next_obs")
    env.predict_rew_end = wrap(env.predict_rew_end, "rew_end")
    t = gold["backup_every"]
    ag.setup_training(D.SigmaDistributionConfig(-0.4, 1.2, 2e-3, 20),
                      D.ActorCriticLossConfig(backup_every=t, gamma=0.985, lambda_=0.95, weight_value_loss=1.0, weight_entropy_loss=0.001), env)
    ag.actor_critic.expo_fn = env.expo_fn
    torch.manual_seed(gold["rng_seed"])
    random.seed(0)
    for w in gold["windows"]:
        _, act, rew, end, trunc, *_ = ag.actor_critic.env_loop.send(t)
        assert torch.equal(act.cpu(), w["act"]) and torch.equal(end.cpu(), w["end"]) and torch.equal(rew.cpu(), w["rew"])
    steps = len(gold["windows"]) * t
    assert calls["next_obs"] == steps and calls["rew_end"] == steps, calls


def test_reference_trainer_loop_body_under_ddp_world1():
    """trainer.py:363-382 (`loss, metrics = model(); loss.backward(); clip_grad_norm_; opt.step(); opt.zero_grad()`) for two steps
    with model = DistributedDataParallel(agent.actor_critic) exactly as utils.py:105-106 wraps it, the optimizer from the
    reference's own configure_opt (bytecode) where it travelled -- on RCCL at world size 1, in a worker process"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "tests", "dist_gpu_worker.py"), "--trainer-body"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    print(out)
    assert out["steps"] == 2 and out["losses_finite"] and out["params_moved"] and out["grads_zeroed"]
    assert out["grad_norms"][0] > 0 and out["metrics_keys"] >= 5
