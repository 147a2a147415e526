"""Drop-in boundary conformance (CPU): state-dict layout, optimizer grouping, Agent.load,
sampler schedule, lambda-returns, and that the C-ABI library loads and exports every symbol
declared in include/diamond_hip.h.  No kernel is launched here."""
import ctypes
import os
import re

import pytest
import torch
from torch import nn

import diamond_amd as D
from tests.conftest import ROOT, load_golden


@pytest.fixture(scope="module")
def agent():
    return D.Agent(D.default_agent_config())


def test_state_dict_matches_reference_tree(agent):
    ref = load_golden("state_dict_keys.pt")  # dumped from the instantiated reference Agent
    mine = {k: tuple(v.shape) for k, v in agent.state_dict().items()}
    assert list(mine.keys()) == list(ref.keys())
    assert mine == ref
    assert sum(p.numel() for p in agent.denoiser.parameters()) == 4405955
    assert sum(p.numel() for p in agent.rew_end_model.parameters()) == 5900864
    assert sum(p.numel() for p in agent.actor_critic.parameters()) == 3229637


def test_every_parameter_is_optimizer_groupable(agent):
    """configure_opt (reference utils.py:129-166) asserts each parameter belongs to a
    whitelisted / blacklisted module type or is a bias."""
    white = (nn.Linear, nn.Conv1d, nn.Conv2d, nn.LSTMCell, nn.LSTM)
    black = (nn.LayerNorm, nn.Embedding, nn.GroupNorm)
    for model in (agent.denoiser, agent.rew_end_model, agent.actor_critic):
        decay, no_decay = set(), set()
        for mn, m in model.named_modules():
            for pn, _ in m.named_parameters():
                fpn = f"{mn}.{pn}" if mn else pn
                if "bias" in pn:
                    no_decay.add(fpn)
                elif (pn.endswith("weight") or pn.startswith("weight_")) and isinstance(m, white):
                    decay.add(fpn)
                elif (pn.endswith("weight") or pn.startswith("weight_")) and isinstance(m, black):
                    no_decay.add(fpn)
        names = {n for n, _ in model.named_parameters()}
        assert not (decay & no_decay)
        assert names == (decay | no_decay), names - (decay | no_decay)


def test_agent_load_roundtrip(tmp_path, agent):
    from diamond_amd.testing import fill_module_

    other = D.Agent(D.default_agent_config())
    fill_module_(other, 3)
    path = tmp_path / "agent.pt"
    torch.save(other.state_dict(), path)
    agent.load(path, load_rew_end_model=False)
    assert torch.equal(agent.denoiser.inner_model.conv_in.weight, other.denoiser.inner_model.conv_in.weight)
    assert torch.equal(agent.actor_critic.lstm.weight_hh, other.actor_critic.lstm.weight_hh)
    assert not torch.equal(agent.rew_end_model.head[0].weight, other.rew_end_model.head[0].weight)


def test_default_init_zeroes_like_the_reference():
    agent = D.Agent(D.default_agent_config())
    im = agent.denoiser.inner_model
    assert float(im.conv_out.weight.abs().max()) == 0
    assert float(im.unet.d_blocks[0].resblocks[0].conv2.weight.abs().max()) == 0
    assert float(im.unet.mid_blocks.resblocks[0].attn.out_proj.weight.abs().max()) == 0
    assert float(agent.actor_critic.actor_linear.weight.abs().max()) == 0
    b = agent.actor_critic.lstm.bias_ih
    assert float(b[512:1024].min()) == 1 and float(b[:512].abs().max()) == 0


def test_sigma_schedule_bit_exact_with_reference_values():
    s = D.build_sigmas(3, 2e-3, 5.0, 7, torch.device("cpu"))
    assert torch.equal(s, load_golden("denoiser_default.pt")["sigmas"])
    assert s.shape == (4,) and float(s[-1]) == 0.0


def test_lambda_returns_matches_oracle():
    from oracle import diamond_oracle as O

    g = torch.Generator().manual_seed(0)
    rew = torch.randint(-1, 2, (5, 7), generator=g).float() * 2.5
    end = (torch.rand(5, 7, generator=g) < 0.15).long()
    trunc = (torch.rand(5, 7, generator=g) < 0.1).long()
    vb = torch.randn(5, 7, generator=g)
    for lam in (0.0, 0.95):
        assert torch.equal(D.compute_lambda_returns(rew, end, trunc, vb, 0.985, lam),
                           O.lambda_returns(rew, end, trunc, vb, 0.985, lam))


def test_product_path_has_no_cpu_fallback(agent):
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(Exception):
        agent.denoiser.denoise(x, 1.0, torch.zeros(1, 12, 64, 64), torch.zeros(1, 4, dtype=torch.long))


def test_c_abi_library_exports_every_declared_symbol():
    from diamond_amd import native

    assert os.path.exists(native.LIB_PATH), "libdiamond_hip.so missing: run __graft_entry__.build()"
    header = open(os.path.join(ROOT, "include", "diamond_hip.h")).read()
    declared = set(re.findall(r"\b(dmd_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 18
    lib = ctypes.CDLL(native.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/diamond_hip.h but not exported"
    assert set(native.EXPORTS) == declared
    assert lib.dmd_abi_version() == 11
    assert lib.dmd_conv_stat_tiles(64, 64) == 32 and lib.dmd_conv_stat_tiles(8, 8) == 1


def test_struct_layouts_match_the_header():
    """sizeof of the ctypes mirrors -- and the offsets of the fields most likely to drift -- must equal the C structs
    (compiled with gcc from include/diamond_hip.h)."""
    import subprocess
    import tempfile
    from diamond_amd import native

    structs = [("dmd_norm", native.Norm), ("dmd_conv_src", native.ConvSrc), ("dmd_conv_params", native.ConvParams),
               ("dmd_linear_params", native.LinearParams), ("dmd_gn_bwd_params", native.GnBwdParams),
               ("dmd_wgrad_params", native.WgradParams), ("dmd_wgrad_reduce_job", native.WgradReduceJob), ("dmd_chain_block", native.ChainBlock),
               ("dmd_lowres_chain_params", native.LowresChainParams), ("dmd_reset_slots_params", native.ResetSlotsParams),
               ("dmd_pool_round", native.PoolRound)]
    offsets = [("dmd_conv_params", native.ConvParams, "w_f16"), ("dmd_conv_params", native.ConvParams, "precision"),
               ("dmd_wgrad_params", native.WgradParams, "precision"), ("dmd_wgrad_params", native.WgradParams, "defer_reduce"),
               ("dmd_wgrad_reduce_job", native.WgradReduceJob, "ld_cin"), ("dmd_chain_block", native.ChainBlock, "w1"),
               ("dmd_chain_block", native.ChainBlock, "bo"), ("dmd_lowres_chain_params", native.LowresChainParams, "table_stride"),
               ("dmd_lowres_chain_params", native.LowresChainParams, "blocks"), ("dmd_reset_slots_params", native.ResetSlotsParams, "pool_base"), ("dmd_reset_slots_params", native.ResetSlotsParams, "num_dead"),
               ("dmd_reset_slots_params", native.ResetSlotsParams, "enc_in")]
    body = "".join(f'printf("%zu ", sizeof({c}));' for c, _ in structs)
    body += "".join(f'printf("%zu ", offsetof({c}, {f}));' for c, _, f in offsets)
    src = f'#include <stdio.h>\n#include <stddef.h>\n#include "diamond_hip.h"\nint main(){{{body}return 0;}}'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", os.path.join(d, "s")])
        got = [int(v) for v in subprocess.check_output([os.path.join(d, "s")]).split()]
    mine = [ctypes.sizeof(t) for _, t in structs] + [getattr(t, f).offset for _, t, f in offsets]
    assert got == mine, list(zip([c for c, _ in structs] + [f"{c}.{f}" for c, _, f in offsets], got, mine))
    assert native.CHAIN_MAX_BLOCKS == 8


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree only exists in the build container")
def test_reference_configure_opt_and_agent_load_accept_our_modules(tmp_path):
    """The reference's OWN `utils.configure_opt` (utils.py:129-166) groups every parameter of our three models, and the
    reference's own `Agent.load` reads a checkpoint written from our state dict (and vice versa) -- run in a child
    process so that the reference's top-level module names (`agent`, `utils`, `models`, ...) stay out of this one."""
    import subprocess
    import sys

    script = f"""
import sys, torch
sys.dont_write_bytecode = True
sys.path.insert(0, {os.path.join(ROOT, 'tests', 'golden')!r}); sys.path.insert(0, {ROOT!r})
import _refimport as R
R.install()
import diamond_amd as D
from diamond_amd.testing import fill_module_
from utils import configure_opt            # the reference's
from agent import Agent as RefAgent        # the reference's
mine = D.Agent(D.default_agent_config())
fill_module_(mine, 3)
for m in (mine.denoiser, mine.rew_end_model, mine.actor_critic):
    opt = configure_opt(m, lr=1e-4, weight_decay=1e-2, eps=1e-8)
    n = sum(p.numel() for g in opt.param_groups for p in g['params'])
    assert n == sum(p.numel() for p in m.parameters())
path = {str(tmp_path / 'ckpt.pt')!r}
torch.save(mine.state_dict(), path)
ref = RefAgent(R.default_agent_config(num_actions=4))
ref.load(path)                             # reference Agent.load on OUR checkpoint
assert all(torch.equal(a, b) for a, b in zip(ref.state_dict().values(), mine.state_dict().values()))
fill_module_(ref, 9)
torch.save(ref.state_dict(), path)
mine.load(path)                            # our Agent.load on the REFERENCE's checkpoint
assert all(torch.equal(a, b) for a, b in zip(ref.state_dict().values(), mine.state_dict().values()))
print('ok')
"""
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-3000:]


def test_ring_env_bookkeeping_matches_roll_based_shadow_cpu():
    """Host logic of the new WorldModelEnv on CPU tensors (no kernel is launched: the two re-assignable callables are
    stubbed, the pool is pre-filled as fp32): ring advance, resets from the pool (serving order + the reference's
    'drop the remainder and reload' rule, world_model_env.py:133-139), returned observations, final_observation and
    burnin_obs against a shadow that rolls its buffers like the reference (world_model_env.py:64-89)."""
    from types import SimpleNamespace

    b, t, c, h, w = 3, 4, 1, 2, 2
    g = torch.Generator().manual_seed(0)
    rounds = [(torch.randn(2 * b, t, c, h, w, generator=g), torch.randint(0, 4, (2 * b, t), generator=g)) for _ in range(6)]
    loader = SimpleNamespace(batch_sampler=SimpleNamespace(batch_size=b))
    fake_den = SimpleNamespace(device=torch.device("cpu"))
    env = D.WorldModelEnv.__new__(D.WorldModelEnv)
    env.sampler = SimpleNamespace(denoiser=fake_den, noise_fn=None, cfg=SimpleNamespace(s_churn=0.0),
                                  _randn=lambda shape, dev: torch.zeros(*shape))
    env.rew_end_model, env.horizon, env.return_denoising_trajectory, env.num_envs = None, 3, False, b
    env.graph_sampler, env.expo_fn, env._graph_forced = False, None, False
    env._ctx = env._act = None
    env._head = 0
    from diamond_amd.world_model_env import InitialConditionPool

    pool = InitialConditionPool(None, loader, 2, lambda: torch.device("cpu"))
    served = {"n": 0}

    def preload():  # stands in for the kernel-backed preload: one round = 2 batches of b rows, kept as an fp32 pool
        obs, act = rounds[served["n"]]
        served["n"] += 1
        pool.frames_u8, pool.frames_f32, pool.act = None, obs, act
        pool.hx, pool.cx = torch.zeros(2 * b, 8), torch.zeros(2 * b, 8)
        pool._cursor = 0

    pool._preload = preload
    env.pool = pool
    obs0, _ = env.reset()
    sh_obs, sh_act = rounds[0][0][:b].clone(), rounds[0][1][:b].clone()
    cursor, rnd = b, 0
    assert torch.equal(obs0, sh_obs[:, -1]) and torch.equal(env.obs_buffer, sh_obs)
    state = {}
    env.predict_next_obs = lambda: (state["nxt"], [])
    env.predict_rew_end = lambda next_obs, e_rew=None, e_end=None: (torch.zeros(b), state["end"])
    for step in range(14):
        act = torch.randint(0, 4, (b,), generator=g)
        state["nxt"] = torch.randn(b, c, h, w, generator=g)
        state["end"] = (torch.rand(b, generator=g) < 0.3).long()
        obs, rew, end, trunc, info = env.step(act)
        sh_act[:, -1] = act
        sh_obs, sh_act = sh_obs.roll(-1, dims=1), sh_act.roll(-1, dims=1)
        sh_obs[:, -1] = state["nxt"]
        dead = torch.logical_or(end, trunc)
        if dead.any():
            nd = int(dead.sum())
            if cursor + nd > 2 * b:
                rnd, cursor = rnd + 1, 0
            sh_obs[dead] = rounds[rnd][0][cursor:cursor + nd]
            sh_act[dead] = rounds[rnd][1][cursor:cursor + nd]
            cursor += nd
            assert torch.equal(info["final_observation"], state["nxt"][dead])
            assert torch.equal(info["burnin_obs"], sh_obs[dead, :-1])
        assert torch.equal(obs, sh_obs[:, -1]), step
        assert torch.equal(env.obs_buffer, sh_obs), step
        assert torch.equal(env.act_buffer[:, :-1], sh_act[:, :-1]), step
        assert obs.data_ptr() != env._ctx.data_ptr()  # returned observations never alias the ring
    assert served["n"] == rnd + 1 and rnd >= 1


def test_confusion_matrix_matches_sklearn():
    """RewEndModel.forward's metrics (reference rew_end_model.py:84-85 via torcheval, absent here): rows = true class,
    columns = argmax prediction, checked against scikit-learn's definition."""
    import numpy as np
    from sklearn.metrics import confusion_matrix as sk_cm
    from diamond_amd.rew_end_model import confusion_matrix

    g = torch.Generator().manual_seed(5)
    logits = torch.randn(40, 3, generator=g)
    target = torch.randint(0, 3, (40,), generator=g)
    cm = confusion_matrix(logits, target, 3)
    assert cm.dtype == torch.int64 and cm.shape == (3, 3)
    np.testing.assert_array_equal(cm.numpy(), sk_cm(target.numpy(), logits.argmax(1).numpy(), labels=[0, 1, 2]))


def test_no_kernel_uses_scratch_and_asm_loads_are_clean():
    """Compile-time hygiene of the HIP sources (cross-compiles without a GPU): (1) no instantiated kernel spills to scratch
    memory (tools/resource_usage.py, hipcc -Rpass-analysis=kernel-resource-usage); (2) in conv_f16ws_kernel nothing reads,
    copies or overwrites a destination register of a hand-counted inline-asm global load before its wait names it
    (tools/asm_lint.py: a forward may-analysis over the emitted .s -- hipcc does not know those registers are pending)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "resource_usage.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("scratch    0 B/lane") >= 90, r.stdout[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "asm_lint.py")], capture_output=True, text=True)
    assert r.returncode == 0 and "asm_lint: OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_actor_critic_encoder_takes_its_own_size_unpadded():
    """the default encoder's convolutions run at 64 / 32 / 16 / 8 (the 4 x 4 after the last pool is only flattened): 64 x 64 is on
    the kernels' grid and must NOT go through a padded buffer (a 128 x 128 buffer is 4x the work -- measured once: 12.3k -> 10.8k
    frames/s); 72 x 72 (72 / 36 / 18 / 9) is not"""
    import diamond_amd as D
    from diamond_amd.ac_native import _Plan

    ac = D.Agent(D.default_agent_config()).actor_critic
    plan = _Plan(ac.encoder.encoder)
    assert plan.grid_multiple == 64 and 64 % plan.grid_multiple == 0 and 72 % plan.grid_multiple != 0


def test_a_module_on_another_gpu_than_the_current_one_is_refused(monkeypatch):
    """ctypes launches go to the CURRENT device's stream with raw pointers: an agent on cuda:1 while device 0 is current would
    launch on the wrong GPU (torch ops switch devices by themselves).  The coarse entry points check it and say what to do."""
    from diamond_amd import native as nv

    nv.check_current_device(torch.device("cpu"))
    nv.check_current_device(torch.device("cuda"))  # (no index: nothing to compare)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    nv.check_current_device(torch.device("cuda", 0))
    with pytest.raises(RuntimeError, match=r"set_device\(1\)"):
        nv.check_current_device(torch.device("cuda", 1))
    # every public entry point that launches through ctypes has the check in front of its first launch (round 5 guarded three)
    import inspect

    import diamond_amd as D
    from diamond_amd.rew_end_model import RewEndModel

    for fn in (D.ActorCritic.encode, D.ActorCritic.predict_from_features, D.DiffusionSampler.sample_ring, D.DiffusionSampler.sample_ring_graphed,
               D.Denoiser.compute_model_output, D.Denoiser.forward, RewEndModel.predict_rew_end, RewEndModel.forward, D.WorldModelEnv.reset):
        assert "check_current_device" in inspect.getsource(fn), fn.__qualname__
