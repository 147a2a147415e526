"""WorldModelEnv's device data structures on a real MI355X: the uint8 initial-condition pool (quantise /
gather-dequantise kernels) and the ring-indexed context, checked against a roll-based shadow that follows the
reference's bookkeeping (envs/world_model_env.py:64-89) on the same imagined frames."""
from types import SimpleNamespace

import pytest
import torch

from tests.conftest import WEIGHT_SEED

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_quantize_u8_and_dequant_gather_bit_exact():
    from diamond_amd import native as nv
    from diamond_amd.testing import synthetic_frames

    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (6, 4, 3, 16, 24), generator=g, dtype=torch.uint8)
    frames = u8.float().div(255).mul(2).sub(1)  # Episode.load (data/episode.py:36-41)
    fd = frames.to(DEV)
    q = torch.empty(frames.shape, dtype=torch.uint8, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    nv.check(nv.lib().dmd_quantize_u8(nv.fptr(fd), nv.ptr(q), nv.ptr(flag), fd.numel(), nv.stream()), "quantize")
    assert torch.equal(q.cpu(), u8) and int(flag.item()) == 0
    # off-grid detection
    fd2 = fd.clone()
    fd2[3, 2, 1, 5, 7] += 1e-3
    nv.check(nv.lib().dmd_quantize_u8(nv.fptr(fd2), nv.ptr(q), nv.ptr(flag), fd2.numel(), nv.stream()), "quantize")
    assert int(flag.item()) == 1
    # gather rows [4, 1, 5] of the pool into ring rows [2, 0, 3] with head 3: logical frame t -> slot (3 + t) % 4
    nv.check(nv.lib().dmd_quantize_u8(nv.fptr(fd), nv.ptr(q), nv.ptr(flag), fd.numel(), nv.stream()), "quantize")
    ring = torch.full((5, 4, 3, 16, 24), 7.0, device=DEV)
    idx = torch.tensor([4, 1, 5], device=DEV)
    rows = torch.tensor([2, 0, 3], device=DEV)
    nv.check(nv.lib().dmd_dequant_gather(nv.ptr(q), nv.ptr(idx), nv.ptr(rows), nv.fptr(ring), 3, 4, 3 * 16 * 24, 3, nv.stream()),
             "gather")
    ring = ring.cpu()
    for i, r in zip((4, 1, 5), (2, 0, 3)):
        for t in range(4):
            assert torch.equal(ring[r, (3 + t) % 4], frames[i, t])
    assert float(ring[1].min()) == 7.0 and float(ring[4].min()) == 7.0  # untouched rows


class _Loader:
    def __init__(self, b, batches):
        self.batch_sampler = SimpleNamespace(batch_size=b)
        self._batches = batches

    def __iter__(self):
        i = 0
        while True:
            obs, act = self._batches[i % len(self._batches)]
            i += 1
            yield SimpleNamespace(obs=obs, act=act)


@pytest.mark.parametrize("on_grid", [True, False])
def test_ring_context_matches_roll_based_shadow(on_grid):
    """Drive the env for 9 steps (horizon 3 -> every env resets, at different times once `end` fires) and keep a
    shadow that rolls / overwrites like the reference does, fed with the env's own imagined frames: contexts,
    action buffers, returned observations, final observations and burn-in frames must be identical."""
    import diamond_amd as D
    from diamond_amd.testing import fill_module_, synthetic_actions, synthetic_frames

    agent = D.Agent(D.default_agent_config())
    fill_module_(agent, WEIGHT_SEED)
    agent = agent.to(DEV).eval()
    b = 3
    g = torch.Generator().manual_seed(5)
    batches = []
    for _ in range(4):
        obs = synthetic_frames(g, b, 4, 3, 64, 64)
        if not on_grid:
            obs = (obs + (torch.rand(obs.shape, generator=g) - 0.5) * 1e-3).clamp(-1, 1)
        batches.append((obs, synthetic_actions(g, 4, b, 4)))
    env = D.WorldModelEnv(agent.denoiser, agent.rew_end_model, _Loader(b, batches),
                          D.WorldModelEnvConfig(horizon=3, num_batches_to_preload=2,
                                                diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=1)))
    obs0, _ = env.reset()
    assert (env.pool.frames_u8 is not None) == on_grid
    pool_obs = torch.cat([batches[0][0], batches[1][0]]).to(DEV)
    pool_act = torch.cat([batches[0][1], batches[1][1]]).to(DEV)
    cursor = b
    sh_obs, sh_act = pool_obs[:b].clone(), pool_act[:b].clone()
    assert torch.equal(obs0, sh_obs[:, -1]) and torch.equal(env.obs_buffer, sh_obs) and torch.equal(env.act_buffer, sh_act)
    captured = {}
    inner = env.predict_next_obs

    def spy():
        out = inner()
        captured["next_obs"] = out[0]
        return out

    env.predict_next_obs = spy  # also checks that the callable is re-assignable (trainer.py:182-184)
    pools_seen = 1
    for step in range(9):
        act = torch.randint(0, 4, (b,), generator=g).to(DEV)
        obs, rew, end, trunc, info = env.step(act)
        nxt = captured["next_obs"]
        # ---- shadow: the reference's bookkeeping
        sh_act[:, -1] = act
        sh_obs, sh_act = sh_obs.roll(-1, dims=1), sh_act.roll(-1, dims=1)
        sh_obs[:, -1] = nxt
        dead = torch.logical_or(end, trunc)
        if dead.any():
            nd = int(dead.sum())
            if cursor + nd > pool_obs.shape[0]:  # pool exhausted: the next preload round replaces it
                pool_obs = torch.cat([batches[(2 * pools_seen) % 4][0], batches[(2 * pools_seen + 1) % 4][0]]).to(DEV)
                pool_act = torch.cat([batches[(2 * pools_seen) % 4][1], batches[(2 * pools_seen + 1) % 4][1]]).to(DEV)
                pools_seen += 1
                cursor = 0
            sh_obs[dead] = pool_obs[cursor:cursor + nd]
            sh_act[dead] = pool_act[cursor:cursor + nd]
            cursor += nd
            assert torch.equal(info["final_observation"], nxt[dead])
            assert torch.equal(info["burnin_obs"], sh_obs[dead, :-1])
        else:
            assert "final_observation" not in info
        assert torch.equal(obs, sh_obs[:, -1]), f"step {step}: returned observation"
        assert torch.equal(env.obs_buffer, sh_obs), f"step {step}: context ring != rolled shadow"
        # the newest action slot is written by the NEXT step (like the reference's act_buffer[:, -1])
        assert torch.equal(env.act_buffer[:, :-1], sh_act[:, :-1]), f"step {step}: action ring"
    assert pools_seen >= 1


def test_graph_captured_sampler_equals_eager_bitwise():
    """Latency mode (play.py's B=1 env): the sampler replayed as a hipGraph gives bit-identical frames to the eager
    launch sequence, reads the CURRENT contents of the context rings at replay time, and one graph exists per ring
    head."""
    import diamond_amd as D
    from diamond_amd.testing import fill_module_, synthetic_actions, synthetic_frames

    agent = D.Agent(D.default_agent_config())
    fill_module_(agent, WEIGHT_SEED)
    agent = agent.to(DEV).eval()
    g = torch.Generator().manual_seed(9)
    sampler = D.DiffusionSampler(agent.denoiser, D.DiffusionSamplerConfig(num_steps_denoising=3))
    ctx = synthetic_frames(g, 1, 4, 3, 64, 64).to(DEV)
    act = synthetic_actions(g, 4, 1, 4).to(DEV)
    noise = torch.randn(1, 3, 64, 64, generator=g).to(DEV)
    sampler.noise_fn = lambda shape, dev: noise  # a device tensor: nothing is copied during capture
    for head in (0, 3):
        xe, te = sampler.sample_ring(ctx, act, head, head)
        xg, tg = sampler.sample_ring_graphed(ctx, act, head, head)
        assert torch.equal(xe, xg) and all(torch.equal(a, b) for a, b in zip(te, tg))
        # new context in the same buffers: the replay must see it
        ctx.copy_(synthetic_frames(g, 1, 4, 3, 64, 64).to(DEV))
        act.copy_(synthetic_actions(g, 4, 1, 4).to(DEV))
        noise.copy_(torch.randn(1, 3, 64, 64, generator=g).to(DEV))
        xe2, _ = sampler.sample_ring(ctx, act, head, head)
        xg2, _ = sampler.sample_ring_graphed(ctx, act, head, head)
        assert torch.equal(xe2, xg2) and not torch.equal(xe2, xe)
    assert len(sampler._graphs) == 2
    # without the hook the noise comes from torch's generator inside the graph: a fresh draw per replay
    sampler.noise_fn = None
    a, _ = sampler.sample_ring_graphed(ctx, act, 1, 1)
    b, _ = sampler.sample_ring_graphed(ctx, act, 1, 1)
    assert torch.isfinite(a).all() and not torch.equal(a, b)


def test_captured_sampler_graphs_survive_a_weight_update():
    """An optimizer step / load changes every parameter: the packed copies are rebuilt IN PLACE, so the captured sampler graphs
    stay valid (same graph objects, no recapture) and the next replay computes with the NEW weights -- also the copies that are
    the parameter's own storage (cache.f32 of a contiguous fp32 parameter: a version change there is not a freed buffer)."""
    import diamond_amd as D
    from diamond_amd.testing import fill_module_, synthetic_actions, synthetic_frames

    agent = D.Agent(D.default_agent_config())
    fill_module_(agent, WEIGHT_SEED)
    agent = agent.to(DEV).eval()
    g = torch.Generator().manual_seed(10)
    sampler = D.DiffusionSampler(agent.denoiser, D.DiffusionSamplerConfig(num_steps_denoising=3))
    ctx = synthetic_frames(g, 1, 4, 3, 64, 64).to(DEV)
    act = synthetic_actions(g, 4, 1, 4).to(DEV)
    noise = torch.randn(1, 3, 64, 64, generator=g).to(DEV)
    sampler.noise_fn = lambda shape, dev: noise
    x0, _ = sampler.sample_ring_graphed(ctx, act, 0, 0)
    x0b, _ = sampler.sample_ring_graphed(ctx, act, 0, 0)
    caps = dict(sampler._graphs)
    assert len(caps) == 1 and torch.equal(x0, x0b)
    with torch.no_grad():
        for p in agent.denoiser.parameters():
            p.mul_(1.01)  # (an in-place update like an eager optimizer step: bumps every version)
    x1, _ = sampler.sample_ring_graphed(ctx, act, 0, 0)
    assert dict(sampler._graphs) == caps, "the weight update voided the captured graph"
    xe, _ = sampler.sample_ring(ctx, act, 0, 0)
    assert torch.equal(x1, xe) and not torch.equal(x1, x0)


def test_batch_shard_invariance_at_full_batch():
    """§8e: imagined envs shard along the batch axis with no data-path exchange, so an env's trajectory must not depend
    on which other envs share its launch.  The 256 envs of configs[1] stepped as ONE batch vs as two shards of 128 with
    the same per-env initial conditions, sampler noise, exponential draws and policy: frames, rewards, ends, policy
    logits, values and sampled actions are BITWISE identical (convolutions: launch-configuration independent tiles;
    linears: the summation order depends on K only; pointwise kernels: per element)."""
    import diamond_amd as D
    from diamond_amd.env_loop import sample_categorical
    from diamond_amd.testing import fill_module_, synthetic_actions, synthetic_frames

    agent = D.Agent(D.default_agent_config())
    fill_module_(agent, WEIGHT_SEED)
    agent = agent.to(DEV).eval()
    n, steps = 256, 3
    g = torch.Generator().manual_seed(21)
    obs0 = synthetic_frames(g, n, 4, 3, 64, 64)
    act0 = synthetic_actions(g, 4, n, 4)

    def bank(tag, k, shape_tail, lo, hi, exponential=False):
        gg = torch.Generator().manual_seed(1000 * tag + k)
        t = torch.empty(n, *shape_tail)
        t = t.exponential_(1, generator=gg) if exponential else t.normal_(generator=gg)
        return t[lo:hi].to(DEV)

    def run(lo, hi):
        b = hi - lo
        env = D.WorldModelEnv(agent.denoiser, agent.rew_end_model, _Loader(b, [(obs0[lo:hi], act0[lo:hi])]),
                              D.WorldModelEnvConfig(horizon=50, num_batches_to_preload=1,
                                                    diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=3)))
        calls = {"noise": 0, "expo": 0, "pi": 0}

        def noise_fn(shape, device):
            calls["noise"] += 1
            return bank(1, calls["noise"], shape[1:], lo, hi)

        def expo_fn(logits):
            calls["expo"] += 1
            e = bank(2, calls["expo"], logits.shape[1:], lo, hi, exponential=True)
            if logits.shape[-1] == 2:  # termination head: class 0 always wins -> no mid-window resets (pool order)
                e[..., 0] = 1e-30
            return e

        env.sampler.noise_fn = noise_fn
        env.expo_fn = expo_fn
        obs, _ = env.reset()
        hx = torch.zeros(b, agent.actor_critic.lstm_dim, device=DEV)
        cx = torch.zeros(b, agent.actor_critic.lstm_dim, device=DEV)
        out = []
        for _ in range(steps):
            with torch.no_grad():
                o = agent.actor_critic.predict_act_value(obs, (hx, cx))
            hx, cx = o.hx_cx
            calls["pi"] += 1
            act = sample_categorical(o.logits_act, bank(3, calls["pi"], o.logits_act.shape[1:], lo, hi, exponential=True))
            obs, rew, end, trunc, _ = env.step(act)
            assert not bool(end.any()) and not bool(trunc.any())
            out.append((o.logits_act.clone(), o.val.clone(), act.clone(), obs.clone(), rew.clone(), end.clone()))
        return out

    full = run(0, n)
    for lo, hi in ((0, 128), (128, 256)):
        part = run(lo, hi)
        for s, (f, p) in enumerate(zip(full, part)):
            for name, a, c in zip(("logits", "value", "action", "frame", "reward", "end"), f, p):
                assert torch.equal(a[lo:hi], c), f"step {s}: {name} of envs [{lo}, {hi}) depends on the batch it ran in"
