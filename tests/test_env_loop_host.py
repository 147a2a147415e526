"""Host logic of env_loop on the CPU (no kernel: torch toy policies, toy envs and a torch stand-in for the categorical-sample
kernel).  With ONE shared random stream for every draw (policy exponentials, env noise, reward / end exponentials) a rollout
must be bitwise the sequential one -- i.e. the stream is consumed in the same order.  The sequential loop's batched burn-in
(one encoder pass over all burn-in frames of a reset) against frame-by-frame calls; the slots loop (a step's deaths resolved
into fixed-shape reset slots, the host one step behind, windows repeated after a slot overflow) against the sequential loop.
(The GPU twins with the real models: tests/test_gpu_models.py.)"""
import random

import pytest
import torch

import diamond_amd.env_loop as EL


class ToyPolicy:
    lstm_dim = 5

    def __init__(self):
        g = torch.Generator().manual_seed(3)
        self.w = torch.randn(4, 6, generator=g)
        self.device = torch.device("cpu")
        self.calls = 0

    def predict_act_value(self, obs, hx_cx):
        self.calls += 1
        hx, cx = hx_cx
        feat = obs.flatten(1)[:, :6] + hx[:, :1]
        logits = (feat[:, None, :] * self.w[None]).sum(-1)  # (per-row reduction: a GEMM's blocking depends on the batch)
        val = feat.sum(1)
        return logits, val, (torch.tanh(hx + feat[:, :5]), cx + 1)


class ToyEnv:
    """the reference's env protocol: `step`, deaths reset the row, info carries final_observation / any_dead"""
    num_actions = 4

    def __init__(self, b, p_end):
        self.num_envs, self.p_end = b, p_end
        self.log = []

    def reset(self, **kw):
        self.state = torch.randn(self.num_envs, 2, 3)
        self.t = torch.zeros(self.num_envs, dtype=torch.long)
        return self.state.clone(), {}

    def step(self, act):
        self.log.append("step")
        nxt = self.state * 0.5 + act.float().view(-1, 1, 1) * 0.1 + torch.randn(self.num_envs, 2, 3)  # "denoiser noise"
        e_rew, e_end = torch.empty(self.num_envs, 3).exponential_(1), torch.empty(self.num_envs, 2).exponential_(1)
        rew = (torch.zeros(self.num_envs, 3) / e_rew).argmax(1).float() - 1
        end = ((torch.tensor([1 - self.p_end, self.p_end]).log().expand(self.num_envs, 2).exp()) / e_end).argmax(1)
        self.t += 1
        trunc = (self.t >= (7 if self.p_end > 0 else 10 ** 6)).long()
        dead = torch.logical_or(end, trunc)
        info = {"any_dead": bool(dead.any())}
        self.state = nxt
        obs = nxt
        if info["any_dead"]:
            info["final_observation"] = nxt[dead]
            self.state = nxt.clone()
            self.state[dead] = 7.0  # "fresh episode"
            self.t[dead] = 0
            obs = self.state.clone()
        info["_dead"] = dead
        return obs, rew, end, trunc, info


class SeparablePolicy(ToyPolicy):
    """a policy whose encoder does not see the LSTM state (like ActorCritic): env_loop may encode all burn-in frames of a reset in
    one pass (`encode` + `predict_from_features`) instead of frame by frame"""

    def __init__(self, batched):
        super().__init__()
        self.encodes = 0
        if not batched:
            self.predict_from_features = None
            del self.predict_from_features

    def encode(self, obs):
        self.encodes += 1
        return obs.flatten(1)[:, :6] * 1.5 - 0.25

    def predict_from_features(self, feat, hx_cx):
        hx, cx = hx_cx
        z = feat + hx[:, :1]
        return (z[:, None, :] * self.w[None]).sum(-1), z.sum(1), (torch.tanh(hx + z[:, :5]), cx + 1)

    def predict_act_value(self, obs, hx_cx):
        self.calls += 1
        return SeparablePolicy.predict_from_features(self, self.encode(obs), hx_cx)


class BurninEnv(ToyEnv):
    """ToyEnv + what WorldModelEnv hands over at a reset: burn-in frames of the new episodes and (optionally) the device index
    list of the dead rows"""

    def __init__(self, b, p_end, with_rows):
        super().__init__(b, p_end)
        self.with_rows = with_rows

    def step(self, act):
        out = super().step(act)
        info = out[4]
        dead = info.pop("_dead")
        if info["any_dead"]:
            rows = dead.nonzero(as_tuple=True)[0]
            g = torch.Generator().manual_seed(int(rows.sum()) + 17 * len(self.log))
            info["burnin_obs"] = torch.randn(rows.numel(), 3, 2, 3, generator=g)
            if self.with_rows:
                info["dead_rows"] = rows
        return out


@pytest.mark.parametrize("p_end", [0.12, 0.35])
def test_batched_burn_in_is_bitwise_the_frame_by_frame_one(monkeypatch, p_end):
    """resets: V(final observation) and the burn-in of the policy LSTM with index lists and ONE encoder pass over all burn-in frames
    (env_loop's fast path for WorldModelEnv + ActorCritic) against boolean masks and one policy call per frame"""
    monkeypatch.setattr(EL, "sample_categorical", lambda logits, expo: (torch.softmax(logits.detach(), -1) / expo).argmax(-1))
    outs = []
    for batched in (False, True):
        torch.manual_seed(11)
        random.seed(5)
        env, pol = BurninEnv(5, p_end, with_rows=batched), SeparablePolicy(batched)
        loop = EL.make_env_loop(env, pol, epsilon=0.0)
        cols = []
        for _ in range(3):
            *c, infos = loop.send(6)
            cols.append([x.clone() for x in c])
        outs.append((cols, pol))
    for wa, wb in zip(outs[0][0], outs[1][0]):
        for a, b in zip(wa, wb):
            assert torch.equal(a, b)
    assert outs[1][1].encodes < outs[0][1].encodes, "the batched path did not save encoder passes"


# ------------------------------------------------------------------------------------------------------------------------------
# toy envs whose resets are served from a POOL in row order (so that a wrong pool order, a wrong row, a stale frame or a draw out of
# order changes the result)


def _pool_row(k, t=4):
    """the k-th initial condition the pool serves: (t, 2, 3) frames"""
    g = torch.Generator().manual_seed(1000 + k)
    return torch.randn(t, 2, 3, generator=g)


class PoolEnv:
    """Reference semantics (world_model_env.py:64-89) with plain `step`: next frame from (state, act, noise), reward / end draws,
    truncation at `horizon`, dead rows re-initialised from the pool in row order, info = final_observation / burnin_obs."""
    num_actions = 4

    def __init__(self, b, p_end, horizon, stagger=False):
        self.num_envs, self.p_end, self.horizon, self.stagger = b, p_end, horizon, stagger
        self.cursor = 0
        self.log = []

    def _serve(self, k):
        rows = [_pool_row(self.cursor + i) for i in range(k)]
        self.cursor += k
        return torch.stack(rows)

    def reset(self, **kw):
        self.ctx = self._serve(self.num_envs)  # (B, 4, 2, 3): the newest frame is ctx[:, -1]
        self.t = torch.arange(self.num_envs) % self.horizon if self.stagger else torch.zeros(self.num_envs, dtype=torch.long)
        return self.ctx[:, -1].clone(), {}

    def _dynamics(self, ctx, act, noise):
        return ctx[:, -1] * 0.5 + ctx[:, 0] * 0.1 + act.float().view(-1, 1, 1) * 0.1 + noise

    def _rew_end(self, nxt, e_rew, e_end):
        rew = ((nxt.flatten(1)[:, :3]).softmax(-1) / e_rew).argmax(1).float() - 1
        end = (torch.tensor([1 - self.p_end, self.p_end]).expand(nxt.shape[0], 2) / e_end).argmax(1)
        return rew, end

    def _draw(self):
        b = self.num_envs
        return torch.randn(b, 2, 3), torch.empty(b, 3).exponential_(1), torch.empty(b, 2).exponential_(1)

    def step(self, act):
        self.log.append("step")
        noise, e_rew, e_end = self._draw()
        nxt = self._dynamics(self.ctx, act, noise)
        rew, end = self._rew_end(nxt, e_rew, e_end)
        self.t += 1
        trunc = (self.t >= self.horizon).long()
        self.ctx = torch.cat([self.ctx[:, 1:], nxt[:, None]], 1)
        dead = torch.logical_or(end, trunc)
        info = {}
        if dead.any():
            self.ctx[dead] = self._serve(int(dead.sum()))
            self.t[dead] = 0
            info["final_observation"] = nxt[dead]
            info["burnin_obs"] = self.ctx[dead, :-1]
        return self.ctx[:, -1].clone(), rew, end, trunc, info


def _windows(env, pol, windows, t, monkeypatch):
    monkeypatch.setattr(EL, "sample_categorical", lambda logits, expo: (torch.softmax(logits.detach(), -1) / expo).argmax(-1))
    torch.manual_seed(11)
    random.seed(5)
    loop = EL.make_env_loop(env, pol, epsilon=0.0)
    outs = []
    for _ in range(windows):
        *cols, infos = loop.send(t)
        outs.append([c.clone() for c in cols])
    return outs


# ------------------------------------------------------------------------------------------------------------------------------
# The slots loop (env_loop._slots_env_loop): a step's deaths resolved "on the device" into fixed-shape reset slots, the host one
# step behind, a window repeated from its snapshot when a step had more deaths than slots -- against the sequential loop on the
# same pool-served toy env.


class ToySlots:
    def __init__(self, k, slot_row, row_slot, dead):
        self.K, self.slot_row, self.row_slot, self.dead = k, slot_row, row_slot, dead
        self.gather_rows = slot_row.clamp_min(0)

    def merge(self, base, values):
        sel = values.index_select(0, self.row_slot.clamp_min(0))
        return torch.where((self.row_slot >= 0).view(-1, *[1] * (base.ndim - 1)), sel, base)


class SlotsEnv(PoolEnv):
    """PoolEnv behind env_loop.SLOTS_PROTOCOL: K slots per step = the truncations the host mirror foresees + `margin` (the env's
    guess for sampled ends), dead rows into slots in row order, the report read one step late (an overflow shows at the NEXT
    slots_finish), snapshot / restore of env + pool cursor + the shared random stream."""

    def __init__(self, b, p_end, horizon, stagger=False, margin=1):
        super().__init__(b, p_end, horizon, stagger)
        self.margin = margin
        self._pending = self._inflight = None
        self.counts = {"overflows": 0, "slots": 0, "dead": 0, "restores": 0}

    def reset(self, **kw):
        out = super().reset(**kw)
        self.t_host = self.t.clone()
        return out

    def slots_can_repeat(self):
        return True

    def step_begin(self, act):
        assert self._pending is None
        self.log.append("begin")
        noise, e_rew, e_end = self._draw()
        self._pending = (self._dynamics(self.ctx, act, noise), e_rew, e_end)
        return self._pending[0]

    def step_end_slots(self, all_slots=False):
        nxt, e_rew, e_end = self._pending
        self._pending = None
        rew, end = self._rew_end(nxt, e_rew, e_end)
        self.slots_finish()
        b = self.num_envs
        k = b if all_slots else min(b, int((self.t_host + 1 >= self.horizon).sum()) + self.margin)
        self.t += 1
        trunc = (self.t >= self.horizon).long()
        dead = torch.logical_or(end, trunc)
        rows = dead.nonzero(as_tuple=True)[0]
        n_dead = int(rows.numel())
        self.t[dead] = 0
        used = rows[:k]
        slot_row = torch.full((max(k, 1),), -1, dtype=torch.long)
        slot_row[:used.numel()] = used
        row_slot = torch.full((b,), -1, dtype=torch.long)
        row_slot[used] = torch.arange(used.numel())
        self.ctx = torch.cat([self.ctx[:, 1:], nxt[:, None]], 1)
        fresh = torch.stack([_pool_row(self.cursor + j) for j in range(used.numel())]) if used.numel() else self.ctx[:0]
        fin = nxt[:1].repeat(k, 1, 1)
        burn = nxt[:1].repeat(3 * k, 1, 1).reshape(3, k, *nxt.shape[1:])
        if used.numel():
            self.ctx[used] = fresh
            fin[:used.numel()] = nxt[used]
            burn[:, :used.numel()] = fresh[:, :-1].transpose(0, 1)
        self._inflight = (dead.clone(), n_dead, k)
        self.counts["slots"] += k
        obs_ext = torch.cat([self.ctx[:, -1], fin, burn.reshape(3 * k, *nxt.shape[1:])])
        slots = ToySlots(k, slot_row[:k], row_slot, dead) if k > 0 else None
        return obs_ext, rew, end, trunc, slots, {"dead": dead}

    def slots_finish(self):
        inflight, self._inflight = self._inflight, None
        if inflight is None:
            return
        dead, n_dead, k = inflight
        self.t_host += 1
        self.t_host[dead] = 0
        self.counts["dead"] += n_dead
        if n_dead > k:
            self.counts["overflows"] += 1
            raise EL_SlotOverflow(f"{n_dead} > {k}")
        self.cursor += n_dead

    def slots_snapshot(self):
        self.slots_finish()
        return (self.ctx.clone(), self.t.clone(), self.t_host.clone(), self.cursor, torch.get_rng_state(), len(self.log))

    def slots_restore(self, snap):
        self.counts["restores"] += 1
        self._inflight = self._pending = None
        self.ctx, self.t, self.t_host, self.cursor = snap[0].clone(), snap[1].clone(), snap[2].clone(), snap[3]
        torch.set_rng_state(snap[4])
        del self.log[snap[5]:]


from diamond_amd.world_model_env import SlotOverflow as EL_SlotOverflow  # noqa: E402


class SlotsPolicy(SeparablePolicy):
    def __init__(self):
        super().__init__(True)

    def burn_in_from_features(self, x, num_frames):
        k = x.shape[0] // num_frames
        hz, cz = torch.zeros(k, self.lstm_dim), torch.zeros(k, self.lstm_dim)
        for i in range(num_frames):
            _, _, (hz, cz) = self.predict_from_features(x[i * k:(i + 1) * k], (hz, cz))
        return hz, cz


@pytest.mark.parametrize("margin", [0, 1, 3, 100])
@pytest.mark.parametrize("p_end,horizon,stagger", [(0.0, 6, False), (0.0, 7, True), (0.02, 7, True), (0.12, 7, True), (0.35, 5, False), (0.6, 9, True)])
def test_slots_loop_is_bitwise_the_sequential_one(monkeypatch, p_end, horizon, stagger, margin):
    """deaths resolved into reset slots with the host one step behind: truncations, sampled ends (in front of truncating rows: pool
    order), deaths at a window's last step, unused slots, and windows REPEATED from their snapshot after a slot overflow -- all on
    ONE shared random stream against the reference's order of operations"""
    b, t, windows = 9, 6, 5
    monkeypatch.setenv("DIAMOND_ENV_LOOP", "sequential")
    seq = _windows(PoolEnv(b, p_end, horizon, stagger), SlotsPolicy(), windows, t, monkeypatch)
    monkeypatch.setenv("DIAMOND_ENV_LOOP", "slots")
    env = SlotsEnv(b, p_end, horizon, stagger, margin)
    got = _windows(env, SlotsPolicy(), windows, t, monkeypatch)
    names = ("obs", "act", "rew", "end", "trunc", "logits", "val", "val_bootstrap")
    for w, (wa, wb) in enumerate(zip(seq, got)):
        for name, a, b_ in zip(names, wa, wb):
            assert torch.equal(a, b_), (w, name)
    c = env.counts
    assert c["restores"] == c["overflows"]
    if margin == 100:
        assert c["overflows"] == 0
    if margin == 0 and p_end >= 0.12:
        assert c["overflows"] > 0, "the repeated-window path was not exercised"
    if p_end > 0 or stagger or horizon % t:
        assert c["dead"] > 0


def test_slots_loop_gradients_match_the_sequential_ones(monkeypatch):
    """the burn-in of the reset rows rides in the policy's next encoder pass, unused slots included: the gradients of a window's
    loss w.r.t. the policy's parameter are those of the sequential graph"""
    grads = []
    for kind, make in (("sequential", lambda: PoolEnv(9, 0.12, 7, True)), ("slots", lambda: SlotsEnv(9, 0.12, 7, True, 2))):
        monkeypatch.setenv("DIAMOND_ENV_LOOP", kind)
        monkeypatch.setattr(EL, "sample_categorical", lambda logits, expo: (torch.softmax(logits.detach(), -1) / expo).argmax(-1))
        torch.manual_seed(11)
        random.seed(5)
        pol = SlotsPolicy()
        pol.w.requires_grad_(True)
        loop = EL.make_env_loop(make(), pol, epsilon=0.0)
        for _ in range(2):
            obs, act, rew, end, trunc, logits, val, vb, _ = loop.send(6)
        loss = (logits.logsumexp(-1) * 0.3 + (val - vb) ** 2).mean()
        (g,) = torch.autograd.grad(loss, pol.w)
        grads.append(g)
    assert torch.allclose(grads[0], grads[1], rtol=1e-6, atol=1e-7), float((grads[0] - grads[1]).abs().max())


def test_epsilon_greedy_rollouts_take_the_sequential_loop(monkeypatch):
    """epsilon > 0 (the collector's loop over real envs): the reference's order of calls whatever the env offers -- one env.step per
    step, the python generator consumed once per step (reference env_loop.py:34)"""
    monkeypatch.setattr(EL, "sample_categorical", lambda logits, expo: (torch.softmax(logits.detach(), -1) / expo).argmax(-1))
    torch.manual_seed(11)
    random.seed(5)
    calls = []
    real = random.random
    monkeypatch.setattr(random, "random", lambda: (calls.append(1), real())[1])
    env = SlotsEnv(5, 0.2, 7)
    loop = EL.make_env_loop(env, SlotsPolicy(), epsilon=0.3)
    loop.send(6)
    assert env.log.count("step") == 6 and "begin" not in env.log and len(calls) == 6


def test_slot_margin_quantile_and_estimator():
    """WorldModelEnv.slot_count: truncations the host mirror foresees + a Poisson-tail margin for sampled ends from a running mean that
    is bias-corrected while young, never below the last step's count, with a prior that fades (host logic only: no kernel, no GPU)"""
    import math

    import numpy as np
    from diamond_amd.world_model_env import WorldModelEnv, _poisson_quantile

    for mean, tail in ((0.77, 1e-4), (0.77, 1e-7), (2.56, 1e-4), (0.02, 1e-4), (30.0, 1e-4)):
        x = _poisson_quantile(mean, tail)
        cdf = lambda k: sum(math.exp(-mean) * mean ** i / math.factorial(i) for i in range(k + 1))
        assert 1.0 - cdf(x) < tail and (x == 0 or 1.0 - cdf(x - 1) >= tail), (mean, tail, x)
    assert _poisson_quantile(0.0) == 0 and _poisson_quantile(128.0, 1e-4) >= 128 + 5 * math.sqrt(128)

    env = WorldModelEnv.__new__(WorldModelEnv)
    env.num_envs, env.horizon = 256, 15
    env._ep_len_host = np.arange(256) % 15  # the staggered steady state: 17 rows are at 14 -> they truncate in the pending step
    env._end_mean, env._end_last, env._end_steps = 0.0, 0, 0
    n_trunc = int((env._ep_len_host + 1 >= 15).sum())
    assert n_trunc == 17
    k0 = env.slot_count()  # a fresh env: the prior of half an end per step gives its first `end` a spare slot
    assert k0 % 4 == 0 and n_trunc + _poisson_quantile(0.5, env.DR_END_TAIL) <= k0 < n_trunc + 12
    assert env.slot_count(all_slots=True) == 256
    rng = np.random.default_rng(0)
    for _ in range(200):  # p = 0.003: 0.77 ends per step
        n_end = int(rng.binomial(256, 0.003))
        env._end_mean = 0.95 * env._end_mean + 0.05 * n_end
        env._end_last, env._end_steps = n_end, env._end_steps + 1
    env._end_last = 0
    k = env.slot_count()
    assert n_trunc + 4 <= k <= n_trunc + 12 and k % 4 == 0, k
    env._end_last = 40  # a regime whose ends jump up is believed at once
    assert env.slot_count() >= n_trunc + 40 + 3 * 6
    env.reset_statistics()
    env._ep_len_host = np.zeros(256, dtype=np.int64)
    env._end_steps = 500  # no truncation ahead, no end seen for a long time: no slots at all (the no-ends regime pays nothing)
    assert env.slot_count() == 0
    env._ep_len_host = None  # (an env without the mirror: a slot per env)
    assert env.slot_count() == 256
