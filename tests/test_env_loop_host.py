"""Host logic of env_loop's speculative policy step on the CPU (no kernel: a torch toy policy, a toy two-phase env and a torch
stand-in for the categorical-sample kernel): the policy's step n + 1 is issued between env.step_begin and env.step_end and dropped
when an episode ended.  With ONE shared random stream for every draw (policy exponentials, env noise, reward / end
exponentials) the rollout must be bitwise the sequential one -- i.e. the stream is consumed in the same order -- and the env
must see exactly the same sequence of calls.  (The GPU twin with the real models: tests/test_gpu_models.py.)"""
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
        logits = feat @ self.w.t()
        val = feat.sum(1)
        return logits, val, (torch.tanh(hx + feat[:, :5]), cx + 1)


class ToyEnv:
    """WorldModelEnv's protocol: step = step_begin + step_end, deaths reset the row, info carries final_observation / any_dead."""
    num_actions = 4

    def __init__(self, b, p_end, two_phase):
        self.num_envs, self.p_end = b, p_end
        if not two_phase:
            self.step_begin = None  # (hasattr is what env_loop looks at)
            del self.step_begin
        self.log = []

    def reset(self, **kw):
        self.state = torch.randn(self.num_envs, 2, 3)
        self.t = torch.zeros(self.num_envs, dtype=torch.long)
        return self.state.clone(), {}

    def _begin(self, act):
        self.log.append("begin")
        nxt = self.state * 0.5 + act.float().view(-1, 1, 1) * 0.1 + torch.randn(self.num_envs, 2, 3)  # "denoiser noise"
        self._pending = (nxt, torch.empty(self.num_envs, 3).exponential_(1), torch.empty(self.num_envs, 2).exponential_(1))
        return nxt

    def _end(self):
        self.log.append("end")
        nxt, e_rew, e_end = self._pending
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
        return obs, rew, end, trunc, info


class ToyEnv3(ToyEnv):
    """... with WorldModelEnv's third phase: step_end = step_end_issue + step_end_finish, a step_begin(act, speculative=True) may
    be issued in between; an ended episode drops it and keeps its draws for the repetition (same bookkeeping as the real env)."""

    def __init__(self, b, p_end):
        super().__init__(b, p_end, True)
        self._saved, self._spec, self.cool, self.wasted, self.used = None, False, 0, 0, 0

    def step_begin(self, act, speculative=False):
        self.log.append("begin-spec" if speculative else "begin")
        if self._saved is not None:
            noise, e_rew, e_end = self._saved
            self._saved = None
        else:
            noise = torch.randn(self.num_envs, 2, 3)
            e_rew, e_end = torch.empty(self.num_envs, 3).exponential_(1), torch.empty(self.num_envs, 2).exponential_(1)
        nxt = self.state * 0.5 + act.float().view(-1, 1, 1) * 0.1 + noise
        self._pending, self._spec = (nxt, e_rew, e_end, noise), speculative
        return nxt

    def may_speculate(self):
        return self.cool == 0

    def step_end_issue(self):
        nxt, e_rew, e_end, _ = self._pending
        self._pending, self._spec = None, False
        rew = (torch.zeros(self.num_envs, 3) / e_rew).argmax(1).float() - 1
        end = ((torch.tensor([1 - self.p_end, self.p_end]).log().expand(self.num_envs, 2).exp()) / e_end).argmax(1)
        self.t += 1
        trunc = (self.t >= (7 if self.p_end > 0 else 10 ** 6)).long()
        self._issued = (nxt, rew, end, trunc, torch.logical_or(end, trunc))
        self.state = nxt  # (the ring advance: device-side state the speculative step_begin reads)

    def step_end_finish(self):
        self.log.append("end")
        nxt, rew, end, trunc, dead = self._issued
        info = {"any_dead": bool(dead.any())}
        self.cool = max(0, self.cool - 1)
        if self._pending is not None and self._spec:
            if info["any_dead"]:
                self._saved = (self._pending[3], self._pending[1], self._pending[2])
                self._pending, self._spec = None, False
                self.wasted += 1
            else:
                self.used += 1
        obs = nxt
        if info["any_dead"]:
            self.cool = 2
            info["final_observation"] = nxt[dead]
            self.state = nxt.clone()
            self.state[dead] = 7.0
            self.t[dead] = 0
            obs = self.state.clone()
        return obs, rew, end, trunc, info

    def step_end(self):
        self.step_end_issue()
        return self.step_end_finish()


def _make_env(b, p_end, two_phase):
    if two_phase == 3:
        return ToyEnv3(b, p_end)
    env = ToyEnv(b, p_end, two_phase)
    if two_phase:
        env.step_begin, env.step_end = env._begin, env._end
    env.step = lambda act: (env._begin(act), env._end())[1]
    return env


def _rollout(monkeypatch, two_phase, p_end, windows=3, t=6, b=5, epsilon=0.2):
    monkeypatch.setattr(EL, "sample_categorical", lambda logits, expo: (torch.softmax(logits.detach(), -1) / expo).argmax(-1))
    torch.manual_seed(11)
    random.seed(5)
    env, pol = _make_env(b, p_end, two_phase), ToyPolicy()
    loop = EL.make_env_loop(env, pol, epsilon=epsilon)
    outs = []
    for _ in range(windows):
        *cols, infos = loop.send(t)
        outs.append([c.clone() for c in cols])
    return outs, env, pol


@pytest.mark.parametrize("p_end", [0.0, 0.35])
def test_speculative_policy_step_consumes_the_streams_in_the_sequential_order(monkeypatch, p_end):
    seq, env_s, pol_s = _rollout(monkeypatch, False, p_end)
    spec, env_p, pol_p = _rollout(monkeypatch, True, p_end)
    for wa, wb in zip(seq, spec):
        for a, b in zip(wa, wb):
            assert torch.equal(a, b)
    assert env_s.log == env_p.log
    deaths = sum(int(w[3].sum() + w[4].sum()) for w in seq)
    if p_end == 0.0:
        assert deaths == 0 and pol_p.calls == pol_s.calls  # nobody ends mid-window: every speculative step is used
    else:
        assert deaths > 0 and pol_p.calls > pol_s.calls  # dropped speculative steps were recomputed


@pytest.mark.parametrize("p_end", [0.0, 0.12, 0.35])
def test_speculative_sampler_step_consumes_the_streams_in_the_sequential_order(monkeypatch, p_end):
    """the next step's env.step_begin issued between step_end_issue and step_end_finish (WorldModelEnv's third phase): bitwise the
    sequential rollout on ONE shared random stream, whether the speculation is used or dropped and repeated"""
    seq, env_s, pol_s = _rollout(monkeypatch, False, p_end, epsilon=0.0)
    spec, env_p, pol_p = _rollout(monkeypatch, 3, p_end, epsilon=0.0)
    for wa, wb in zip(seq, spec):
        for a, b in zip(wa, wb):
            assert torch.equal(a, b)
    assert env_p.used > 0 or p_end > 0.3, "no speculative step was ever used"  # (5 envs at p = 0.35: somebody ends at almost every step)
    if p_end == 0.0:
        assert env_p.wasted == 0 and sorted(x.replace("-spec", "") for x in env_p.log) == sorted(env_s.log)
    else:
        assert env_p.wasted > 0, "the dropped-and-repeated path was not exercised"
        # a dropped half-step shows as an extra begin: begin-spec (dropped) ... begin (repetition)
        assert env_p.log.count("begin") + env_p.log.count("begin-spec") == env_s.log.count("begin") + env_p.wasted


def test_epsilon_greedy_rollouts_do_not_speculate_the_sampler(monkeypatch):
    """the epsilon override of step n + 1 is drawn at the top of that step: the action a speculative step_begin used could change"""
    _, env, _ = _rollout(monkeypatch, 3, 0.0, epsilon=0.2)
    assert "begin-spec" not in env.log


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
        return z @ self.w.t(), z.sum(1), (torch.tanh(hx + z[:, :5]), cx + 1)

    def predict_act_value(self, obs, hx_cx):
        self.calls += 1
        return SeparablePolicy.predict_from_features(self, self.encode(obs), hx_cx)


class BurninEnv(ToyEnv3):
    """ToyEnv3 + what WorldModelEnv hands over at a reset: burn-in frames of the new episodes and (optionally) the device index
    list of the dead rows"""

    def __init__(self, b, p_end, with_rows):
        super().__init__(b, p_end)
        self.with_rows = with_rows

    def step_end_finish(self):
        nxt, rew, end, trunc, dead = self._issued
        out = super().step_end_finish()
        info = out[4]
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
