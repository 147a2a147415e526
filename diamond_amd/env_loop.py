"""Policy <-> environment rollout driver (reference coroutines/env_loop.py:12-74).

Same generator protocol as the reference (`send(num_steps)` -> stacked (B,T,...) tensors) so
`ActorCritic.forward` and the reference's collector can drive it; action sampling goes
through `sample_categorical` (argmax(softmax/E) HIP kernel, E drawn by torch on the device or
injected by tests).
"""
from __future__ import annotations

import os
import random
from functools import wraps
from typing import Callable, Optional

import torch
from torch import Tensor

from . import native as nv


def coroutine(func):
    @wraps(func)
    def primer(*args, **kwargs):
        gen = func(*args, **kwargs)
        next(gen)
        return gen
    return primer


def sample_categorical(logits: Tensor, expo: Optional[Tensor] = None) -> Tensor:
    """== torch.distributions.Categorical(logits=logits).sample() for the same exponential
    draws (torch.multinomial is argmax(probs / E), E ~ Exp(1)); logits (..., A)."""
    shape = logits.shape[:-1]
    a = logits.shape[-1]
    l2 = logits.detach().reshape(-1, a).float().contiguous()
    if expo is None:
        expo = torch.empty_like(l2).exponential_(1)
    expo = expo.reshape(-1, a).to(device=l2.device, dtype=torch.float32).contiguous()
    out = torch.empty(l2.shape[0], dtype=torch.long, device=l2.device)
    nv.check(nv.lib().dmd_categorical_sample(nv.fptr(l2), nv.fptr(expo), nv.ptr(out), l2.shape[0], a, nv.stream()),
             "dmd_categorical_sample")
    return out.reshape(shape)


PIPELINE_PROTOCOL = ("step_begin", "step_begin_repair", "step_end_issue", "step_end_finish", "may_speculate", "plan_resets", "policy_speculation_pays")


def _reset_chain(model, final_obs: Tensor, burnin: Tensor, new_obs: Optional[Tensor], h_d: Tensor, c_d: Tensor):
    """What a reset asks of the policy, for the k rows it concerns (reference env_loop.py:45-56 + the next step's :31):
    V(final observation) without grad on the state the episode ended with; the burn-in of the LSTM over the new episode's context
    frames WITH grad from a zero state; and -- if `new_obs` is given -- the policy's step on the new episode's newest frame.
    The encoder does not see the LSTM state, so all (1 + T-1 + 1) k frames go through it in ONE pass (`encode` +
    `predict_from_features`); per sample the arithmetic is that of separate calls (batch-invariant kernels).
    Returns (val_final, (h, c) after the burn-in, None | (logits, val, (h, c)) of the step on new_obs)."""
    k, tb = burnin.shape[:2]
    if hasattr(model, "predict_from_features"):
        parts = [final_obs, burnin.transpose(0, 1).reshape(k * tb, *burnin.shape[2:])] + ([new_obs] if new_obs is not None else [])
        feats = model.encode(torch.cat(parts))
        with torch.no_grad():
            _, val_final, _ = model.predict_from_features(feats[:k], (h_d, c_d))
        hz, cz = torch.zeros_like(h_d), torch.zeros_like(c_d)  # (the gated state of a dead row)
        for i in range(tb):
            _, _, (hz, cz) = model.predict_from_features(feats[(1 + i) * k:(2 + i) * k], (hz, cz))
        step = model.predict_from_features(feats[(1 + tb) * k:], (hz, cz)) if new_obs is not None else None
    else:
        with torch.no_grad():
            _, val_final, _ = model.predict_act_value(final_obs, (h_d, c_d))
        hz, cz = torch.zeros_like(h_d), torch.zeros_like(c_d)
        for i in range(tb):
            _, _, (hz, cz) = model.predict_act_value(burnin[:, i], (hz, cz))
        step = model.predict_act_value(new_obs, (hz, cz)) if new_obs is not None else None
    return val_final, (hz, cz), step


def _policy_step(model, obs: Tensor, hx: Tensor, cx: Tensor, resets=None):
    """The policy's step on `obs` -- and, in the SAME encoder pass, what the resets in front of it ask for (reference
    env_loop.py:45-56): resets = (rows, final_obs, burnin_obs) of the envs whose episode just ended (`obs` already shows their new
    episode's newest frame): V(final observation) without grad on the state the episode ended with, the LSTM's burn-in over the new
    episode's context frames WITH grad from a zero state (= the reference's gate-then-burn-in), then the step for everybody.
    The encoder does not see the LSTM state, so obs + final + burn-in frames are ONE batch for it (`encode` /
    `predict_from_features`; per sample the arithmetic of separate calls: batch-invariant kernels) -- no small-batch encoder pass,
    forward or backward.  Returns (logits, val, (h, c) after the step, (h0, c0) the state the step started from, val_final | None)."""
    if resets is None:
        logits, val, hc = model.predict_act_value(obs, (hx, cx))
        return logits, val, hc, (hx, cx), None
    rows, fin, burn = resets
    k, tb = burn.shape[:2]
    b = obs.shape[0]
    h_d, c_d = hx.index_select(0, rows), cx.index_select(0, rows)
    if hasattr(model, "predict_from_features"):
        feats = model.encode(torch.cat([obs, fin, burn.transpose(0, 1).reshape(k * tb, *burn.shape[2:])]))
        run = lambda lo, hi, frames, hc: model.predict_from_features(feats[lo:hi], hc)
    else:  # a model without a separable encoder: the reference's own sequence of calls
        run = lambda lo, hi, frames, hc: model.predict_act_value(frames, hc)
    hz, cz = torch.zeros_like(h_d), torch.zeros_like(c_d)  # (the gated state of a dead row)
    if tb > 0 and hasattr(model, "burn_in_from_features"):
        # the burn-in as ONE autograd node (its weight gradients once, not once per frame)
        with torch.no_grad():
            _, val_final, _ = run(b, b + k, fin, (h_d, c_d))
        hz, cz = model.burn_in_from_features(feats[b + k:b + (1 + tb) * k], tb)
        first = tb
    elif tb > 0 and hasattr(model, "predict_from_features"):
        # V(final observation) and the first burn-in step are both ONE LSTM step on k rows: one call on 2k rows (feats rows
        # [b, b + 2k) are the final observations followed by the first burn-in frames); V is used without grad
        _, v2, (h2, c2) = run(b, b + 2 * k, None, (torch.cat([h_d, hz]), torch.cat([c_d, cz])))
        val_final, hz, cz, first = v2[:k].detach(), h2[k:], c2[k:], 1
    else:
        with torch.no_grad():
            _, val_final, _ = run(b, b + k, fin, (h_d, c_d))
        first = 0
    for i in range(first, tb):
        _, _, (hz, cz) = run(b + (1 + i) * k, b + (2 + i) * k, burn[:, i], (hz, cz))
    h0, c0 = hx.index_copy(0, rows, hz), cx.index_copy(0, rows, cz)
    logits, val, hc = run(0, b, obs, (h0, c0))
    return logits, val, hc, (h0, c0), val_final


def _pipelined_env_loop(env, model, expo_fn: Optional[Callable[[Tensor], Tensor]], num_steps: int):
    """make_env_loop for an env that implements WorldModelEnv's pipelining protocol (PIPELINE_PROTOCOL), epsilon = 0.

    Per imagined step the reference has ONE data-dependent branch (`if dead.any()`, world_model_env.py:77, env_loop.py:45): a
    host wait for the device.  Two things are done about it, both exact:

    (a) ONE encoder pass per step.  What a reset asks of the policy -- V(final observation), the burn-in frames of the new
        episode -- rides in the batch of the policy's next step (_policy_step): no small-batch encoder pass forward or backward.
    (b) While unforeseen deaths are rare (env.policy_speculation_pays()), the work of step n + 1 is issued BEFORE the wait of
        step n: the policy's step n + 1 on the imagined frame (between env.step_begin and env.step_end_issue); the sampler's step
        n + 1 (env.step_begin(..., speculative=True) between step_end_issue and step_end_finish) when the env says it pays
        (may_speculate); and the resets the host can foresee -- truncations: env.plan_resets() -- are part of that pipeline:
        their frames are in the speculated policy pass, the env resets those rows right behind the reward / end model, and the
        speculative sampler step already runs on the new episodes.  Deaths nobody planned (`end` sampled by the reward / end
        model) void ONLY their own rows (info["void_rows"]): the policy is recomputed for them after the reset with the
        exponential draws the speculation made, the env repeats their sampler step (step_begin_repair), every other row keeps
        its speculated result.  Where unforeseen deaths are frequent nothing is speculated: the reference's order, with (a).

    Per row the same arithmetic in the same order on batch-invariant kernels, every random stream consumed in the reference's
    order: bitwise the sequential rollout whatever the mode (tests/test_env_loop_host.py on a toy env; the window goldens and the
    sequential A/B on the GPU)."""
    dev = model.device
    hx = torch.zeros(env.num_envs, model.lstm_dim, device=dev)
    cx = torch.zeros(env.num_envs, model.lstm_dim, device=dev)
    seed = random.randint(0, 2 ** 31 - 1)
    obs, _ = env.reset(seed=[seed + i for i in range(env.num_envs)])

    def draw_expo(logits: Tensor) -> Tensor:
        e = expo_fn(logits) if expo_fn is not None else None
        if e is None:
            return torch.empty(logits.shape, device=logits.device, dtype=torch.float32).exponential_(1)
        return e.to(device=logits.device, dtype=torch.float32)  # (hook-injected draws may come from the CPU generator)

    def merge(cand, rows: Tensor, step, expo: Tensor):
        """rows of the candidate policy output <- the step recomputed for them after their reset (their own exponential draws)"""
        logits, val, (h, c) = step
        act = sample_categorical(logits, expo.index_select(0, rows))
        return [cand[0].index_copy(0, rows, logits), cand[1].index_copy(0, rows, val),
                (cand[2][0].index_copy(0, rows, h), cand[2][1].index_copy(0, rows, c)), cand[3].index_copy(0, rows, act)]

    def full(rows: Tensor, v: Tensor, like: Tensor) -> Tensor:
        return torch.zeros_like(like).index_copy(0, rows, v)

    pending = None  # (rows, final_obs, burnin_obs): deaths whose policy-side reset rides in the next _policy_step
    while True:
        hx, cx = hx.detach(), cx.detach()  # BPTT window boundary
        rows_out, infos = [], []
        pol = None    # (logits, val, (hx, cx), act) of this step when it was issued during the previous one
        begun = None  # the imagined frame of this step when env.step_begin was issued during the previous one
        saved_expo = None  # the exponential draws of a speculated policy step that was dropped: its repetition uses them
        prev_dead = prev_vfinal = None
        for n in range(num_steps):
            if pol is None:
                logits_act, val, (hx, cx), _, vfin = _policy_step(model, obs, hx, cx, pending)
                if pending is not None:
                    prev_vfinal, pending = full(pending[0], vfin, val.detach()), None
                act = sample_categorical(logits_act, saved_expo if saved_expo is not None else draw_expo(logits_act))
                saved_expo = None
            else:
                logits_act, val, (hx, cx), act = pol
                pol = None
            if n > 0:  # the bootstrap value of step n-1 is this step's value, V(final observation) where the episode ended (:39-43)
                vb = val.detach().clone()
                rows_out[-1][-1] = vb if prev_dead is None else torch.where(prev_dead, prev_vfinal, vb)
            nxt = begun if begun is not None else env.step_begin(act)
            begun = None
            cand = s_expo = vfinal = None
            if n + 1 < num_steps and env.policy_speculation_pays():
                plan = env.plan_resets()
                if plan is None:
                    c_logits, c_val, c_hc, _, _ = _policy_step(model, nxt, hx, cx, None)
                else:
                    r = plan["rows"]
                    c_logits, c_val, c_hc, _, v = _policy_step(model, nxt.index_copy(0, r, plan["obs"]), hx, cx,
                                                               (r, nxt.index_select(0, r), plan["burnin_obs"]))
                    vfinal = full(r, v, val.detach())
                s_expo = draw_expo(c_logits)
                cand = [c_logits, c_val, c_hc, sample_categorical(c_logits, s_expo)]
            env.step_end_issue()
            if cand is not None and env.may_speculate():
                begun = env.step_begin(cand[3], speculative=True)
            elif cand is None and n + 1 < num_steps and hasattr(env, "predraw"):
                # nothing of the next step is issued ahead -- but its random draws depend on nothing: the policy's exponentials,
                # then the env's (the reference's order), in front of the host synchronisation instead of behind it
                saved_expo = draw_expo(logits_act)
                env.predraw()
            next_obs, rew, end, trunc, info = env.step_end_finish()

            prev_dead = prev_vfinal = None
            if info["any_dead"]:
                prev_dead = info["dead"] if "dead" in info else torch.logical_or(end, trunc)
                void = info.get("void_rows")
                if cand is None:  # nothing was speculated: the resets ride in the next policy step (all of them: no plan either)
                    pending = (info["dead_rows"], info["final_observation"], info["burnin_obs"])
                elif void is not None and not info.get("repair_pending"):
                    # deaths no plan covered, and no sampler step in flight that would keep the device busy meanwhile: drop the
                    # speculated policy step (its draws are kept) -- ALL of this step's resets ride in its repetition's encoder
                    # pass, which costs less than a small-batch chain for the void rows now plus its own backward later
                    pending, cand, saved_expo, vfinal = (info["dead_rows"], info["final_observation"], info["burnin_obs"]), None, s_expo, None
                elif void is not None:  # ... with the next sampler step already queued: their part now, on the rows concerned only
                    pos = info.get("void_pos")
                    fin, burn = info["final_observation"], info["burnin_obs"]
                    if pos is not None:
                        fin, burn = fin.index_select(0, pos), burn.index_select(0, pos)
                    v, _, step = _reset_chain(model, fin, burn, next_obs.index_select(0, void), hx.index_select(0, void), cx.index_select(0, void))
                    vfinal = (torch.zeros_like(val.detach()) if vfinal is None else vfinal).index_copy(0, void, v)
                    cand = merge(cand, void, step, s_expo)
                    begun = env.step_begin_repair(cand[3])
                prev_vfinal = vfinal
            pol = cand
            rows_out.append([obs, act, rew, end, trunc, logits_act, val, None])
            infos.append(info)
            obs = next_obs

        with torch.no_grad():
            _, vb, _, (h0, c0), vfin = _policy_step(model, obs, hx, cx, pending)
        if pending is not None:  # deaths at the window's last step: the state the next window starts from is the burnt-in one
            prev_vfinal, pending = full(pending[0], vfin, vb), None
            hx, cx = h0, c0
        rows_out[-1][-1] = vb if prev_dead is None else torch.where(prev_dead, prev_vfinal, vb)
        stacked = tuple(torch.stack(col, dim=1) for col in zip(*rows_out))
        num_steps = yield (*stacked, infos)


SLOTS_PROTOCOL = ("step_begin", "step_end_slots", "slots_finish", "slots_snapshot", "slots_restore", "slots_can_repeat")


def _policy_step_slots(model, obs_ext: Tensor, hx: Tensor, cx: Tensor, slots):
    """The policy's step on the batch -- and, in the SAME encoder pass, what the previous step's resets ask of it (reference
    env_loop.py:45-56), fixed-shape over the reset SLOTS the device filled (WorldModelEnv.step_end_slots): obs_ext =
    [B newest frames | K final observations | (T - 1) * K burn-in frames, frame-major].  V(final observation) without grad on the
    state the episode ended with (:49); the LSTM's burn-in over the new episode's context frames WITH grad from a zero state
    (:51-56, one autograd node); the burnt-in states merged into the batch's (slots.merge); then the step for everybody.  Unused
    slots hold finite stand-in frames, their results go nowhere and receive zero gradient.  Per row the arithmetic of the
    reference's separate calls (batch-invariant kernels).  Returns (logits, val, (h, c), (h0, c0) the state the step started
    from, V(final observation) spread over the batch | None)."""
    if slots is None:
        logits, val, hc = model.predict_act_value(obs_ext, (hx, cx))
        return logits, val, hc, (hx, cx), None
    b, k = hx.shape[0], slots.K
    tb = (obs_ext.shape[0] - b) // k - 1
    feats = model.encode(obs_ext)
    g = slots.gather_rows
    with torch.no_grad():
        _, val_final, _ = model.predict_from_features(feats[b:b + k], (hx.index_select(0, g), cx.index_select(0, g)))
    if tb > 0:
        hz, cz = model.burn_in_from_features(feats[b + k:], tb)
    else:
        hz, cz = torch.zeros_like(hx[:k]), torch.zeros_like(cx[:k])
    h0, c0 = slots.merge(hx, hz), slots.merge(cx, cz)
    logits, val, hc = model.predict_from_features(feats[:b], (h0, c0))
    return logits, val, hc, (h0, c0), slots.merge(torch.zeros_like(val.detach()), val_final.detach())


def _slots_env_loop(env, model, expo_fn: Optional[Callable[[Tensor], Tensor]], num_steps: int):
    """make_env_loop for an env that resolves a step's deaths on the device (SLOTS_PROTOCOL: WorldModelEnv), epsilon = 0.

    The reference has ONE data-dependent branch per imagined step (`if dead.any()`, world_model_env.py:77, env_loop.py:45): a host
    wait, behind which the host issues the reset and the policy's next step into an idle device.  Here the host never asks about
    the step it is issuing: env.step_end_slots() hands back device-side reset slots, the policy's next step carries them in its
    encoder pass (_policy_step_slots), and the env reads a step's report while the NEXT step is already queued.  The only waits are
    for the previous step (never idle: the device holds a full step of work) and one per window, before the loss.
    Bitwise the sequential rollout: per row the same arithmetic on batch-invariant kernels, every random stream consumed in the
    reference's order (action draws, then the env's noise / reward / end draws; a reset draws nothing).  A step with more deaths
    than slots (SlotOverflow; the env sizes the slots so that this is a once-in-1e7-steps event) restores the window's
    snapshot -- env, pool position, random generators -- and repeats the window with a slot per env."""
    from .world_model_env import SlotOverflow

    dev = model.device
    b = env.num_envs
    hx = torch.zeros(b, model.lstm_dim, device=dev)
    cx = torch.zeros(b, model.lstm_dim, device=dev)
    seed = random.randint(0, 2 ** 31 - 1)
    obs, _ = env.reset(seed=[seed + i for i in range(b)])
    slots = None  # the resets of the last step: they ride in the next policy step

    def draw_expo(logits: Tensor) -> Tensor:
        e = expo_fn(logits) if expo_fn is not None else None
        if e is None:
            return torch.empty(logits.shape, device=logits.device, dtype=torch.float32).exponential_(1)
        return e.to(device=logits.device, dtype=torch.float32)  # (hook-injected draws may come from the CPU generator)

    def window(obs, slots, hx, cx, steps: int, all_slots: bool):
        rows_out, infos = [], []
        prev_dead = prev_vfinal = None
        for n in range(steps):
            logits_act, val, (hx, cx), _, vfin = _policy_step_slots(model, obs, hx, cx, slots)
            if slots is not None:
                prev_dead, prev_vfinal = slots.dead, vfin
            act = sample_categorical(logits_act, draw_expo(logits_act))
            if n > 0:  # the bootstrap value of step n-1 is this step's value, V(final observation) where the episode ended (:39-43)
                vb = val.detach().clone()
                rows_out[-1][-1] = vb if prev_dead is None else torch.where(prev_dead, prev_vfinal, vb)
            prev_dead = prev_vfinal = None
            env.step_begin(act)
            nxt, rew, end, trunc, slots, info = env.step_end_slots(all_slots)
            rows_out.append([obs[:b], act, rew, end, trunc, logits_act, val, None])
            infos.append(info)
            obs = nxt
        # the one wait of a window for its OWN last step (an overflow must show before the window is used) -- in front of the
        # bootstrap pass, which then keeps the device busy while the host stacks the window and builds the loss
        env.slots_finish()
        with torch.no_grad():
            _, vb, _, (h0, c0), vfin = _policy_step_slots(model, obs, hx, cx, slots)
        if slots is not None:  # deaths at the window's last step: the state the next window starts from is the burnt-in one
            vb = torch.where(slots.dead, vfin, vb)
            hx, cx = h0, c0
        rows_out[-1][-1] = vb
        stacked = tuple(torch.stack(col, dim=1) for col in zip(*rows_out))
        return (*stacked, infos), (obs[:b], None, hx, cx)

    while True:
        hx, cx = hx.detach(), cx.detach()  # BPTT window boundary
        hooks = expo_fn is not None and getattr(model, "expo_fn", True) is not None  # (ActorCritic passes a forwarding lambda)
        can_repeat = env.slots_can_repeat() and not hooks
        snap = env.slots_snapshot() if can_repeat else None
        try:
            out, (obs_n, slots_n, hx_n, cx_n) = window(obs, slots, hx, cx, num_steps, not can_repeat)
        except SlotOverflow:
            if snap is None:
                raise
            env.slots_restore(snap)
            out, (obs_n, slots_n, hx_n, cx_n) = window(obs, slots, hx, cx, num_steps, True)
        obs, slots, hx, cx = obs_n, slots_n, hx_n, cx_n
        num_steps = yield out


@coroutine
def make_env_loop(env, model, epsilon: float = 0.0, expo_fn: Optional[Callable[[Tensor], Tensor]] = None):
    num_steps = yield
    if epsilon == 0.0 and os.environ.get("DIAMOND_SPECULATIVE_POLICY", "1") == "1":
        # DIAMOND_ENV_LOOP: "slots" (default: the step's deaths resolved on the device), "pipelined" (round 5: host-planned resets
        # and speculation), "sequential" (the reference's order of calls)
        kind = os.environ.get("DIAMOND_ENV_LOOP", "slots")
        assert kind in ("slots", "pipelined", "sequential"), f"DIAMOND_ENV_LOOP={kind!r}"
        separable = all(hasattr(model, a) for a in ("encode", "predict_from_features", "burn_in_from_features"))
        if kind == "slots" and separable and all(hasattr(env, a) for a in SLOTS_PROTOCOL):
            yield from _slots_env_loop(env, model, expo_fn, num_steps)
            return
        if kind != "sequential" and all(hasattr(env, a) for a in PIPELINE_PROTOCOL):
            yield from _pipelined_env_loop(env, model, expo_fn, num_steps)
            return
    dev = model.device
    hx = torch.zeros(env.num_envs, model.lstm_dim, device=dev)
    cx = torch.zeros(env.num_envs, model.lstm_dim, device=dev)
    seed = random.randint(0, 2 ** 31 - 1)
    obs, _ = env.reset(seed=[seed + i for i in range(env.num_envs)])

    # Two-phase env (WorldModelEnv.step_begin / step_end): the policy's NEXT step is issued between the imagined frame and
    # the reward / end model, i.e. BEFORE the step's one host synchronisation (`if dead.any()`), on the assumption that no
    # episode ends.  The device then already holds the next action when the host comes back from that wait and the
    # denoiser can be launched at once; without this the GPU idles ~1 ms per step while the host issues the policy's ~40
    # small launches.  If an episode did end, the speculative result is dropped and the step is recomputed after the reset,
    # with the SAME exponential draws: every random stream is consumed in the reference's order either way.
    spec_mode = os.environ.get("DIAMOND_SPECULATIVE_POLICY", "1")  # "0": neither speculation, "policy": the policy step only
    assert spec_mode in ("0", "1", "policy"), f"DIAMOND_SPECULATIVE_POLICY={spec_mode!r}: one of 0, 1, policy"
    two_phase = hasattr(env, "step_begin") and spec_mode != "0"
    # ... and, where the env can hand its synchronisation over (WorldModelEnv.step_end_issue / step_end_finish), the NEXT
    # step's imagined frame as well: env.step_begin(act of step n + 1) is issued before the host asks whether an episode
    # ended in step n.  The device then holds a whole sampler step of queued work while the host waits, instead of running
    # dry until the host has issued the first launches of the next step (~1-2 ms per step on a 20 ms step).  A speculation an
    # ended episode voids is DROPPED by such an env, which keeps its draws for the repetition.  That is the protocol of an env
    # WITHOUT step_begin_repair (the toy envs of tests/test_env_loop_host.py): WorldModelEnv keeps a voided half-step pending and
    # expects step_begin_repair (_pipelined_env_loop above), so it never takes this branch.  Not with epsilon-greedy actions
    # either (the override of step n + 1 is drawn at the top of that step: the speculated action could change).
    three_phase = (two_phase and spec_mode != "policy" and epsilon == 0.0 and not hasattr(env, "step_begin_repair")
                   and all(hasattr(env, a) for a in ("step_end_issue", "step_end_finish", "may_speculate")))

    def draw_expo(logits: Tensor) -> Tensor:
        if expo_fn is not None:
            return expo_fn(logits)
        return torch.empty(logits.shape, device=logits.device, dtype=torch.float32).exponential_(1)

    while True:
        hx, cx = hx.detach(), cx.detach()  # BPTT window boundary
        rows, infos = [], []
        dead = val_final_obs = ridx = None
        any_dead = False
        spec = None        # (logits, val, (hx, cx), act) of this step, issued during the previous one
        saved_expo = None  # draws of a dropped speculative step, to be used by its recomputation
        begun = None       # the imagined frame of this step if env.step_begin(act) was issued during the previous one
        for n in range(num_steps):
            if spec is not None:
                logits_act, val, (hx, cx), act = spec
                spec = None
            else:
                logits_act, val, (hx, cx) = model.predict_act_value(obs, (hx, cx))
                expo = saved_expo if saved_expo is not None else draw_expo(logits_act)
                saved_expo = None
                act = sample_categorical(logits_act, expo)
            if random.random() < epsilon:  # python RNG consumed every step, like the reference (:34)
                act = torch.randint(low=0, high=env.num_actions, size=(obs.size(0),), device=obs.device)
            cand = None
            if two_phase:
                nxt, begun = (begun, None) if begun is not None else (env.step_begin(act), None)
                if n + 1 < num_steps:
                    s_logits, s_val, s_hc = model.predict_act_value(nxt, (hx, cx))
                    s_expo = draw_expo(s_logits)
                    cand = (s_logits, s_val, s_hc, sample_categorical(s_logits, s_expo))
                if three_phase:
                    env.step_end_issue()
                    if cand is not None and env.may_speculate():
                        begun = env.step_begin(cand[3], speculative=True)
                    next_obs, rew, end, trunc, info = env.step_end_finish()
                else:
                    next_obs, rew, end, trunc, info = env.step_end()
            else:
                next_obs, rew, end, trunc, info = env.step(act)

            if n > 0:  # the bootstrap value of step n-1 is this step's value (:39-43)
                vb = val.detach().clone()
                if any_dead:
                    vb = vb.index_copy(0, ridx, val_final_obs) if ridx is not None else vb.masked_scatter(dead, val_final_obs)
                rows[-1][-1] = vb

            dead = torch.logical_or(end, trunc)
            any_dead = info["any_dead"] if "any_dead" in info else bool(dead.any())
            if any_dead:
                begun = None  # (the env dropped the speculative half-step itself and kept its draws)
                if cand is not None:
                    saved_expo, cand = s_expo, None  # an episode ended: this step's policy output is recomputed after the reset
                ridx = info.get("dead_rows")  # device index list of the dead rows (WorldModelEnv), else boolean masks
                if ridx is not None:
                    h_d, c_d = hx.index_select(0, ridx), cx.index_select(0, ridx)
                else:
                    h_d, c_d = hx[dead], cx[dead]
                with torch.no_grad():
                    _, val_final_obs, _ = model.predict_act_value(info["final_observation"], (h_d, c_d))
                gate = 1 - dead.float().unsqueeze(1)
                hx, cx = hx * gate, cx * gate
                if "burnin_obs" in info:  # burn-in of the policy LSTM on the new episode, WITH grad (:53-56)
                    burnin = info["burnin_obs"]
                    if ridx is not None and hasattr(model, "predict_from_features"):
                        # the encoder does not depend on the LSTM state: all burn-in frames in ONE pass (frame-major), then
                        # the recurrence over their features -- per sample the same arithmetic as frame by frame (the
                        # kernels are batch-invariant), a third of the launches forward and backward, no boolean-mask
                        # gathers (each one a host synchronisation)
                        nd, tb = burnin.shape[:2]
                        feats = model.encode(burnin.transpose(0, 1).reshape(nd * tb, *burnin.shape[2:]))
                        hz, cz = torch.zeros_like(h_d), torch.zeros_like(c_d)  # (the gated state of a dead row)
                        for i in range(tb):
                            _, _, (hz, cz) = model.predict_from_features(feats[i * nd:(i + 1) * nd], (hz, cz))
                        hx, cx = hx.index_copy(0, ridx, hz), cx.index_copy(0, ridx, cz)
                    else:
                        for i in range(burnin.size(1)):
                            _, _, (hx[dead], cx[dead]) = model.predict_act_value(burnin[:, i], (hx[dead], cx[dead]))
            spec = cand

            rows.append([obs, act, rew, end, trunc, logits_act, val, None])
            infos.append(info)
            obs = next_obs

        with torch.no_grad():
            _, vb, _ = model.predict_act_value(obs, (hx, cx))
        if any_dead:
            vb = vb.index_copy(0, ridx, val_final_obs) if ridx is not None else vb.masked_scatter(dead, val_final_obs)
        rows[-1][-1] = vb
        stacked = tuple(torch.stack(col, dim=1) for col in zip(*rows))
        num_steps = yield (*stacked, infos)
