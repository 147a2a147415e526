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


def _slots_env_loop(env, model, expo_fn: Optional[Callable[[Tensor], Tensor]], num_steps: int, obs: Tensor):
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
    """Reference coroutines/env_loop.py:12-74.  For an env that resolves a step's deaths on the device (WorldModelEnv) and a policy
    with a separable encoder (ActorCritic), epsilon = 0: _slots_env_loop above -- DIAMOND_ENV_LOOP=sequential selects the loop
    below for them too (A/B, tests).  Anything else (the real envs of the collector, epsilon-greedy, other policies): the
    reference's order of calls, one `env.step` and one host look at `dead` per step."""
    num_steps = yield
    kind = os.environ.get("DIAMOND_ENV_LOOP", "slots")
    assert kind in ("slots", "sequential"), f"DIAMOND_ENV_LOOP={kind!r}: slots | sequential"
    separable = all(hasattr(model, a) for a in ("encode", "predict_from_features", "burn_in_from_features"))
    dev = model.device
    seed = random.randint(0, 2 ** 31 - 1)
    obs, _ = env.reset(seed=[seed + i for i in range(env.num_envs)])
    # (an env may decline, once it has been reset: WorldModelEnv with a replayed sampler graph -- at most 8 small frames -- is
    #  latency-bound anyway, and its in-graph random draws cannot be rewound for a window's repetition: it would need a slot per
    #  env, i.e. 5x the encoder frames, at every step)
    if (epsilon == 0.0 and kind == "slots" and separable and all(hasattr(env, a) for a in SLOTS_PROTOCOL)
            and getattr(env, "slots_preferred", lambda: True)()):
        yield from _slots_env_loop(env, model, expo_fn, num_steps, obs)
        return
    hx = torch.zeros(env.num_envs, model.lstm_dim, device=dev)
    cx = torch.zeros(env.num_envs, model.lstm_dim, device=dev)

    def draw_expo(logits: Tensor) -> Tensor:
        e = expo_fn(logits) if expo_fn is not None else None
        if e is None:
            return torch.empty(logits.shape, device=logits.device, dtype=torch.float32).exponential_(1)
        return e.to(device=logits.device, dtype=torch.float32)

    while True:
        hx, cx = hx.detach(), cx.detach()  # BPTT window boundary
        rows, infos = [], []
        dead = val_final_obs = ridx = None
        any_dead = False
        for n in range(num_steps):
            logits_act, val, (hx, cx) = model.predict_act_value(obs, (hx, cx))
            act = sample_categorical(logits_act, draw_expo(logits_act))
            if random.random() < epsilon:  # python RNG consumed every step, like the reference (:34)
                act = torch.randint(low=0, high=env.num_actions, size=(obs.size(0),), device=obs.device)
            next_obs, rew, end, trunc, info = env.step(act)

            if n > 0:  # the bootstrap value of step n-1 is this step's value (:39-43)
                vb = val.detach().clone()
                if any_dead:
                    vb = vb.index_copy(0, ridx, val_final_obs) if ridx is not None else vb.masked_scatter(dead, val_final_obs)
                rows[-1][-1] = vb

            dead = torch.logical_or(end, trunc)
            any_dead = info["any_dead"] if "any_dead" in info else bool(dead.any())  # THE host look of a step (:45)
            if any_dead:
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
