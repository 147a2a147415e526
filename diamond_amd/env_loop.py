"""Policy <-> environment rollout driver (reference coroutines/env_loop.py:12-74).

Same generator protocol as the reference (`send(num_steps)` -> stacked (B,T,...) tensors) so
`ActorCritic.forward` and the reference's collector can drive it; action sampling goes
through `sample_categorical` (argmax(softmax/E) HIP kernel, E drawn by torch on the device or
injected by tests).
"""
from __future__ import annotations

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


@coroutine
def make_env_loop(env, model, epsilon: float = 0.0, expo_fn: Optional[Callable[[Tensor], Tensor]] = None):
    num_steps = yield
    dev = model.device
    hx = torch.zeros(env.num_envs, model.lstm_dim, device=dev)
    cx = torch.zeros(env.num_envs, model.lstm_dim, device=dev)
    seed = random.randint(0, 2 ** 31 - 1)
    obs, _ = env.reset(seed=[seed + i for i in range(env.num_envs)])

    while True:
        hx, cx = hx.detach(), cx.detach()  # BPTT window boundary
        rows, infos = [], []
        dead = val_final_obs = None
        for n in range(num_steps):
            logits_act, val, (hx, cx) = model.predict_act_value(obs, (hx, cx))
            act = sample_categorical(logits_act, None if expo_fn is None else expo_fn(logits_act))
            if random.random() < epsilon:  # python RNG consumed every step, like the reference (:34)
                act = torch.randint(low=0, high=env.num_actions, size=(obs.size(0),), device=obs.device)
            next_obs, rew, end, trunc, info = env.step(act)

            if n > 0:  # the bootstrap value of step n-1 is this step's value (:39-43)
                vb = val.detach().clone()
                if dead.any():
                    vb[dead] = val_final_obs
                rows[-1][-1] = vb

            dead = torch.logical_or(end, trunc)
            if dead.any():
                with torch.no_grad():
                    _, val_final_obs, _ = model.predict_act_value(info["final_observation"], (hx[dead], cx[dead]))
                gate = 1 - dead.float().unsqueeze(1)
                hx, cx = hx * gate, cx * gate
                if "burnin_obs" in info:  # burn-in of the policy LSTM on the new episode, WITH grad (:53-56)
                    burnin = info["burnin_obs"]
                    for i in range(burnin.size(1)):
                        _, _, (hx[dead], cx[dead]) = model.predict_act_value(burnin[:, i], (hx[dead], cx[dead]))

            rows.append([obs, act, rew, end, trunc, logits_act, val, None])
            infos.append(info)
            obs = next_obs

        with torch.no_grad():
            _, vb, _ = model.predict_act_value(obs, (hx, cx))
        if dead.any():
            vb[dead] = val_final_obs
        rows[-1][-1] = vb
        stacked = tuple(torch.stack(col, dim=1) for col in zip(*rows))
        num_steps = yield (*stacked, infos)
