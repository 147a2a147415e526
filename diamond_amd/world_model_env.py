"""Batched imagination environment (reference envs/world_model_env.py:25-139).

Same constructor / attributes / step contract as the reference so src/trainer.py and
src/play.py can use it unchanged; `predict_next_obs` and `predict_rew_end` are plain
instance-assignable callables (trainer.py:182-184 overwrites them).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict, Generator, List, Optional, Tuple

import torch
from torch import Tensor

from .diffusion_sampler import DiffusionSampler, DiffusionSamplerConfig
from .env_loop import coroutine, sample_categorical


@dataclass
class WorldModelEnvConfig:
    horizon: int
    num_batches_to_preload: int
    diffusion_sampler: DiffusionSamplerConfig


class WorldModelEnv:
    def __init__(self, denoiser, rew_end_model, data_loader, cfg: WorldModelEnvConfig,
                 return_denoising_trajectory: bool = False) -> None:
        self.sampler = DiffusionSampler(denoiser, cfg.diffusion_sampler)
        self.rew_end_model = rew_end_model
        self.horizon = cfg.horizon
        self.return_denoising_trajectory = return_denoising_trajectory
        self.num_envs = data_loader.batch_sampler.batch_size
        self.generator_init = self.make_generator_init(data_loader, cfg.num_batches_to_preload)
        # test hook: injected exponential draws for the reward / end samples (host RNG parity)
        self.expo_fn: Optional[Callable[[Tensor], Tensor]] = None

    @property
    def device(self) -> torch.device:
        return self.sampler.denoiser.device

    @torch.no_grad()
    def reset(self, **kwargs) -> Tuple[Tensor, Dict[str, Any]]:
        obs, act, (hx, cx) = self.generator_init.send(self.num_envs)
        self.obs_buffer, self.act_buffer = obs, act
        self.hx_rew_end, self.cx_rew_end = hx, cx
        self.ep_len = torch.zeros(self.num_envs, dtype=torch.long, device=obs.device)
        return self.obs_buffer[:, -1], {}

    @torch.no_grad()
    def reset_dead(self, dead: Tensor) -> None:
        obs, act, (hx, cx) = self.generator_init.send(int(dead.sum().item()))
        self.obs_buffer[dead] = obs
        self.act_buffer[dead] = act
        self.hx_rew_end[:, dead] = hx
        self.cx_rew_end[:, dead] = cx
        self.ep_len[dead] = 0

    @torch.no_grad()
    def step(self, act: Tensor):
        self.act_buffer[:, -1] = act
        next_obs, denoising_trajectory = self.predict_next_obs()
        rew, end = self.predict_rew_end(next_obs.unsqueeze(1))

        self.ep_len += 1
        trunc = (self.ep_len >= self.horizon).long()
        self.obs_buffer = self.obs_buffer.roll(-1, dims=1)
        self.act_buffer = self.act_buffer.roll(-1, dims=1)
        self.obs_buffer[:, -1] = next_obs
        dead = torch.logical_or(end, trunc)

        info: Dict[str, Any] = {}
        if self.return_denoising_trajectory:
            info["denoising_trajectory"] = torch.stack(denoising_trajectory, dim=1)
        if dead.any():
            self.reset_dead(dead)
            info["final_observation"] = next_obs[dead]
            info["burnin_obs"] = self.obs_buffer[dead, :-1]
        return self.obs_buffer[:, -1], rew, end, trunc, info

    @torch.no_grad()
    def predict_next_obs(self) -> Tuple[Tensor, List[Tensor]]:
        return self.sampler.sample(self.obs_buffer, self.act_buffer)

    @torch.no_grad()
    def predict_rew_end(self, next_obs: Tensor) -> Tuple[Tensor, Tensor]:
        logits_rew, logits_end, (self.hx_rew_end, self.cx_rew_end) = self.rew_end_model.predict_rew_end(
            self.obs_buffer[:, -1:], self.act_buffer[:, -1:], next_obs, (self.hx_rew_end, self.cx_rew_end))
        e_rew = None if self.expo_fn is None else self.expo_fn(logits_rew)
        e_end = None if self.expo_fn is None else self.expo_fn(logits_end)
        rew = sample_categorical(logits_rew, e_rew).squeeze(1) - 1.0  # {-1, 0, 1}
        end = sample_categorical(logits_end, e_end).squeeze(1)
        return rew, end

    @coroutine
    def make_generator_init(self, data_loader, num_batches_to_preload: int) -> Generator:
        """Pool of initial conditions (reference :107-139): preload batches, burn the rew/end
        LSTM in on their first T-1 transitions, then serve `num_dead` samples per request."""
        num_dead = yield
        data_iterator = iter(data_loader)
        while True:
            obs_, act_, hx_, cx_ = [], [], [], []
            for _ in range(num_batches_to_preload):
                batch = next(data_iterator)
                obs = batch.obs.to(self.device, non_blocking=True)  # async when the loader pins its batches
                act = batch.act.to(self.device, non_blocking=True)
                *_, (hx, cx) = self.rew_end_model.predict_rew_end(obs[:, :-1], act[:, :-1], obs[:, 1:])
                assert hx.size(0) == cx.size(0) == 1
                obs_.append(obs)
                act_.append(act)
                hx_.append(hx[0])
                cx_.append(cx[0])
            # one device-resident pool tensor per field (the reference keeps python lists of rows)
            obs_p, act_p, hx_p, cx_p = torch.cat(obs_), torch.cat(act_), torch.cat(hx_), torch.cat(cx_)
            c = 0
            while c + num_dead <= obs_p.size(0):
                sl = slice(c, c + num_dead)
                c += num_dead
                num_dead = yield obs_p[sl].clone(), act_p[sl].clone(), (hx_p[sl].unsqueeze(0).clone(), cx_p[sl].unsqueeze(0).clone())
