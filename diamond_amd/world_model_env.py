"""Batched imagination environment with a ring-indexed context and a device-resident uint8 pool of
initial conditions.

Contract = the reference's `WorldModelEnv` (envs/world_model_env.py:25-139): constructor, `num_envs`,
`horizon`, `device`, `sampler`, `reset(**kw) -> (obs, {})`, `step(act) -> (obs, rew, end, trunc, info)` with
`info["final_observation"]` / `info["burnin_obs"]` / `info["denoising_trajectory"]`, and the two re-assignable
callables `predict_next_obs()` / `predict_rew_end(next_obs)` (trainer.py:182-184 overwrites them).  The data
structures behind it are designed for the GPU instead of transcribed:

* context ring -- the T = 4 conditioning frames and actions of every env live in ONE (B, T, C, H, W) fp32 /
  (B, T) int64 buffer that is never rolled (the reference copies the whole buffer every step, :74-75).  Logical
  step t sits at physical slot (head + t) % T; a step overwrites the oldest slot and advances `head`.  The ring
  order is resolved inside `dmd_edm_pack_input` / `dmd_cond_embed` (C ABI: `T, head` arguments).
* initial-condition pool -- preloaded batches are quantised back to the uint8 levels they were decoded from
  (data/episode.py:36-50: `x.div(255).mul(2).sub(1)`) by `dmd_quantize_u8` and stay on the device as uint8
  (4x smaller than the reference's python lists of fp32 rows, :128-131); a reset gathers + dequantises the rows it
  needs straight into the ring slots of the dead envs (`dmd_dequant_gather`).  A loader that yields frames off the
  256-level grid (detected by the quantiser) keeps an fp32 pool: results never change.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple

import torch
from torch import Tensor

from . import native as nv
from .diffusion_sampler import DiffusionSampler, DiffusionSamplerConfig
from .env_loop import sample_categorical


GRAPH_SAMPLER_MAX_ENVS = 8  # below this the sampler is launch-latency-bound and runs as a replayed hipGraph


@dataclass
class WorldModelEnvConfig:
    horizon: int
    num_batches_to_preload: int
    diffusion_sampler: DiffusionSamplerConfig


class InitialConditionPool:
    """Device-resident queue of (context frames, context actions, burnt-in reward/end LSTM state).

    Same serving rule as the reference generator (:133-139): requests are served in order from the preloaded
    batches; when fewer rows are left than a request needs, the remainder is dropped and fresh batches are
    preloaded (each one burns the reward/end LSTM in on its first T-1 transitions, :123-124)."""

    _warned_fp32 = False  # one warning per process when a pool falls back to fp32 (off-grid frames)

    def __init__(self, rew_end_model, data_loader, num_batches: int, device_fn: Callable[[], torch.device]) -> None:
        self._model = rew_end_model
        self._loader = data_loader
        self._iter: Optional[Iterator] = None
        self._num_batches = num_batches
        self._device_fn = device_fn
        self.frames_u8: Optional[Tensor] = None   # (P, T, C, H, W) uint8, or
        self.frames_f32: Optional[Tensor] = None  # (P, T, C, H, W) fp32 when the loader's frames are off the uint8 grid
        self.act: Optional[Tensor] = None         # (P, T) int64
        self.hx: Optional[Tensor] = None          # (P, lstm_dim)
        self.cx: Optional[Tensor] = None
        self._cursor = 0

    @property
    def size(self) -> int:
        return 0 if self.act is None else self.act.shape[0]

    @torch.no_grad()
    def _preload(self) -> None:
        if self._iter is None:
            self._iter = iter(self._loader)
        dev = self._device_fn()
        q_, f_, act_, hx_, cx_ = [], [], [], [], []
        off_grid = torch.zeros(1, dtype=torch.int32, device=dev)
        for _ in range(self._num_batches):
            batch = next(self._iter)
            obs = batch.obs.to(dev, non_blocking=True).float().contiguous()  # async when the loader pins its batches
            act = batch.act.to(dev, non_blocking=True)
            *_, (hx, cx) = self._model.predict_rew_end(obs[:, :-1], act[:, :-1], obs[:, 1:])
            assert hx.size(0) == cx.size(0) == 1
            q = torch.empty(obs.shape, dtype=torch.uint8, device=dev)
            nv.check(nv.lib().dmd_quantize_u8(nv.fptr(obs), nv.ptr(q), nv.ptr(off_grid), obs.numel(), nv.stream()),
                     "dmd_quantize_u8")
            q_.append(q)
            f_.append(obs)
            act_.append(act)
            hx_.append(hx[0])
            cx_.append(cx[0])
        if int(off_grid.item()) == 0:  # one sync per preload round
            self.frames_u8, self.frames_f32 = torch.cat(q_), None
        else:
            if not InitialConditionPool._warned_fp32:
                InitialConditionPool._warned_fp32 = True
                import warnings

                warnings.warn(f"initial-condition pool: {int(off_grid.item())} preloaded values are not on the uint8 grid "
                              "(x = k / 255 * 2 - 1): keeping this pool in fp32 (4x the memory; results are unchanged)")
            self.frames_u8, self.frames_f32 = None, torch.cat(f_)
        self.act, self.hx, self.cx = torch.cat(act_), torch.cat(hx_), torch.cat(cx_)
        self._cursor = 0

    def take(self, count: int) -> Tensor:
        """Device index vector of the next `count` pool rows (preloading when the pool runs short)."""
        if self.size == 0:
            self._preload()
        while self._cursor + count > self.size:
            assert count <= self._num_batches * self._loader.batch_sampler.batch_size, \
                "more simultaneous resets than one preload round holds"
            self._preload()
        idx = torch.arange(self._cursor, self._cursor + count, device=self.act.device)
        self._cursor += count
        return idx

    def scatter_frames(self, idx: Tensor, rows: Optional[Tensor], ring: Tensor, head: int) -> None:
        """ring[rows[i]] <- pool frames idx[i] in logical order (slot (head + t) % T); rows None = all rows."""
        b, t = ring.shape[:2]
        per_frame = ring[0, 0].numel()
        if self.frames_u8 is not None:
            nv.check(nv.lib().dmd_dequant_gather(nv.ptr(self.frames_u8), nv.ptr(idx), nv.ptr(rows), nv.fptr(ring),
                                                 idx.numel(), t, per_frame, head, nv.stream()), "dmd_dequant_gather")
        else:  # fp32 pool (off-grid frames): plain indexed copy
            cols = (head + torch.arange(t, device=ring.device)) % t
            r = rows if rows is not None else torch.arange(b, device=ring.device)
            ring[r[:, None], cols[None, :]] = self.frames_f32[idx]


class WorldModelEnv:
    def __init__(self, denoiser, rew_end_model, data_loader, cfg: WorldModelEnvConfig,
                 return_denoising_trajectory: bool = False, graph_sampler: Optional[bool] = None) -> None:
        """graph_sampler: replay the diffusion sampler as a captured hipGraph (latency mode).  None = automatic: on for
        at most GRAPH_SAMPLER_MAX_ENVS envs (play.py's interactive env runs ONE), overridable with DIAMOND_GRAPH_SAMPLER=0/1."""
        self.sampler = DiffusionSampler(denoiser, cfg.diffusion_sampler)
        self.rew_end_model = rew_end_model
        self.horizon = cfg.horizon
        self.return_denoising_trajectory = return_denoising_trajectory
        self.num_envs = data_loader.batch_sampler.batch_size
        if graph_sampler is None:
            env = os.environ.get("DIAMOND_GRAPH_SAMPLER")
            graph_sampler = (self.num_envs <= GRAPH_SAMPLER_MAX_ENVS) if env is None else env == "1"
        self.graph_sampler = bool(graph_sampler)
        self.pool = InitialConditionPool(rew_end_model, data_loader, cfg.num_batches_to_preload, lambda: self.device)
        self._ctx: Optional[Tensor] = None  # (B, T, C, H, W) fp32 context ring
        self._act: Optional[Tensor] = None  # (B, T) int64 action ring
        self._head = 0                      # physical slot of logical step 0 (both rings advance together)
        # test hook: injected exponential draws for the reward / end samples (host RNG parity)
        self.expo_fn: Optional[Callable[[Tensor], Tensor]] = None
        # speculative step_begin (env_loop issues step n + 1's sampler before step n's host synchronisation, see
        # step_end_issue): the draws of a dropped speculation, re-used by its repetition; how long not to speculate after a
        # step in which an episode ended (doubles with every wasted speculation, back to 4 after 16 that were used)
        self._flag_host: Optional[Tensor] = None
        self._flag_event = None
        self._reset_speculation()

    def _reset_speculation(self) -> None:
        self._pending = None
        self._pending_speculative = False
        self._saved_draws: Optional[Tuple[Optional[Tensor], Tensor, Tensor]] = None
        self._issued = None
        self._spec_cooldown, self._spec_penalty, self._spec_streak = 0, 4, 0

    @property
    def device(self) -> torch.device:
        return self.sampler.denoiser.device

    # -- ring bookkeeping ------------------------------------------------------------------------
    def _slot(self, t: int) -> int:
        """Physical slot of logical step t (negative t counts from the newest step)."""
        steps = self._ctx.shape[1]
        return (self._head + (t % steps)) % steps

    def _cols(self) -> Tensor:
        steps = self._ctx.shape[1]
        return (self._head + torch.arange(steps, device=self._ctx.device)) % steps

    @property
    def obs_buffer(self) -> Tensor:
        """The context in the reference's logical layout (B, T, C, H, W) -- a COPY, for inspection."""
        return self._ctx[:, self._cols()]

    @property
    def act_buffer(self) -> Tensor:
        return self._act[:, self._cols()]

    # -- gym-style API ---------------------------------------------------------------------------
    @torch.no_grad()
    def reset(self, **kwargs) -> Tuple[Tensor, Dict[str, Any]]:
        idx = self.pool.take(self.num_envs)
        dev = idx.device
        frames = self.pool.frames_u8 if self.pool.frames_u8 is not None else self.pool.frames_f32
        shape = (self.num_envs,) + tuple(frames.shape[1:])
        if self._ctx is None or tuple(self._ctx.shape) != shape or self._ctx.device != dev:
            # the rings are allocated once and reused by later resets: captured sampler graphs stay valid
            self._ctx = torch.empty(shape, dtype=torch.float32, device=dev)
            self._act = torch.empty(shape[:2], dtype=torch.long, device=dev)
        self._head = 0
        self._reset_speculation()
        self.pool.scatter_frames(idx, None, self._ctx, 0)
        self._act.copy_(self.pool.act[idx])
        self.hx_rew_end = self.pool.hx[idx].unsqueeze(0).clone()
        self.cx_rew_end = self.pool.cx[idx].unsqueeze(0).clone()
        self.ep_len = torch.zeros(self.num_envs, dtype=torch.long, device=dev)
        return self._ctx[:, self._slot(-1)].clone(), {}  # a copy: ring slots are overwritten T steps later

    @torch.no_grad()
    def reset_dead(self, dead: Tensor) -> Tensor:
        """Replace the context / action ring / reward-end LSTM state of the dead envs by fresh pool rows;
        returns the dead row indices.  INVARIANT (env_loop's speculation relies on it): no draw from torch's random
        generators happens here or in the pool's preload -- a reset consumes no random stream."""
        rows = dead.nonzero(as_tuple=True)[0]
        idx = self.pool.take(int(rows.numel()))
        self.pool.scatter_frames(idx, rows, self._ctx, self._head)
        self._act[rows[:, None], self._cols()[None, :]] = self.pool.act[idx]
        self.hx_rew_end[0, rows] = self.pool.hx[idx]
        self.cx_rew_end[0, rows] = self.pool.cx[idx]
        self.ep_len[rows] = 0
        return rows

    @torch.no_grad()
    def step(self, act: Tensor):
        self.step_begin(act)
        return self.step_end()

    @torch.no_grad()
    def step_begin(self, act: Tensor, speculative: bool = False) -> Tensor:
        """First half of `step`: the imagined next frame.  Nothing here waits for the device.  The random draws of the step
        (initial noise of the sampler, reward / end samples) are made NOW, in the reference's order, so that a caller may
        interleave its own draws between the two halves without changing the order in which the streams are consumed
        (env_loop issues the policy's next step in between).
        speculative: issued BEFORE the previous step's host synchronisation (between step_end_issue and step_end_finish), on
        the assumption that no episode ended there.  If one did, step_end_finish drops this half-step, keeps its draws, and
        the caller's next step_begin -- with the action recomputed after the reset -- consumes them: every random stream is
        consumed in the same order either way (no draw is made by a reset: `reset_dead` and the pool preload use none)."""
        newest = self._slot(-1)
        self._act[:, newest] = act
        saved, self._saved_draws = self._saved_draws, None
        own_noise = not self._use_graph()  # (a captured sampler graph draws inside the graph: never speculated, see may_speculate)
        if saved is not None:
            noise, e_rew, e_end = saved
        else:
            b, dev = self._ctx.shape[0], self._ctx.device
            noise = self.sampler._randn((b,) + tuple(self._ctx.shape[2:]), dev) if own_noise else None
            if self.expo_fn is None:
                e_rew = torch.empty(b, 3, device=dev).exponential_(1)
                e_end = torch.empty(b, 2, device=dev).exponential_(1)
            else:  # (test hook: draws injected in the reference's order -- the hook looks at the shape only)
                e_rew = self.expo_fn(torch.empty(b, 1, 3, device=dev))
                e_end = self.expo_fn(torch.empty(b, 1, 2, device=dev))
        self._next_noise = noise  # (handed over out of band: predict_next_obs keeps the reference's zero-argument signature,
        next_obs, denoising_trajectory = self.predict_next_obs()  # trainer.py:182-184 re-assigns it with a wrapper)
        self._pending = (next_obs, denoising_trajectory, e_rew, e_end, noise)
        self._pending_speculative = speculative
        return next_obs

    def may_speculate(self) -> bool:
        """May the caller issue the NEXT step's step_begin before this step's step_end_finish?  Not while the cool-down after an
        ended episode runs (a dropped speculation costs a whole sampler step), not with a captured sampler graph (its noise is
        drawn inside the graph) or stochastic churn (more draws inside the sampler than this class keeps)."""
        return self._spec_cooldown == 0 and not self._use_graph() and self.sampler.cfg.s_churn == 0

    @torch.no_grad()
    def step_end(self):
        """Second half of `step`: reward / termination, ring bookkeeping and -- the one host synchronisation of a step -- the
        check for finished episodes (`if dead.any()`, world_model_env.py:77-83)."""
        self.step_end_issue()
        return self.step_end_finish()

    @torch.no_grad()
    def step_end_issue(self) -> None:
        """Everything of step_end that the host can issue without knowing whether an episode ended; the answer travels to the
        host asynchronously (pinned flag + event).  The caller may issue more work -- the policy's and the sampler's next step
        -- before it asks for it with step_end_finish: the device then never runs dry while the host waits."""
        next_obs, denoising_trajectory, e_rew, e_end, _ = self._pending
        self._pending, self._pending_speculative = None, False
        rew, end = self.predict_rew_end(next_obs.unsqueeze(1), e_rew, e_end)

        self.ep_len += 1
        trunc = (self.ep_len >= self.horizon).long()
        # advance the rings: the oldest slot becomes the newest and receives the imagined frame (its action slot is
        # written by the next step, exactly when the reference writes act_buffer[:, -1])
        oldest = self._head
        self._head = (self._head + 1) % self._ctx.shape[1]
        self._ctx[:, oldest] = next_obs
        dead = torch.logical_or(end, trunc)

        info: Dict[str, Any] = {}
        if self.return_denoising_trajectory:
            info["denoising_trajectory"] = torch.stack(denoising_trajectory, dim=1)
        flag = dead.any()
        if flag.is_cuda:
            if getattr(self, "_flag_host", None) is None:
                self._flag_host = torch.zeros((), dtype=torch.bool).pin_memory()
                self._flag_event = torch.cuda.Event()
            self._flag_host.copy_(flag, non_blocking=True)
            self._flag_event.record()
        self._issued = (next_obs, rew, end, trunc, dead, info, flag)

    @torch.no_grad()
    def step_end_finish(self):
        """THE host synchronisation of a step: did an episode end?  If so: resets, and a speculative step_begin issued
        meanwhile is dropped (its draws are kept for the repetition)."""
        next_obs, rew, end, trunc, dead, info, flag = self._issued
        self._issued = None
        if flag.is_cuda:
            self._flag_event.synchronize()
            any_dead = bool(self._flag_host)
        else:
            any_dead = bool(flag)
        obs = next_obs  # a fresh tensor every step: never aliases the ring
        info["any_dead"] = any_dead  # (so that the caller does not have to synchronise again for the same answer)
        if self._spec_cooldown > 0:
            self._spec_cooldown -= 1
        if self._pending is not None and self._pending_speculative:
            if any_dead:  # wasted: remember the draws, back off for longer
                self._saved_draws = (self._pending[4], self._pending[2], self._pending[3])
                self._pending, self._pending_speculative = None, False
                self._spec_penalty, self._spec_streak = min(64, 2 * self._spec_penalty), 0
            else:
                self._spec_streak += 1
                if self._spec_streak >= 16:
                    self._spec_penalty = 4
        if any_dead:
            self._spec_cooldown = self._spec_penalty
            rows = self.reset_dead(dead)
            info["dead_rows"] = rows  # device index list: the caller gathers / scatters with it (no further synchronisation)
            info["final_observation"] = next_obs[rows]
            cols = self._cols()
            info["burnin_obs"] = self._ctx[rows[:, None], cols[None, :-1]]
            obs = self._ctx[:, self._slot(-1)].clone()  # dead envs now show the newest frame of their new episode
        return obs, rew, end, trunc, info

    def _use_graph(self) -> bool:
        # (no replay while a launch profiler is installed: a replayed graph issues no launches it could time, and a first
        #  capture inside the profiled window would record timing events into the graph)
        return self.graph_sampler and self.sampler.noise_fn is None and nv.PROFILER is None

    @torch.no_grad()
    def predict_next_obs(self) -> Tuple[Tensor, List[Tensor]]:
        """Reference signature (world_model_env.py:91-93).  The initial noise step_begin drew for this call, if any, waits in
        `_next_noise`; a direct call draws its own."""
        noise, self._next_noise = getattr(self, "_next_noise", None), None
        if self._use_graph():
            return self.sampler.sample_ring_graphed(self._ctx, self._act, self._head, self._head)
        return self.sampler.sample_ring(self._ctx, self._act, self._head, self._head, noise)

    @torch.no_grad()
    def predict_rew_end(self, next_obs: Tensor, e_rew: Optional[Tensor] = None, e_end: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """e_rew / e_end: exponential draws made earlier (step_begin); None: drawn here."""
        newest = self._slot(-1)
        logits_rew, logits_end, (self.hx_rew_end, self.cx_rew_end) = self.rew_end_model.predict_rew_end(
            self._ctx[:, newest:newest + 1], self._act[:, newest:newest + 1], next_obs, (self.hx_rew_end, self.cx_rew_end))
        if e_rew is None:
            e_rew = None if self.expo_fn is None else self.expo_fn(logits_rew)
            e_end = None if self.expo_fn is None else self.expo_fn(logits_end)
        rew = sample_categorical(logits_rew, e_rew).squeeze(1) - 1.0  # {-1, 0, 1}
        end = sample_categorical(logits_end, e_end).squeeze(1)
        return rew, end
