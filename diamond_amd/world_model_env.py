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

import ctypes as C
import os
import time
from dataclasses import dataclass
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import native as nv
from .engine import check_weight_audits
from .diffusion_sampler import DiffusionSampler, DiffusionSamplerConfig
from .env_loop import sample_categorical


class _no_random_draws:
    """Guard of the invariant the slots env loop relies on: a reset (pool preload / prefetch included) draws from NO torch
    generator, so resolving resets on the device, prefetching pool rounds and repeating a window from its snapshot cannot change
    the order in which the random streams are consumed.  With
    DIAMOND_CHECK_RESET_RNG=1 (the test suites set it) the CPU and the device generator states are compared around the guarded
    block -- a loader or pool that draws from them fails loudly instead of silently reordering the streams."""

    def __init__(self, dev: torch.device, what: str, cpu: bool = True) -> None:
        # cpu=False: only the device generator is watched (a DataLoader seeds its workers from the CPU default generator; that
        # is harmless where the env's own draws come from the device generator, i.e. everywhere but in the hook-driven tests)
        self.dev, self.what, self.cpu = dev, what, cpu or dev.type != "cuda"
        self.on = os.environ.get("DIAMOND_CHECK_RESET_RNG") == "1"

    def _state(self):
        st = [torch.get_rng_state()] if self.cpu else []
        if self.dev.type == "cuda":
            st.append(torch.cuda.get_rng_state(self.dev))
        return st

    def __enter__(self):
        if self.on:
            self.before = self._state()

    def __exit__(self, *exc):
        if self.on and exc[0] is None:
            assert all(torch.equal(a, b) for a, b in zip(self.before, self._state())), \
                f"{self.what} drew from a torch random generator: the slots env loop (env_loop.py) requires resets to be RNG-free " \
                "(DIAMOND_ENV_LOOP=sequential runs the reference's order of calls)"
        return False


def _without_torch_compile(fn, default):
    """`fn` unless it is a torch.compile(...) wrapper (dynamo marks those with `_torchdynamo_orig_callable`): then what it wraps --
    `default` (the env's own bound method) when that is the function inside, as in trainer.py:182-184."""
    orig = getattr(fn, "_torchdynamo_orig_callable", None)
    if orig is None:
        return fn
    if orig is getattr(default, "__func__", None):
        return default  # (only the method's function survived the wrapping)
    return orig


class SlotOverflow(RuntimeError):
    """More episodes ended in a step than the step had reset slots for (step_end_slots): the rows beyond the last slot were not
    reset.  env_loop restores the window's snapshot and repeats the window with a slot for every env."""


class ResetSlots:
    """The deaths of one step as the device resolved them (dmd_resolve_deaths): K slots, the j-th dead row in slot j (ascending row
    order: the order of the reference's boolean masks and of its pool), -1 for unused slots.  Everything the policy does for a
    reset (reference env_loop.py:45-56) is fixed-shape over the slots."""

    def __init__(self, k: int, slot_row: Tensor, row_slot: Tensor, dead: Tensor) -> None:
        self.K, self.slot_row, self.row_slot, self.dead = k, slot_row, row_slot, dead
        self._gather: Optional[Tensor] = None

    @property
    def gather_rows(self) -> Tensor:
        """(K,) row index per slot for GATHERS (unused slots read row 0: finite values nobody uses)"""
        if self._gather is None:
            self._gather = self.slot_row.clamp_min(0)
        return self._gather

    def merge(self, base: Tensor, values: Tensor) -> Tensor:
        """base (B, ...) with the rows of the used slots replaced by values (K, ...); differentiable in both."""
        from .lstm_native import merge_slots

        return merge_slots(base, values, self.row_slot, self.slot_row)


def _poisson_quantile(mean: float, tail: float = 1e-7) -> int:
    """smallest x with P(X > x) < tail for X ~ Poisson(mean)"""
    import math

    if mean <= 0:
        return 0
    if mean > 50:  # normal approximation with a continuity margin
        return int(math.ceil(mean + 5.5 * math.sqrt(mean) + 1))
    term = math.exp(-mean)
    cdf, x = term, 0
    while 1.0 - cdf >= tail and x < 1000:
        x += 1
        term *= mean / x
        cdf += term
    return x


GRAPH_SAMPLER_MAX_ENVS = 8  # below this the sampler is launch-latency-bound and runs as a replayed hipGraph ...
GRAPH_SAMPLER_MAX_PIXELS = 8 * 64 * 64  # ... if its launches are small: 8 envs at 256x256 are not (configs[4]: eager measured
#                                         367-370 frames/s against 364-365 replayed, same box, alternating: round 4)


@dataclass
class WorldModelEnvConfig:
    horizon: int
    num_batches_to_preload: int
    diffusion_sampler: DiffusionSamplerConfig


class _Round:
    """One preloaded round of initial conditions.  Whether it is kept as uint8 levels or as fp32 frames depends on three device
    counters (values off the 256-level grid, padded frames, padded frames that are not zero): they travel to a pinned buffer
    behind the round's kernels and are read when the round is first USED -- a prefetched round costs no host wait."""

    def __init__(self, q_, f_, act: Tensor, hx: Tensor, cx: Tensor, pad_, counters: Tensor) -> None:
        self._q, self._f, self.act, self.hx, self.cx, self._pad = q_, f_, act, hx, cx, pad_
        self._fields = None
        if counters.is_cuda:
            self._host = torch.zeros(3, dtype=torch.int64).pin_memory()
            self._host.copy_(counters, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._host, self._event = counters.clone(), None

    def fields(self):
        """(frames_u8, frames_f32, act, hx, cx, pad) -- decided once"""
        if self._fields is None:
            if self._event is not None:
                self._event.synchronize()
            n_off, n_pad, n_pad_nonzero = self._host.tolist()
            pad = None
            if n_off == 0 and n_pad_nonzero == 0:
                frames_u8, frames_f32 = torch.cat(self._q), None
                if n_pad:
                    pad = torch.cat(self._pad)
            else:
                if not InitialConditionPool._warned_fp32:
                    InitialConditionPool._warned_fp32 = True
                    import warnings

                    warnings.warn(f"initial-condition pool: {n_off + n_pad_nonzero} preloaded values are not on the uint8 grid "
                                  "(x = k / 255 * 2 - 1): keeping this pool in fp32 (4x the memory; results are unchanged)")
                frames_u8, frames_f32 = None, torch.cat(self._f)
            self._fields = (frames_u8, frames_f32, self.act, self.hx, self.cx, pad)
            self._q = self._f = self._pad = None
        return self._fields


class InitialConditionPool:
    """Device-resident queue of (context frames, context actions, burnt-in reward/end LSTM state).

    Same serving rule as the reference generator (:133-139): requests are served in order from the preloaded
    batches; when fewer rows are left than a request needs, the remainder is dropped and fresh batches are
    preloaded (each one burns the reward/end LSTM in on its first T-1 transitions, :123-124)."""

    _warned_fp32 = False  # one warning per process when a pool falls back to fp32 (off-grid frames)
    PAD_AWARE = True  # (False: the loader's zero-padded frames count as off-grid values, as before -- A/B and the tests' reference arm)

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
        self._watch_cpu_rng = False  # (WorldModelEnv sets it while draws are injected from the CPU generator: tests)
        self._generation = 0  # counts preload rounds (a peeked plan is void once the pool it pointed into was replaced)
        # (P, T) bool, True = a PADDED frame of the uint8 pool, or None when the pool holds none.  The reference's segments are
        # zero-padded in front of an episode's first step (data/utils.py:18-41, data/batch_sampler.py:63-68: "padding allowed only
        # before start") and WorldModelEnv uses those zero frames as they are (world_model_env.py:116-131 ignores mask_padding);
        # 0.0 is level 127.5, i.e. off the uint8 grid -- with real data nearly every preload round holds a few, and the whole pool
        # would fall back to fp32.  Frames the loader itself marks as padding (batch.mask_padding False) and that ARE zero are
        # therefore stored as a stand-in level and put back as exact zeros behind the dequantising gather.
        self.pad: Optional[Tensor] = None
        self._next: Optional[_Round] = None  # the prefetched next round (prefetch)
        self._copy_stream = None

    @property
    def size(self) -> int:
        return 0 if self.act is None else self.act.shape[0]

    _FIELDS = ("frames_u8", "frames_f32", "act", "hx", "cx", "pad", "_cursor", "_generation", "_next")

    def snapshot(self):
        """What a window's repetition needs to see the same pool again (env_loop repeats a window after a SlotOverflow): the
        current round, the cursor and the prefetched round; rounds loaded from now on are remembered and served again, in order,
        after `restore` -- the loader's iterator cannot be rewound."""
        self._recorded: List["_Round"] = []
        return tuple(getattr(self, f) for f in self._FIELDS)

    def restore(self, snap) -> None:
        for f, v in zip(self._FIELDS, snap):
            setattr(self, f, v)
        self._replay = list(getattr(self, "_recorded", [])) + list(getattr(self, "_replay", []))
        self._recorded = []

    def _next_round(self) -> "_Round":
        replay = getattr(self, "_replay", None)
        r = replay.pop(0) if replay else self._load_round()  # (a round a repeated window already loaded once)
        if hasattr(self, "_recorded"):
            self._recorded.append(r)
        return r

    def prefetch(self) -> None:
        """Load the NEXT round now (loader batches, upload, reward/end burn-in, quantisation: all asynchronous) without touching
        the current one.  The rounds are what the reference's generator would preload, in its order; only the moment differs."""
        if self._next is None:
            self._next = self._next_round()

    @torch.no_grad()
    def _preload(self) -> None:
        """The current round is replaced by the next one (reference :133-139: the remainder is dropped)."""
        r, self._next = (self._next if self._next is not None else self._next_round()), None
        self.frames_u8, self.frames_f32, self.act, self.hx, self.cx, self.pad = r.fields()
        self._cursor = 0
        self._generation += 1

    @torch.no_grad()
    def _load_round(self) -> "_Round":
        if self._iter is None:
            self._iter = iter(self._loader)
        dev = self._device_fn()
        q_, f_, act_, hx_, cx_, pad_ = [], [], [], [], [], []
        off_grid = torch.zeros(1, dtype=torch.int32, device=dev)
        pad_count = torch.zeros(1, dtype=torch.int64, device=dev)  # padded frames / values in them that are not zero
        pad_nonzero = torch.zeros(1, dtype=torch.int64, device=dev)
        side = None
        if dev.type == "cuda":  # uploads on a stream of their own: the copy engine works while the compute queue drains
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=dev)
            side = self._copy_stream
        for _ in range(self._num_batches):
            with _no_random_draws(dev, "the initial-condition loader", cpu=self._watch_cpu_rng):
                batch = next(self._iter)
            mask = getattr(batch, "mask_padding", None) if self.PAD_AWARE else None
            if side is not None:
                with torch.cuda.stream(side):
                    obs_up = batch.obs.to(dev, non_blocking=True)  # async when the loader pins its batches
                    act_up = batch.act.to(dev, non_blocking=True)
                    mask_up = None if mask is None else mask.to(dev, non_blocking=True)
                torch.cuda.current_stream().wait_stream(side)
                for t_ in (obs_up, act_up, mask_up):
                    if t_ is not None:
                        t_.record_stream(torch.cuda.current_stream())
            else:
                obs_up, act_up, mask_up = batch.obs.to(dev), batch.act.to(dev), None if mask is None else mask.to(dev)
            obs = obs_up.float().contiguous()
            act = act_up.long()  # (a loader may yield int32 / uint8 actions: the rings are int64)
            *_, (hx, cx) = self._model.predict_rew_end(obs[:, :-1], act[:, :-1], obs[:, 1:])
            assert hx.size(0) == cx.size(0) == 1
            pad = None if mask_up is None else ~mask_up.bool()
            src = obs
            if pad is not None:  # (the stand-in is level 0; what the frames really hold is checked when the round is first used)
                where = pad[:, :, None, None, None]
                pad_count += pad.sum()
                pad_nonzero += (obs.masked_fill(~where, 0.0) != 0).sum()
                src = obs.masked_fill(where, -1.0)
            q = torch.empty(obs.shape, dtype=torch.uint8, device=dev)
            nv.check(nv.lib().dmd_quantize_u8(nv.fptr(src), nv.ptr(q), nv.ptr(off_grid), obs.numel(), nv.stream()),
                     "dmd_quantize_u8")
            q_.append(q)
            f_.append(obs)
            act_.append(act)
            hx_.append(hx[0])
            cx_.append(cx[0])
            pad_.append(pad if pad is not None else torch.zeros(obs.shape[:2], dtype=torch.bool, device=dev))
        return _Round(q_, f_, torch.cat(act_), torch.cat(hx_), torch.cat(cx_), pad_, torch.cat([off_grid.long(), pad_count, pad_nonzero]))

    def peek(self, count: int) -> Tuple[Tensor, Tuple[int, int]]:
        """Device index vector of the next `count` pool rows WITHOUT serving them (preloading when the pool runs short, exactly
        as `take` would for this count), and a token (pool generation, cursor) for `commit`."""
        start, token = self.peek_start(count)
        return torch.arange(start, start + count, device=self.act.device), token

    def peek_start(self, count: int) -> Tuple[int, Tuple[int, int]]:
        """`peek` for a caller that builds the index vector itself (on the host, next to its row list: one upload for both)."""
        if self.size == 0:
            self._preload()
        while self._cursor + count > self.size:
            assert count <= self._num_batches * self._loader.batch_sampler.batch_size, \
                "more simultaneous resets than one preload round holds"
            self._preload()
        return self._cursor, (self._generation, self._cursor)

    def commit(self, token: Tuple[int, int], count: int) -> None:
        assert token == (self._generation, self._cursor), "pool rows were served between peek and commit"
        self._cursor += count

    def take(self, count: int) -> Tensor:
        """Device index vector of the next `count` pool rows (preloading when the pool runs short)."""
        idx, token = self.peek(count)
        self.commit(token, count)
        return idx

    def gather_frames(self, idx: Tensor) -> Tensor:
        """(len(idx), T, C, H, W) fp32 copies of pool rows in logical order."""
        src = self.frames_u8 if self.frames_u8 is not None else self.frames_f32
        out = torch.empty((idx.numel(),) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
        self.scatter_frames(idx, None, out, 0)
        return out

    def scatter_frames(self, idx: Tensor, rows: Optional[Tensor], ring: Tensor, head: int) -> None:
        """ring[rows[i]] <- pool frames idx[i] in logical order (slot (head + t) % T); rows None = all rows."""
        b, t = ring.shape[:2]
        per_frame = ring[0, 0].numel()
        if self.frames_u8 is not None:
            nv.check(nv.lib().dmd_dequant_gather(nv.ptr(self.frames_u8), nv.ptr(idx), nv.ptr(rows), nv.fptr(ring),
                                                 idx.numel(), t, per_frame, head, nv.stream()), "dmd_dequant_gather")
            if self.pad is not None:  # the loader's zero-padded frames, stored as a stand-in level: exact zeros again
                cols = (head + torch.arange(t, device=ring.device)) % t
                r = rows if rows is not None else torch.arange(idx.numel(), device=ring.device)
                at = (r[:, None], cols[None, :])
                ring[at] = ring[at].masked_fill_(self.pad.index_select(0, idx)[:, :, None, None, None], 0.0)
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
        self._graph_forced = graph_sampler is not None or os.environ.get("DIAMOND_GRAPH_SAMPLER") is not None
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
        self._dead_host: Optional[Tensor] = None  # pinned (B,) copy of a step's `dead` mask: THE host synchronisation of a step
        self._flag_event = None
        self._rows_pinned: Optional[Tensor] = None
        self._report_host: Optional[Tensor] = None  # pinned ring of step reports (step_end_slots)
        self._ep_len_host: Optional[np.ndarray] = None  # host mirror of ep_len (the truncations of a step are known ahead: slot_count)
        self._reset_state()

    def _reset_state(self) -> None:
        self._pending = None                              # (imagined frame, trajectory, reward / end draws, noise) between the two halves of a step
        self._slots_inflight = None                       # (event, pinned report, K, pool token, two rounds given) of the last step_end_slots
        self._end_mean = getattr(self, "_end_mean", 0.0)  # running mean of sampled `end`s per step (survives a reset())
        self._end_last = getattr(self, "_end_last", 0)    # ... the last step's count
        self._end_steps = getattr(self, "_end_steps", 0)  # ... and how many steps the mean has seen
        self.stats = {"steps": 0, "steps_with_deaths": 0, "slots": 0, "dead_rows": 0, "slot_overflows": 0, "pool_rounds": 0, "report_wait_ms": 0.0}

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

    def _rows_to_device(self, rows_host) -> Tensor:
        """Host row list -> device index vector through a pinned staging buffer (no synchronisation)."""
        k = len(rows_host)
        dev = self._ctx.device
        if dev.type != "cuda":
            return torch.as_tensor(np.asarray(rows_host, dtype=np.int64), device=dev)
        # (a fresh pinned buffer per call would synchronise in the allocator; the staging buffer is reused only after the copy
        #  that read it was issued on the same stream as everything else, and H2D copies from pinned memory read at execution
        #  time: so each call gets its own slice of a ring of slices)
        if self._rows_pinned is None:
            self._rows_pinned = torch.empty(64, 2 * self.num_envs, dtype=torch.int64).pin_memory()
            self._rows_slot = 0
        self._rows_slot = (self._rows_slot + 1) % self._rows_pinned.shape[0]
        buf = self._rows_pinned[self._rows_slot]
        buf[:k] = torch.as_tensor(np.asarray(rows_host, dtype=np.int64))
        return buf[:k].to(dev, non_blocking=True)

    def set_episode_lengths(self, ep_len: Tensor) -> None:
        """Overwrite ep_len (device counter AND the host mirror the planned resets are derived from)."""
        self.ep_len = ep_len.to(device=self._ctx.device, dtype=torch.long).clone()
        self._ep_len_host = self.ep_len.cpu().numpy().copy()

    # -- gym-style API ---------------------------------------------------------------------------
    @torch.no_grad()
    def reset(self, **kwargs) -> Tuple[Tensor, Dict[str, Any]]:
        nv.check_current_device(self.device)
        idx = self.pool.take(self.num_envs)
        dev = idx.device
        frames = self.pool.frames_u8 if self.pool.frames_u8 is not None else self.pool.frames_f32
        shape = (self.num_envs,) + tuple(frames.shape[1:])
        if self._ctx is None or tuple(self._ctx.shape) != shape or self._ctx.device != dev:
            # the rings are allocated once and reused by later resets: captured sampler graphs stay valid
            self._ctx = torch.empty(shape, dtype=torch.float32, device=dev)
            self._act = torch.empty(shape[:2], dtype=torch.long, device=dev)
        self._head = 0
        self._reset_state()
        self.pool.scatter_frames(idx, None, self._ctx, 0)
        self._act.copy_(self.pool.act[idx])
        self.hx_rew_end = self.pool.hx[idx].unsqueeze(0).clone()
        self.cx_rew_end = self.pool.cx[idx].unsqueeze(0).clone()
        self.ep_len = torch.zeros(self.num_envs, dtype=torch.long, device=dev)
        self._ep_len_host = np.zeros(self.num_envs, dtype=np.int64)
        return self._ctx[:, self._slot(-1)].clone(), {}  # a copy: ring slots are overwritten T steps later

    @torch.no_grad()
    def _reset_rows(self, rows: Tensor, idx: Tensor) -> None:
        """Context / action ring / reward-end LSTM state / episode length of `rows` <- pool rows `idx` (reference reset_dead,
        world_model_env.py:56-62).  INVARIANT (the slots env loop relies on it): no draw from torch's random generators
        happens here or in the pool's preload -- a reset consumes no random stream."""
        with _no_random_draws(self._ctx.device, "WorldModelEnv._reset_rows"):
            self.pool.scatter_frames(idx, rows, self._ctx, self._head)
            hx, cx, pool = self.hx_rew_end, self.cx_rew_end, self.pool
            f32, i64 = torch.float32, torch.long  # (dmd_reset_state reads raw pointers: int64 actions / counters, fp32 states)
            if (pool.frames_u8 is not None and hx.dtype == cx.dtype == pool.hx.dtype == pool.cx.dtype == f32
                    and self._act.dtype == pool.act.dtype == self.ep_len.dtype == i64 and hx.is_contiguous() and cx.is_contiguous()
                    and pool.hx.is_contiguous() and pool.cx.is_contiguous() and self._act.is_contiguous() and pool.act.is_contiguous()):
                # action ring, reward/end LSTM state and episode length of the rows in ONE launch (this sits on the step's
                # critical path, right behind its host synchronisation: the four indexed assignments below are ~12 launches)
                nv.check(nv.lib().dmd_reset_state(nv.ptr(idx), nv.ptr(rows), int(rows.numel()), nv.ptr(pool.act), nv.ptr(self._act),
                                                  self._act.shape[1], self._head, nv.fptr(pool.hx), nv.fptr(pool.cx), nv.fptr(hx), nv.fptr(cx),
                                                  hx.shape[-1], nv.ptr(self.ep_len), nv.stream()), "dmd_reset_state")
            else:  # (the fp32 pool of off-grid frames: plain indexed copies throughout, like its frames in scatter_frames)
                self._act[rows[:, None], self._cols()[None, :]] = pool.act[idx]
                hx[0, rows] = pool.hx[idx]
                cx[0, rows] = pool.cx[idx]
                self.ep_len[rows] = 0

    @torch.no_grad()
    def reset_dead(self, dead: Tensor) -> Tensor:
        """Replace the state of the dead envs by fresh pool rows; returns the dead row indices (synchronises: prefer the
        step_end path, which knows the rows on the host)."""
        rows = dead.nonzero(as_tuple=True)[0]
        self._reset_rows(rows, self.pool.take(int(rows.numel())))
        if self._ep_len_host is not None:
            self._ep_len_host[rows.cpu().numpy()] = 0
        return rows

    @torch.no_grad()
    def step(self, act: Tensor):
        self.step_begin(act)
        return self.step_end()

    def _draw_step(self, own_noise: bool):
        b, dev = self._ctx.shape[0], self._ctx.device
        self.pool._watch_cpu_rng = self.expo_fn is not None or self.sampler.noise_fn is not None
        noise = self.sampler._randn((b,) + tuple(self._ctx.shape[2:]), dev) if own_noise else None
        if self.expo_fn is None:
            e_rew = torch.empty(b, 3, device=dev).exponential_(1)
            e_end = torch.empty(b, 2, device=dev).exponential_(1)
        else:  # (test hook: draws injected in the reference's order -- the hook looks at the shape only)
            e_rew = self.expo_fn(torch.empty(b, 1, 3, device=dev))
            e_end = self.expo_fn(torch.empty(b, 1, 2, device=dev))
        return noise, e_rew, e_end

    @torch.no_grad()
    def step_begin(self, act: Tensor) -> Tensor:
        """First half of `step`: the imagined next frame.  Nothing here waits for the device.  The random draws of the step
        (initial noise of the sampler, reward / end samples) are made NOW, in the reference's order (diffusion_sampler.py:36,
        world_model_env.py:103-104)."""
        assert self._pending is None, "step_begin twice without step_end / step_end_slots"
        newest = self._slot(-1)
        self._act[:, newest] = act
        own_noise = not self._use_graph()  # (a captured sampler graph draws inside the graph)
        noise, e_rew, e_end = self._draw_step(own_noise)
        self._next_noise = noise  # (handed over out of band: predict_next_obs keeps the reference's zero-argument signature,
        next_obs, denoising_trajectory = self.predict_next_obs()  # trainer.py:182-184 re-assigns it with a wrapper)
        self._pending = (next_obs, denoising_trajectory, e_rew, e_end, noise)
        return next_obs

    @torch.no_grad()
    def step_end(self):
        """Second half of `step` in the REFERENCE's order (world_model_env.py:64-89): reward / termination, ring bookkeeping and
        -- the one host synchronisation of such a step -- `if dead.any(): reset_dead`.  info: any_dead, dead, and where an
        episode ended dead_rows (device index vector, ascending), final_observation, burnin_obs (both in dead_rows order, as
        the reference's boolean-mask forms are).  (env_loop's default loop does not come here: step_end_slots below.)"""
        next_obs, denoising_trajectory, e_rew, e_end, _ = self._pending
        self._pending = None
        self.slots_finish()  # (a caller that mixes the two forms: the mirror and the pool cursor are current)
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
        if dead.is_cuda:
            if self._dead_host is None or self._dead_host.numel() != dead.numel():
                self._dead_host = torch.zeros(dead.numel(), dtype=torch.bool).pin_memory()
                self._flag_event = torch.cuda.Event()
            self._dead_host.copy_(dead, non_blocking=True)
            self._flag_event.record()
            self._flag_event.synchronize()  # THE host synchronisation of a step in the reference's order
            rows_host = np.flatnonzero(self._dead_host.numpy())
            check_weight_audits()  # (the host is synchronised anyway: did an audit of the packed weight copies find a silent write?)
        else:
            rows_host = np.flatnonzero(dead.numpy())
        if self._ep_len_host is not None:
            self._ep_len_host += 1
            self._ep_len_host[rows_host] = 0
        total = int(rows_host.size)
        info["any_dead"] = total > 0  # (so that the caller does not have to synchronise again for the same answer)
        info["dead"] = dead
        self.stats["steps"] += 1
        obs = next_obs  # a fresh tensor every step: never aliases the ring
        if total:
            self.stats["steps_with_deaths"] += 1
            self.stats["dead_rows"] += total
            # (the dead rows and their pool rows in ONE upload)
            start, token = self.pool.peek_start(total)
            self.pool.commit(token, total)
            both = self._rows_to_device(np.concatenate([rows_host, np.arange(start, start + total)]))
            rows, fresh_idx = both[:total], both[total:]
            self._reset_rows(rows, fresh_idx)
            fresh = self.pool.gather_frames(fresh_idx)  # (one launch; the same values the ring rows just received)
            info["dead_rows"] = rows  # device index list: the caller gathers / scatters with it (no further synchronisation)
            info["final_observation"] = next_obs.index_select(0, rows)
            info["burnin_obs"] = fresh[:, :-1]
            obs = next_obs.index_copy(0, rows, fresh[:, -1])  # dead envs now show the newest frame of their new episode
        return obs, rew, end, trunc, info

    # -- the step's deaths resolved on the device (env_loop._slots_env_loop) ------------------------------------------------------
    # The reference's one data-dependent branch per step (`if dead.any()`, :77) costs a host round trip, and whatever the host
    # issues behind it starts on an idle device.  Here the branch is taken ON THE DEVICE: a step works on K reset SLOTS, K chosen
    # before its deaths are known -- the truncations the host can count from its mirror of ep_len, plus a margin for sampled `end`s
    # -- dmd_resolve_deaths assigns the dead rows to slots in row order, dmd_reset_slots resets them from the next pool rows and
    # builds the policy's next input, and the host reads the step's report while it issues the NEXT step: always one step behind
    # a device that holds a full step of queued work.  Exact: per row the reference's arithmetic and pool order; no random draw
    # depends on it.  More deaths than slots (SlotOverflow, about once in 1e7 steps by the margin's design) repeats the window.
    # per-step probability the margin for sampled ends accepts of running out of slots: a repeated window per ~10,000 steps
    # (0.15 % of the time) against four encoder frames per step and spare slot (forward and backward)
    DR_END_TAIL = 1e-4
    PREFETCH_STEPS = 6  # the next pool round is loaded when fewer than this many steps' worth of slots is left in the current one

    def slot_count(self, all_slots: bool = False) -> int:
        """Reset slots of the pending step: the truncations it WILL have (host mirror of ep_len, exact) + a Poisson-tail margin for
        the `end`s the reward/end model may sample, from the running mean of ends per step; multiples of 4 (batches of 64 envs and more), at most one per env."""
        b = self.num_envs
        if all_slots or self._ep_len_host is None:
            return b
        n_trunc = int(np.count_nonzero(self._ep_len_host + 1 >= self.horizon))
        # running mean of ends per step: bias-corrected while young (the mean of the steps seen so far, not a mean pulled to zero),
        # never below the last step's count (fast attack: a regime whose ends jump up is believed at once), and with a prior of
        # half an end per step that fades within the first two windows (a fresh env's first `end` then finds a spare slot; a
        # regime in which nobody ever ends stops paying for slots after ~28 steps)
        n = self._end_steps
        m = max(self._end_mean / (1.0 - 0.95 ** n) if n else 0.0, float(self._end_last), 0.5 * 0.8 ** n)
        k = n_trunc + (0 if m < 1e-3 else _poisson_quantile(m, self.DR_END_TAIL))
        g = 4 if b >= 64 else 1  # (small batches -- configs[4] has 8 envs per GPU -- pay four encoder frames per spare slot: none is added)
        return 0 if k == 0 else min(b, (k + g - 1) // g * g)

    def reset_statistics(self) -> None:
        """Forget the running statistics the slot margin is sized from (a caller that CHANGES the regime, e.g. bench.py between its
        end-rate lines; otherwise they adapt within ~20 steps)."""
        self._end_mean, self._end_last, self._end_steps = 0.0, 0, 0

    def slots_preferred(self) -> bool:
        """Should env_loop take the slots loop for this env (asked after reset())?  Not while the sampler is replayed as a hipGraph
        (a few small frames: see make_env_loop)."""
        return not self._use_graph()

    def slots_can_repeat(self) -> bool:
        """Can a window be repeated after a SlotOverflow?  Not when random draws come from stateful hooks (the tests' injected
        draws) or from inside a replayed sampler graph: then every step gets a slot per env (no overflow is possible)."""
        return self.expo_fn is None and self.sampler.noise_fn is None and not self._use_graph()

    @torch.no_grad()
    def step_end_slots(self, all_slots: bool = False):
        """Second half of a step with its deaths resolved on the device.  Returns (obs_ext, rew, end, trunc, slots, info):
        obs_ext (B + T * K, C, H, W) = [every env's newest frame (a reset row: of its new episode) | the K slots' final observations
        | their T - 1 burn-in frames, frame-major]; slots None when the step has no slots (K = 0).  Waits only for the PREVIOUS
        step's report (slots_finish)."""
        next_obs, denoising_trajectory, e_rew, e_end, _ = self._pending
        self._pending = None
        rew, end = self.predict_rew_end(next_obs.unsqueeze(1), e_rew, e_end)
        self.slots_finish()  # the previous step's report: episode-length mirror, pool cursor, overflow
        dev, b = next_obs.device, self.num_envs
        t = self._ctx.shape[1]
        k = self.slot_count(all_slots)
        end = end.long().contiguous()
        if self.ep_len.dtype != torch.long or not self.ep_len.is_contiguous():
            self.ep_len = self.ep_len.long().contiguous()
        trunc = torch.empty(b, dtype=torch.long, device=dev)
        dead = torch.empty(b, dtype=torch.uint8, device=dev)
        slot_row = torch.empty(max(k, 1), dtype=torch.long, device=dev)
        row_slot = torch.empty(b, dtype=torch.int32, device=dev)
        report = torch.empty(b + 4, dtype=torch.int32, device=dev)
        lib = nv.lib()
        nv.check(lib.dmd_resolve_deaths(nv.ptr(end), nv.ptr(self.ep_len), int(self.horizon), nv.ptr(trunc), nv.ptr(dead), b, k, nv.ptr(slot_row),
                                        nv.ptr(row_slot), nv.ptr(report), nv.stream()), "dmd_resolve_deaths")
        if report.is_cuda:
            if self._report_host is None or self._report_host.shape[1] != b + 4:
                self._report_host = torch.zeros(4, b + 4, dtype=torch.int32).pin_memory()
                self._report_slot = 0
            self._report_slot = (self._report_slot + 1) % self._report_host.shape[0]
            host = self._report_host[self._report_slot]
            host.copy_(report, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
        else:
            host, event = report.clone(), None
        pool = self.pool
        token, two_rounds = None, False
        if k > 0:
            if pool.size == 0:
                pool._preload()
            token = (pool._generation, pool._cursor)
            # The reference drops the rest of a round and preloads the next one when a request does not fit (:133-139): whether that
            # happens depends on this step's exact number of deaths.  If it MAY happen (k slots would not fit), the launch gets both
            # rounds and the device picks by the count (dmd_reset_slots); the host follows when it reads the report.
            two_rounds = pool._cursor + k > pool.size
            if two_rounds:
                pool.prefetch()  # (normally loaded steps ago, below)
        self._slots_inflight = (event, host, k, token, two_rounds)
        self._head = (self._head + 1) % t  # the ring advances: dmd_reset_slots writes the imagined frame into the slot this frees
        enc_in = torch.empty((b + t * k,) + tuple(next_obs.shape[1:]), dtype=torch.float32, device=dev)
        nxt = next_obs if (next_obs.dtype == torch.float32 and next_obs.is_contiguous()) else next_obs.float().contiguous()
        p = nv.ResetSlotsParams()
        p.B, p.K, p.T, p.head, p.per_frame = b, k, t, self._head, nxt[0].numel()
        p.row_slot, p.next_obs, p.ctx, p.enc_in = nv.ptr(row_slot), nv.fptr(nxt), nv.fptr(self._ctx), nv.fptr(enc_in)
        keep = []  # (tensor views whose pointers the launch parameters hold)
        if k > 0:
            assert self._act.is_contiguous() and self._act.dtype == torch.long
            rounds = [(pool.frames_u8, pool.frames_f32, pool.act, pool.hx, pool.cx, pool.pad)] + ([pool._next.fields()] if two_rounds else [])
            for i, (fu8, ff32, pact, phx, pcx, ppad) in enumerate(rounds):
                frames = fu8 if fu8 is not None else ff32
                assert frames.is_contiguous() and pact.dtype == torch.long and pact.is_contiguous()
                r = p.pool[i]
                r.frames, r.is_f32, r.rows = nv.ptr(frames), int(fu8 is None), frames.shape[0]
                if ppad is not None and fu8 is not None:
                    pad_u8 = ppad.view(torch.uint8)
                    keep.append(pad_u8)
                    r.pad = nv.ptr(pad_u8)
                r.act, r.hx, r.cx = nv.ptr(pact), nv.fptr(phx), nv.fptr(pcx)
            if two_rounds:
                assert k <= rounds[1][2].shape[0], "more simultaneous resets than one preload round holds"
                p.num_dead = report.data_ptr() + 4 * b
            p.pool_base, p.hd = int(pool._cursor), pool.hx.shape[-1]
            p.slot_row, p.act_ring, p.hx, p.cx = nv.ptr(slot_row), nv.ptr(self._act), nv.fptr(self.hx_rew_end), nv.fptr(self.cx_rew_end)
        with _no_random_draws(dev, "WorldModelEnv.step_end_slots"):
            nv.check(lib.dmd_reset_slots(C.byref(p), nv.stream()), "dmd_reset_slots")
        dead_b = dead.view(torch.bool)
        info: Dict[str, Any] = {"dead": dead_b}
        if self.return_denoising_trajectory:
            info["denoising_trajectory"] = torch.stack(denoising_trajectory, dim=1)
        self.stats["steps"] += 1
        self.stats["slots"] += k
        slots = ResetSlots(k, slot_row[:k], row_slot, dead_b) if k > 0 else None
        # the next round ahead of need: its uploads, burn-in and quantisation are queued behind this step, steps before the round
        # can be asked for (nothing waits for it; the loader's batches are consumed in the reference's order, only earlier)
        if k > 0 and pool._next is None and pool.size - pool._cursor < self.PREFETCH_STEPS * k:
            with _no_random_draws(dev, "the initial-condition pool's prefetch", cpu=False):
                pool.prefetch()
        return enc_in, rew, end, trunc, slots, info

    def slots_finish(self) -> None:
        """Read the report of the last step_end_slots (waits for that step: the device is then a full step of queued work ahead,
        except at a window's end): episode-length mirror, pool cursor, running mean of ends, slot overflow (raises)."""
        inflight, self._slots_inflight = self._slots_inflight, None
        if inflight is None:
            return
        event, host, k, token, two_rounds = inflight
        if event is not None:
            t0 = time.perf_counter()
            event.synchronize()
            self.stats["report_wait_ms"] += 1e3 * (time.perf_counter() - t0)  # (the host's slack: ~0 means the HOST paces the loop)
            check_weight_audits()  # (the host is synchronised anyway: did an audit of the packed weight copies find a silent write?)
        self._slots_account(host.numpy(), k, token, two_rounds)

    def _slots_account(self, report: np.ndarray, k: int, token, two_rounds: bool) -> None:
        b = self.num_envs
        n_dead, n_end, overflow = int(report[b]), int(report[b + 1]), int(report[b + 2])
        rows_host = np.flatnonzero(report[:b])
        if self._ep_len_host is not None:
            self._ep_len_host += 1
            self._ep_len_host[rows_host] = 0
        self._end_mean = 0.95 * self._end_mean + 0.05 * n_end
        self._end_last = n_end
        self._end_steps += 1
        self.stats["dead_rows"] += n_dead
        if n_dead:
            self.stats["steps_with_deaths"] += 1
        if overflow:
            self.stats["slot_overflows"] += 1
            raise SlotOverflow(f"{n_dead} episodes ended in a step with {k} reset slots")
        if n_dead:
            pool = self.pool
            assert token == (pool._generation, pool._cursor), "pool rows were served between a step's launch and its report"
            if pool._cursor + n_dead > pool.size:  # the device served this step from the next round (dmd_reset_slots): follow it
                assert two_rounds
                pool._preload()
                self.stats["pool_rounds"] += 1
            pool._cursor += n_dead

    @torch.no_grad()
    def slots_snapshot(self):
        """Everything a repetition of the coming window starts from (env state, pool position, random generators)."""
        self.slots_finish()
        dev = self._ctx.device
        return {"ctx": self._ctx.clone(), "act": self._act.clone(), "head": self._head, "hx": self.hx_rew_end.clone(), "cx": self.cx_rew_end.clone(),
                "ep_len": self.ep_len.clone(), "ep_host": None if self._ep_len_host is None else self._ep_len_host.copy(),
                "pool": self.pool.snapshot(), "cpu_rng": torch.get_rng_state(),
                "dev_rng": torch.cuda.get_rng_state(dev) if dev.type == "cuda" else None}

    @torch.no_grad()
    def slots_restore(self, snap) -> None:
        self._slots_inflight = None
        self._pending = None
        self._ctx.copy_(snap["ctx"])
        self._act.copy_(snap["act"])
        self._head = snap["head"]
        self.hx_rew_end, self.cx_rew_end, self.ep_len = snap["hx"].clone(), snap["cx"].clone(), snap["ep_len"].clone()
        self._ep_len_host = None if snap["ep_host"] is None else snap["ep_host"].copy()
        self.pool.restore(snap["pool"])
        torch.set_rng_state(snap["cpu_rng"])
        if snap["dev_rng"] is not None:
            torch.cuda.set_rng_state(snap["dev_rng"], self._ctx.device)

    def _use_graph(self) -> bool:
        # (no replay while a launch profiler is installed: a replayed graph issues no launches it could time, and a first
        #  capture inside the profiled window would record timing events into the graph)
        if not (self.graph_sampler and self.sampler.noise_fn is None and nv.PROFILER is None):
            return False
        if self._graph_forced or self._ctx is None:
            return True
        b, _, _, h, w = self._ctx.shape
        return b * h * w <= GRAPH_SAMPLER_MAX_PIXELS

    # -- the two callables trainer.py:182-184 re-assigns -----------------------------------------------------------------
    # `rl_env.predict_next_obs = torch.compile(rl_env.predict_next_obs, mode="reduce-overhead")` (and the same for predict_rew_end) is
    # what the reference does under its DEFAULT configuration (config/trainer.yaml:60 `compile_wm: True`).  Both are sequences of
    # HIP launches through ctypes: there is nothing for a tracing compiler in them, and letting dynamo trace the host code around
    # the launches costs a minute of compilation, hits its recompilation limit on the per-layer host objects and moves the
    # initial-noise draw into a compiled region (another random stream).  So they are properties: any callable may be assigned
    # (wrappers, spies: called like the reference's attribute), and a torch.compile wrapper is undone on assignment
    # (_without_torch_compile) -- the no-op wrapper SURVEY 8(b) asks for; the trainer runs unchanged either way.
    @property
    def predict_next_obs(self):
        fn = self.__dict__.get("_predict_next_obs_fn")
        return self._predict_next_obs if fn is None else fn

    @predict_next_obs.setter
    def predict_next_obs(self, fn) -> None:
        self.__dict__["_predict_next_obs_fn"] = _without_torch_compile(fn, self._predict_next_obs)

    @property
    def predict_rew_end(self):
        fn = self.__dict__.get("_predict_rew_end_fn")
        return self._predict_rew_end if fn is None else fn

    @predict_rew_end.setter
    def predict_rew_end(self, fn) -> None:
        self.__dict__["_predict_rew_end_fn"] = _without_torch_compile(fn, self._predict_rew_end)

    @torch.no_grad()
    def _predict_next_obs(self) -> Tuple[Tensor, List[Tensor]]:
        """Reference signature (world_model_env.py:91-93).  The initial noise step_begin drew for this call, if any, waits in
        `_next_noise`; a direct call draws its own."""
        noise, self._next_noise = getattr(self, "_next_noise", None), None
        if self._use_graph():
            return self.sampler.sample_ring_graphed(self._ctx, self._act, self._head, self._head)
        return self.sampler.sample_ring(self._ctx, self._act, self._head, self._head, noise)

    @torch.no_grad()
    def _predict_rew_end(self, next_obs: Tensor, e_rew: Optional[Tensor] = None, e_end: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
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
