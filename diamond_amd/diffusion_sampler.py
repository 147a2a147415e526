"""Karras-schedule Euler / Heun sampler (reference models/diffusion/diffusion_sampler.py).

The sigma schedule is evaluated once on the host (the reference compares 0-dim DEVICE
tensors inside the loop, forcing a sync per step, diffusion_sampler.py:39,47); the update
`x + (x - D)/sigma_hat * dt` is one fused HIP kernel per step.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import native as nv
from .denoiser import Denoiser

# the FiLM tables of a frame's denoising steps computed together in front of the loop (DIAMOND_BATCH_FILM=0: per step, the A/B arm)
BATCH_FILM_TABLES = os.environ.get("DIAMOND_BATCH_FILM", "1") != "0"


@dataclass
class DiffusionSamplerConfig:
    num_steps_denoising: int
    sigma_min: float = 2e-3
    sigma_max: float = 5
    rho: int = 7
    order: int = 1
    s_churn: float = 0
    s_tmin: float = 0
    s_tmax: float = float("inf")
    s_noise: float = 1


def build_sigmas(num_steps: int, sigma_min: float, sigma_max: float, rho: int, device: torch.device) -> Tensor:
    """Karras et al. schedule + trailing 0, evaluated in fp32 exactly like the reference
    (linspace then ** rho, diffusion_sampler.py:61-66) -- on the CPU so the values are the
    CPU reference's bit for bit, then moved."""
    lo = sigma_min ** (1 / rho)
    hi = sigma_max ** (1 / rho)
    ramp = torch.linspace(0, 1, num_steps)
    s = (hi + ramp * (lo - hi)) ** rho
    return torch.cat((s, s.new_zeros(1))).to(device)


class DiffusionSampler:
    def __init__(self, denoiser: Denoiser, cfg: DiffusionSamplerConfig) -> None:
        self.denoiser = denoiser
        self.cfg = cfg
        self.sigmas = build_sigmas(cfg.num_steps_denoising, cfg.sigma_min, cfg.sigma_max, cfg.rho, denoiser.device)
        self._host_sigmas = self.sigmas.detach().cpu()  # fp32 values, compared on the host
        # test hook: replaces torch.randn for the initial noise / churn draws (host-injected RNG)
        self.noise_fn: Optional[Callable[[Tuple[int, ...], torch.device], Tensor]] = None
        self._graphs: Dict[tuple, "_CapturedSample"] = {}  # latency mode: one hipGraph per (buffers, ring heads)
        self._graph_stamp = None  # (denoiser parameter versions / pointers, arithmetic switches) the cached graphs were captured with
        self._graph_pool = None                             # ... all drawing their temporaries from one memory pool

    def _randn(self, shape, device) -> Tensor:
        if self.noise_fn is not None:
            return self.noise_fn(tuple(shape), device)
        return torch.randn(*shape, device=device)

    @torch.no_grad()
    def sample(self, prev_obs: Tensor, prev_act: Tensor) -> Tuple[Tensor, List[Tensor]]:
        """Reference entry point (diffusion_sampler.py:30-58): prev_obs (B, T, C, H, W), prev_act (B, T)."""
        return self.sample_ring(prev_obs, prev_act, 0, 0)

    @torch.no_grad()
    def sample_ring_graphed(self, ctx_obs: Tensor, ctx_act: Tensor, obs_head: int, act_head: int) -> Tuple[Tensor, List[Tensor]]:
        """Latency mode (play.py's B=1 world-model env, play.py:105-109 / game/play_env.py:113-124): the ~280 kernel
        launches of one `sample` are captured once into a hipGraph per (context buffers, ring heads) and replayed -- at
        B=1 every launch is latency-bound and the host cannot even issue them as fast as the GPU retires them.
        The sigma schedule, conditioners and ring heads are launch constants of the captured graph; the context is read
        from the SAME buffers at replay time (WorldModelEnv keeps its rings in place).  Returns fresh copies (the
        graph's output buffers are overwritten by the next replay)."""
        # A captured graph holds the POINTERS of the packed weights and the kernel choices of capture time.  The packed copies
        # are rebuilt IN PLACE (engine.PackCache, blocks.FilmTable), so a weight update does not void a graph: it only has to
        # be followed by a refresh of the copies before the next replay -- there is no lookup inside a replay that would do
        # it.  `versions` notices optimizer steps / loads (and, through the caches' stale_epoch, invalidations that bump no
        # `Tensor._version`; a replayed GraphedTrainStep refreshes the copies itself).  Graphs are dropped only when a buffer
        # they point to was replaced (frees_epoch) or an arithmetic / routing switch changed.
        from . import blocks as BL
        from . import engine as E

        nv.check_current_device(ctx_obs.device)
        from .train_graph import _refresh_weight_caches, _weight_caches

        # (the module walk of `parameters()` is ~350 us of Python per call, a tenth of a B = 1 frame: the lists are kept and
        #  re-derived only now and then)
        self._stamp_calls = getattr(self, "_stamp_calls", 0) + 1
        if getattr(self, "_stamp_params", None) is None or self._stamp_calls % 256 == 0:
            self._stamp_params = list(self.denoiser.parameters())
            self._stamp_caches = list(_weight_caches(self.denoiser))
        versions = (tuple((p._version, p.data_ptr()) for p in self._stamp_params), sum(c.stale_epoch for c in self._stamp_caches))
        if versions != getattr(self, "_graph_versions", None):
            _refresh_weight_caches(self.denoiser)
            self._graph_versions = versions
        stamp = lambda: (sum(c.frees_epoch for c in self._stamp_caches), E.WORLD_MODEL_PRECISION, BL.LOWRES_CHAIN)
        if stamp() != self._graph_stamp:
            self._graphs.clear()
            self._graph_stamp = stamp()
        key = (ctx_obs.data_ptr(), ctx_act.data_ptr(), tuple(ctx_obs.shape), obs_head, act_head)
        cap = self._graphs.get(key)
        if cap is None:
            if len(self._graphs) > 64:
                self._graphs.clear()
            cap = _CapturedSample(self, ctx_obs, ctx_act, obs_head, act_head)
            # the capture's warm-up may have CREATED caches (the lazy FilmTable, packed copies on a new device): they belong in
            # the lists `versions` and `stamp` are derived from, from this call on
            self._stamp_params = list(self.denoiser.parameters())
            self._stamp_caches = list(_weight_caches(self.denoiser))
            self._graph_versions = (tuple((p._version, p.data_ptr()) for p in self._stamp_params), sum(c.stale_epoch for c in self._stamp_caches))
            if stamp() != self._graph_stamp:  # the warm-up replaced a buffer (e.g. copies built on another device before)
                self._graphs.clear()
                self._graph_stamp = stamp()
            self._graphs[key] = cap
        x, traj = cap.replay()
        return x.clone(), [t.clone() for t in traj]

    @torch.no_grad()
    def sample_ring(self, ctx_obs: Tensor, ctx_act: Tensor, obs_head: int, act_head: int,
                    noise: Optional[Tensor] = None) -> Tuple[Tensor, List[Tensor]]:
        """`sample` over ring-indexed context buffers (logical step t at slot (head + t) % T): what WorldModelEnv
        calls every step, so its context is never rolled (world_model_env.py:74-75).  noise: the initial draw (B, C, H, W),
        made by the caller (WorldModelEnv keeps it to repeat a dropped speculative step with the same draw)."""
        device = ctx_obs.device
        nv.check_current_device(device)  # (ctypes launches go to the CURRENT device's stream: a sampler on another GPU raises)
        b, t, c, h, w = ctx_obs.size()
        ctx_obs = ctx_obs.contiguous()
        ring = (obs_head, act_head)
        sig = self._host_sigmas
        gamma_ = min(self.cfg.s_churn / (len(sig) - 1), 2 ** 0.5 - 1)
        x = noise if noise is not None else self._randn((b, c, h, w), device)
        trajectory = [x]
        # the conditioning of every step depends on (sigma, actions) only: the FiLM tables of the whole schedule in front of the
        # loop, as three GEMM launches instead of three per step (bitwise the per-step tables; None: schedule too long to keep)
        tables = self.denoiser.film_tables(list(sig[:-1]), ctx_act, act_head) if BATCH_FILM_TABLES else None
        tab = (lambda i: tables[i]) if tables is not None else (lambda i: None)
        for i, (sigma, next_sigma) in enumerate(zip(sig[:-1], sig[1:])):  # 0-dim fp32 CPU tensors
            gamma = gamma_ if self.cfg.s_tmin <= sigma <= self.cfg.s_tmax else 0
            sigma_hat = sigma * (gamma + 1)
            if gamma > 0:
                eps = self._randn(x.shape, device) * self.cfg.s_noise
                x = x + eps * float((sigma_hat ** 2 - sigma ** 2) ** 0.5)
            denoised = self.denoiser.denoise(x, sigma, ctx_obs, ctx_act, ring=ring, table=tab(i))  # sigma, not sigma_hat: reference :44
            dt = next_sigma - sigma_hat  # fp32 subtraction like the reference
            if self.cfg.order == 1 or next_sigma == 0:
                x = self._euler(x, denoised, float(sigma_hat), float(dt))
            else:
                x_2 = self._euler(x, denoised, float(sigma_hat), float(dt))
                # the reference passes next_sigma as a (B,) tensor of equal values (:53); one scalar
                # yields the same per-sample conditioners without a per-sample array
                denoised_2 = self.denoiser.denoise(x_2, next_sigma, ctx_obs, ctx_act, ring=ring, table=tab(i + 1))  # (next_sigma != 0: step i + 1 exists)
                x = self._heun(x, denoised, x_2, denoised_2, float(sigma_hat), float(next_sigma), float(dt))
            trajectory.append(x)
        return x, trajectory

    @staticmethod
    def _euler(x: Tensor, denoised: Tensor, sigma_hat: float, dt: float) -> Tensor:
        x = x.contiguous()
        out = torch.empty_like(x)
        nv.check(nv.lib().dmd_euler_step(nv.fptr(x), nv.fptr(denoised), sigma_hat, dt, nv.fptr(out), x.numel(),
                                         nv.stream()), "dmd_euler_step")
        return out

    @staticmethod
    def _heun(x: Tensor, denoised: Tensor, x_2: Tensor, denoised_2: Tensor, sigma_hat: float, sigma_next: float,
              dt: float) -> Tensor:
        """x + ((x - D)/sigma_hat + (x_2 - D_2)/sigma_next) / 2 * dt  (reference :52-56), one fused kernel."""
        x = x.contiguous()
        out = torch.empty_like(x)
        nv.check(nv.lib().dmd_heun_step(nv.fptr(x), nv.fptr(denoised), nv.fptr(x_2), nv.fptr(denoised_2), sigma_hat,
                                        sigma_next, dt, nv.fptr(out), x.numel(), nv.stream()), "dmd_heun_step")
        return out


class _CapturedSample:
    """One `DiffusionSampler.sample_ring` call captured into a hipGraph (torch.cuda.CUDAGraph).  Every dmd_* entry
    point is asynchronous on torch's current stream, allocates nothing and synchronises nothing (include/diamond_hip.h),
    so the whole launch sequence -- including torch's own randn for the initial noise -- records into the capture
    stream; two eager warm-up calls first populate the packed-weight / conditioner caches and set the kernels' LDS
    attributes (none of which may happen during capture)."""

    def __init__(self, sampler: DiffusionSampler, ctx_obs: Tensor, ctx_act: Tensor, obs_head: int, act_head: int) -> None:
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):
                sampler.sample_ring(ctx_obs, ctx_act, obs_head, act_head)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        if sampler._graph_pool is None:
            sampler._graph_pool = torch.cuda.graph_pool_handle()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads' HIP calls (the RCCL watchdog polls its events) must not invalidate this capture
        with torch.cuda.graph(self.graph, pool=sampler._graph_pool, capture_error_mode="thread_local"):  # replays are sequential, outputs are copied out
            self.x, self.trajectory = sampler.sample_ring(ctx_obs, ctx_act, obs_head, act_head)

    def replay(self) -> Tuple[Tensor, List[Tensor]]:
        self.graph.replay()
        return self.x, self.trajectory
