"""EDM denoiser (reference models/diffusion/denoiser.py), inference path on hand-written HIP.

`denoise` keeps the reference signature `(noisy_next_obs, sigma, obs, act) -> denoised`
with NCHW tensors; sigma may be a 0-dim tensor, a (B,) tensor (Heun's second evaluation,
diffusion_sampler.py:53) or a python float.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor, nn

from . import engine as E
from . import native as nv
from .inner_model import InnerModel, InnerModelConfig


@dataclass
class SigmaDistributionConfig:
    loc: float
    scale: float
    sigma_min: float
    sigma_max: float


@dataclass
class DenoiserConfig:
    inner_model: InnerModelConfig
    sigma_data: float
    sigma_offset_noise: float


class Denoiser(nn.Module):
    def __init__(self, cfg: DenoiserConfig) -> None:
        super().__init__()
        self.cfg = cfg
        self.inner_model = InnerModel(cfg.inner_model)
        self.sample_sigma_training = None
        self._cond_cache = {}  # float sigma -> (1, 4) device conditioners
        # test hook: replaces torch.randn in the training step (host-injected draws, reference RNG order)
        self.randn_fn = None
        self._train_params = None

    @property
    def device(self) -> torch.device:
        return self.inner_model.noise_emb.weight.device

    def setup_training(self, cfg: SigmaDistributionConfig) -> None:
        assert self.sample_sigma_training is None

        def sample_sigma(n: int, device: torch.device):
            s = self._randn((n,), device) * cfg.scale + cfg.loc
            return s.exp().clip(cfg.sigma_min, cfg.sigma_max)

        self.sample_sigma_training = sample_sigma

    @torch.no_grad()
    def compute_conditioners(self, sigma: Union[Tensor, float]) -> Tuple[Tensor, int]:
        """(c_in, c_out, c_skip, c_noise) of reference denoiser.py:66-72 as a device array
        cond[n * stride + k], stride 0 (one sigma for the batch) or 4 (one sigma per sample).

        Four scalars per sample: evaluated on the HOST with the reference's own fp32 torch-CPU
        op order, so all four are the CPU reference's values bit for bit (device sqrt/log/div are
        not guaranteed to round like the host's); the kernels only multiply by them.  Scalar sigmas
        (the sampler's whole schedule) are cached on the device: no host<->device traffic per
        denoising step, and the launch sequence stays graph-capturable."""
        key = None
        if not torch.is_tensor(sigma):
            sigma = torch.tensor(float(sigma), dtype=torch.float32)
        if sigma.numel() == 1 and not sigma.is_cuda:
            key = float(sigma)
            hit = self._cond_cache.get(key)
            if hit is not None and hit.device == self.device:
                return hit, 0
        s = sigma.detach().to(device="cpu", dtype=torch.float32).reshape(-1)  # sync only for device sigmas
        s = (s ** 2 + self.cfg.sigma_offset_noise ** 2).sqrt()
        c_in = 1 / (s ** 2 + self.cfg.sigma_data ** 2).sqrt()
        c_skip = self.cfg.sigma_data ** 2 / (s ** 2 + self.cfg.sigma_data ** 2)
        c_out = s * c_skip.sqrt()
        c_noise = s.log() / 4
        cond = torch.stack((c_in, c_out, c_skip, c_noise), dim=1).contiguous().to(self.device)
        if key is not None:
            if len(self._cond_cache) > 4096:
                self._cond_cache.clear()
            self._cond_cache[key] = cond
        return cond, (0 if cond.shape[0] == 1 else 4)

    @torch.no_grad()
    def compute_model_output(self, noisy_next_obs: Tensor, obs: Tensor, act: Tensor, sigma: Union[Tensor, float],
                             naive: Optional[bool] = None, precision: Optional[str] = None,
                             ring: Optional[Tuple[int, int]] = None, table: Optional[Tensor] = None) -> Tensor:
        """F = inner_model(x * c_in, c_noise, obs / sigma_data, act)  (reference :74-77).
        precision: None (engine.WORLD_MODEL_PRECISION) | "f32" | "f16x2".
        ring = (obs_head, act_head): `obs` is then the PHYSICAL ring (N, T, C, H, W) of conditioning frames and `act`
        the physical (N, T) ring of actions of a WorldModelEnv (logical step t at slot (head + t) % T); the ring
        order is resolved inside dmd_edm_pack_input / dmd_cond_embed, nothing is rolled or copied."""
        nv.check_current_device(noisy_next_obs.device)  # (ctypes launches go to the CURRENT device's stream)
        n, cx, h, w = noisy_next_obs.shape
        t_ring, obs_head, act_head = 1, 0, 0
        if ring is not None:
            assert obs.ndim == 5, "ring mode takes the (N, T, C, H, W) context buffer"
            t_ring, (obs_head, act_head) = obs.shape[1], ring
            obs = obs.reshape(n, -1, h, w)
        cobs = obs.shape[1]
        cond, stride = self.compute_conditioners(sigma)
        assert stride == 0 or cond.shape[0] == n, "sigma must be a scalar or one value per sample"
        cpad = (cx + cobs + 15) // 16 * 16
        # image sizes whose U-Net levels are not multiples of the kernels' tiles (e.g. 72x72: 72 / 36 / 18 / 9) run as the
        # VALID EXTENT of a larger zero-padded buffer (include/diamond_hip.h); the result is cropped back below
        valid = None
        hp, wp = E.padded_extent(h, w, self.inner_model.unet._num_down)
        if (hp, wp) != (h, w):
            import torch.nn.functional as F

            valid = (h, w)
            noisy_next_obs = F.pad(noisy_next_obs, (0, wp - w, 0, hp - h))
            obs = F.pad(obs, (0, wp - w, 0, hp - h))
            h, w = hp, wp
        packed = torch.empty(n, h, w, cpad, device=self.device, dtype=torch.float32)
        # NOTE: pointers are only taken from tensors bound to a name -- a temporary made inside the
        # argument list would be freed (and its block re-used) before the kernel is even launched.
        xc, oc = noisy_next_obs.contiguous(), obs.contiguous()
        nv.check(nv.lib().dmd_edm_pack_input(nv.fptr(xc), nv.fptr(oc), nv.fptr(cond), stride, float(self.cfg.sigma_data),
                                             nv.fptr(packed), n, cx, cobs, h, w, cpad, t_ring, obs_head, nv.stream()),
                 "dmd_edm_pack_input")
        # (table: this forward's FiLM table, if the caller computed the tables of several steps at once -- film_tables below)
        cvec = self.inner_model.cond_vector(cond, stride, act, act_head) if table is None else None
        out = self.inner_model.run(packed, cvec, naive, precision, table=table, valid=valid)
        return out if valid is None else out[:, :, :valid[0], :valid[1]].contiguous()

    @torch.no_grad()
    def wrap_model_output(self, noisy_next_obs: Tensor, model_output: Tensor, sigma: Union[Tensor, float],
                          cond4: Optional[Tensor] = None) -> Tensor:
        """quantise(c_skip * x + c_out * F)  (reference :79-84).  cond4: an already computed (B, 4) device array of
        (c_in, c_out, c_skip, c_noise) to use instead of deriving the conditioners from sigma (training step: no host
        round trip, and the refresh uses exactly the conditioners the loss used)."""
        n = noisy_next_obs.shape[0]
        if cond4 is not None:
            assert cond4.shape == (n, 4) and cond4.dtype == torch.float32 and cond4.is_contiguous()
            cond, stride = cond4, 4
        else:
            cond, stride = self.compute_conditioners(sigma)
        x = noisy_next_obs.contiguous()
        f = model_output.contiguous()
        out = torch.empty_like(x)
        nv.check(nv.lib().dmd_edm_denoised(nv.fptr(x), nv.fptr(f), nv.fptr(cond), stride, nv.fptr(out), n, x[0].numel(),
                                           nv.stream()), "dmd_edm_denoised")
        return out

    @torch.no_grad()
    def denoise(self, noisy_next_obs: Tensor, sigma: Union[Tensor, float], obs: Tensor, act: Tensor,
                ring: Optional[Tuple[int, int]] = None, table: Optional[Tensor] = None) -> Tensor:
        f = self.compute_model_output(noisy_next_obs, obs, act, sigma, ring=ring, table=table)
        return self.wrap_model_output(noisy_next_obs, f, sigma)

    # at most this many bytes of FiLM tables at once (a 50-step schedule at batch 256 would be 367 MB: computed per step instead)
    FILM_TABLES_MAX_BYTES = 64 << 20

    @torch.no_grad()
    def film_tables(self, sigmas: Sequence[Union[Tensor, float]], act: Tensor, act_head: int = 0) -> Optional[List[Tensor]]:
        """The FiLM tables of `denoise` at each of `sigmas` (scalars) for one action context, computed together (the noise
        embedding differs per step, cond_proj and the AdaGroupNorm linears are the same three GEMMs: three launches instead of three
        per denoising step, bitwise the per-step tables).  None: too large to keep (the caller then leaves `table` unset)."""
        im = self.inner_model
        per_step = act.shape[0] * 4 * sum(2 * m.in_channels for m in im.unet.modules() if type(m).__name__ == "AdaGroupNorm")
        if len(sigmas) < 2 or len(sigmas) * per_step > self.FILM_TABLES_MAX_BYTES:
            return None
        conds = [self.compute_conditioners(s) for s in sigmas]
        assert all(stride == 0 for _, stride in conds), "film_tables: scalar sigmas"
        return im.film_tables(conds, act, act_head)

    # -- training step (reference denoiser.py:60-63,93-122) -------------------------------------------------------
    def _randn(self, shape, device) -> Tensor:
        if self.randn_fn is not None:
            return self.randn_fn(tuple(shape)).to(device)
        return torch.randn(*shape, device=device)

    def apply_noise(self, x: Tensor, sigma: Tensor, sigma_offset_noise: float) -> Tensor:
        b, c, _, _ = x.shape
        offset_noise = sigma_offset_noise * self._randn((b, c, 1, 1), x.device)
        return x + offset_noise + self._randn(tuple(x.shape), x.device) * sigma.reshape(-1, 1, 1, 1)

    def _training_conditioners(self, sigma: Tensor):
        """compute_conditioners (:66-72) for a (B,) device sigma, as fp32 torch ops in the reference's order."""
        s = (sigma ** 2 + self.cfg.sigma_offset_noise ** 2).sqrt()
        c_in = 1 / (s ** 2 + self.cfg.sigma_data ** 2).sqrt()
        c_skip = self.cfg.sigma_data ** 2 / (s ** 2 + self.cfg.sigma_data ** 2)
        c_out = s * c_skip.sqrt()
        c_noise = s.log() / 4
        return c_in, c_out, c_skip, c_noise

    def model_output_with_grad(self, noisy_next_obs: Tensor, obs: Tensor, act: Tensor, conditioners,
                               precision: Optional[str] = None) -> Tensor:
        """F = inner_model(x * c_in, c_noise, obs / sigma_data, act) (:74-77) differentiable w.r.t. every parameter:
        the U-Net forward runs on the HIP kernels while its launches are recorded, its backward is
        unet_train.UNetTrainFn; the cond vector / FiLM table (three small GEMMs) run on dmd_linear through lstm_native.LinearFn,
        whose backward is dmd_linear as well (torch autograd only carries the gradient between the nodes)."""
        import math
        import torch.nn.functional as F
        from . import unet_train as UT
        from .blocks import FilmTable

        im = self.inner_model
        c_in, c_out, c_skip, c_noise = conditioners
        n, cx, h, w = noisy_next_obs.shape
        cobs = obs.shape[1]
        cond4 = torch.stack((c_in, c_out, c_skip, c_noise), dim=1).detach().float().contiguous()
        cpad = (cx + cobs + 15) // 16 * 16
        packed = torch.empty(n, h, w, cpad, device=self.device, dtype=torch.float32)
        xc, oc = noisy_next_obs.detach().contiguous(), obs.detach().contiguous()
        nv.check(nv.lib().dmd_edm_pack_input(nv.fptr(xc), nv.fptr(oc), nv.fptr(cond4), 4, float(self.cfg.sigma_data), nv.fptr(packed),
                                             n, cx, cobs, h, w, cpad, 1, 0, nv.stream()), "dmd_edm_pack_input")
        # cond = cond_proj(noise_emb(c_noise) + act_emb(act))  (inner_model.py:45, blocks.py:84-87)
        # (an outer product: every element is ONE fp32 product, exactly what the (B, 1) @ (1, C) GEMM computes)
        f = 2 * math.pi * c_noise.detach().unsqueeze(1) * im.noise_emb.weight
        # the cond MLP and the FiLM table -- three small GEMMs and their backward -- on dmd_linear (lstm_native.LinearFn)
        from .lstm_native import LinearFn

        l0, l2 = im.cond_proj[0], im.cond_proj[2]
        cond = LinearFn.apply(im._cache, torch.cat([f.cos(), f.sin()], dim=-1) + im.act_emb(act), l0.weight, l0.bias)
        cond = LinearFn.apply(im._cache, F.silu(cond), l2.weight, l2.bias)
        if im._film is None:
            im._film = FilmTable(im.unet)
        w_cat = torch.cat([m.linear.weight for m in im._film.norms], dim=0)
        b_cat = torch.cat([m.linear.bias for m in im._film.norms], dim=0)
        table = LinearFn.apply(im._cache, cond, w_cat, b_cat)
        if self._train_params is None:
            self._train_params = UT.trainable_unet_params(im)
        return UT.UNetTrainFn.apply(im, packed, table, precision or UT.TRAIN_PRECISION, *self._train_params)

    def forward(self, batch):
        """Denoising loss over a segment with autoregressive refresh of the context (reference :93-122)."""
        import torch.nn.functional as F

        nv.check_current_device(batch.obs.device)
        n = self.cfg.inner_model.num_steps_conditioning
        seq_length = batch.obs.size(1) - n
        all_obs = batch.obs.clone()
        loss = 0
        for i in range(seq_length):
            obs = all_obs[:, i:n + i]
            next_obs = all_obs[:, n + i]
            act = batch.act[:, i:n + i]
            mask = batch.mask_padding[:, n + i]
            b, t, c, h, w = obs.shape
            obs = obs.reshape(b, t * c, h, w)
            sigma = self.sample_sigma_training(b, self.device)
            noisy_next_obs = self.apply_noise(next_obs, sigma, self.cfg.sigma_offset_noise)
            cs = self._training_conditioners(sigma)
            model_output = self.model_output_with_grad(noisy_next_obs, obs, act, cs)
            c_in, c_out, c_skip, c_noise = (v.reshape(-1, 1, 1, 1) for v in cs)
            target = (next_obs - c_skip * noisy_next_obs) / c_out
            # F.mse_loss(model_output[mask], target[mask]) (reference :114) without the boolean gather: the gather's
            # output shape depends on device data, i.e. a host synchronisation per segment step (and no hipGraph capture
            # of the training step, train_graph.py); same value up to the fp32 summation order; 0 / 0 = nan for an
            # all-padding batch, like the mean over an empty selection
            w = mask.to(model_output.dtype).reshape(-1, 1, 1, 1)
            per = model_output[0].numel()
            loss = loss + ((model_output - target).square() * w).sum() / (w.sum() * per)
            denoised = self.wrap_model_output(noisy_next_obs, model_output.detach(), sigma, cond4=torch.stack(cs, 1).contiguous())
            all_obs[:, n + i] = denoised
        loss = loss / seq_length
        return loss, {"loss_denoising": loss.detach()}
