"""EDM denoiser (reference models/diffusion/denoiser.py), inference path on hand-written HIP.

`denoise` keeps the reference signature `(noisy_next_obs, sigma, obs, act) -> denoised`
with NCHW tensors; sigma may be a 0-dim tensor, a (B,) tensor (Heun's second evaluation,
diffusion_sampler.py:53) or a python float.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple, Union

import torch
from torch import Tensor, nn

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

    @property
    def device(self) -> torch.device:
        return self.inner_model.noise_emb.weight.device

    def setup_training(self, cfg: SigmaDistributionConfig) -> None:
        assert self.sample_sigma_training is None

        def sample_sigma(n: int, device: torch.device):
            s = torch.randn(n, device=device) * cfg.scale + cfg.loc
            return s.exp().clip(cfg.sigma_min, cfg.sigma_max)

        self.sample_sigma_training = sample_sigma

    def _edm(self) -> nv.EdmCfg:
        return nv.EdmCfg(float(self.cfg.sigma_data), float(self.cfg.sigma_offset_noise))

    def _sigma_arg(self, sigma: Union[Tensor, float], n: int) -> Tuple[Tensor, int]:
        if not torch.is_tensor(sigma):
            sigma = torch.tensor(float(sigma), dtype=torch.float32)
        sigma = sigma.to(device=self.device, dtype=torch.float32)
        if sigma.numel() == 1:
            return sigma.reshape(1).contiguous(), 0
        assert sigma.numel() == n, "sigma must be a scalar or one value per sample"
        return sigma.reshape(n).contiguous(), 1

    @torch.no_grad()
    def compute_model_output(self, noisy_next_obs: Tensor, obs: Tensor, act: Tensor, sigma: Union[Tensor, float],
                             naive: Optional[bool] = None) -> Tensor:
        """F = inner_model(x * c_in, c_noise, obs / sigma_data, act)  (reference :74-77).
        Takes sigma instead of the reference's Conditioners: the conditioners are computed
        on the device from sigma, in fp32 with the reference's op order."""
        n, cx, h, w = noisy_next_obs.shape
        cobs = obs.shape[1]
        sig, stride = self._sigma_arg(sigma, n)
        edm = self._edm()
        cpad = (cx + cobs + 15) // 16 * 16
        packed = torch.empty(n, h, w, cpad, device=self.device, dtype=torch.float32)
        # NOTE: pointers are only taken from tensors bound to a name -- a temporary made inside the
        # argument list would be freed (and its block re-used) before the kernel is even launched.
        xc, oc = noisy_next_obs.contiguous(), obs.contiguous()
        nv.check(nv.lib().dmd_edm_pack_input(nv.fptr(xc), nv.fptr(oc), nv.fptr(sig),
                                             stride, edm, nv.fptr(packed), n, cx, cobs, h, w, cpad, nv.stream()),
                 "dmd_edm_pack_input")
        cond = self.inner_model.cond_vector(sig, stride, act, edm)
        return self.inner_model.run(packed, cond, naive)

    @torch.no_grad()
    def wrap_model_output(self, noisy_next_obs: Tensor, model_output: Tensor, sigma: Union[Tensor, float]) -> Tensor:
        """quantise(c_skip * x + c_out * F)  (reference :79-84)."""
        n = noisy_next_obs.shape[0]
        sig, stride = self._sigma_arg(sigma, n)
        x = noisy_next_obs.contiguous()
        f = model_output.contiguous()
        out = torch.empty_like(x)
        nv.check(nv.lib().dmd_edm_denoised(nv.fptr(x), nv.fptr(f), nv.fptr(sig), stride, self._edm(),
                                           nv.fptr(out), n, x[0].numel(), nv.stream()), "dmd_edm_denoised")
        return out

    @torch.no_grad()
    def denoise(self, noisy_next_obs: Tensor, sigma: Union[Tensor, float], obs: Tensor, act: Tensor) -> Tensor:
        f = self.compute_model_output(noisy_next_obs, obs, act, sigma)
        return self.wrap_model_output(noisy_next_obs, f, sigma)

    def forward(self, batch):
        raise NotImplementedError(
            "Denoiser.forward (the denoiser TRAINING loss, reference denoiser.py:93-122) needs the U-Net backward "
            "kernels: SURVEY.md §8(f) row 2, not built yet")
