"""U-Net wrapper of the EDM denoiser (reference models/diffusion/inner_model.py:23-49), native."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import engine as E
from . import native as nv
from .blocks import FilmTable, FourierFeatures, GroupNorm, RunCtx, UNet, conv3x3

_DEBUG_CHECKS = os.environ.get("DIAMOND_DEBUG", "0") == "1"  # extra host-side validation (adds device syncs)


@dataclass
class InnerModelConfig:
    img_channels: int
    num_steps_conditioning: int
    cond_channels: int
    depths: List[int]
    channels: List[int]
    attn_depths: List[bool]
    num_actions: Optional[int] = None


class InnerModel(nn.Module):
    def __init__(self, cfg: InnerModelConfig) -> None:
        super().__init__()
        self.cfg = cfg
        self.noise_emb = FourierFeatures(cfg.cond_channels)
        self.act_emb = nn.Sequential(
            nn.Embedding(cfg.num_actions, cfg.cond_channels // cfg.num_steps_conditioning), nn.Flatten())
        self.cond_proj = nn.Sequential(
            nn.Linear(cfg.cond_channels, cfg.cond_channels), nn.SiLU(), nn.Linear(cfg.cond_channels, cfg.cond_channels))
        self.conv_in = conv3x3((cfg.num_steps_conditioning + 1) * cfg.img_channels, cfg.channels[0])
        self.unet = UNet(cfg.cond_channels, cfg.depths, cfg.channels, cfg.attn_depths)
        self.norm_out = GroupNorm(cfg.channels[0])
        self.conv_out = conv3x3(cfg.channels[0], cfg.img_channels)
        nn.init.zeros_(self.conv_out.weight)
        # host-side launch state (not parameters)
        self._cache = E.PackCache()
        self._film: Optional[FilmTable] = None

    # -- native pieces -------------------------------------------------------------------
    def cond_vector(self, cond: Tensor, cond_stride: int, act: Tensor, act_head: int = 0) -> Tensor:
        """cond_proj(noise_emb(c_noise) + act_emb(act))  (reference :45); c_noise is entry 3 of
        the per-sample conditioner array (Denoiser.compute_conditioners).  `act` (N, T) int64 may be a ring
        whose logical step 0 is column `act_head` (WorldModelEnv keeps its action context that way)."""
        assert act.dtype == torch.long and act.ndim == 2, f"act must be an int64 (N, T) tensor, got {act.dtype} {tuple(act.shape)}"
        n, t = act.shape
        emb = self.act_emb[0].weight
        if _DEBUG_CHECKS:  # nn.Embedding raises on an out-of-range index; the kernel only clamps (one host sync)
            lo, hi = int(act.min()), int(act.max())
            if lo < 0 or hi >= emb.shape[0]:
                raise IndexError(f"action index out of range [0, {emb.shape[0]}): min {lo}, max {hi}")
        half = self.noise_emb.weight.shape[1]
        x = torch.empty(n, 2 * half, device=act.device, dtype=torch.float32)
        act = act.contiguous()
        fw, ew = self._cache.f32(self.noise_emb.weight), self._cache.f32(emb)
        nv.check(nv.lib().dmd_cond_embed(nv.fptr(cond), cond_stride, nv.fptr(fw), nv.ptr(act), nv.fptr(ew), nv.fptr(x), n,
                                         half, t, emb.shape[1], act_head, emb.shape[0], nv.stream()), "dmd_cond_embed")
        return self._cond_proj(x)

    def _cond_proj(self, x: Tensor) -> Tensor:
        l0, l2 = self.cond_proj[0], self.cond_proj[2]
        y = E.linear(x, self._cache.f32(l0.weight), self._cache.f32(l0.bias), silu=True)
        return E.linear(y, self._cache.f32(l2.weight), self._cache.f32(l2.bias))

    def film_tables(self, conds: Sequence[Tuple[Tensor, int]], act: Tensor, act_head: int = 0) -> List[Tensor]:
        """The FiLM tables of SEVERAL forwards that share `act` (the denoising steps of one sampled frame: one sigma each) as
        three launches of dmd_linear instead of three per forward: the embeddings of all steps are rows of one (K * N, .) matrix.
        dmd_linear sums an output row in the same order whatever the row count (csrc/dmd_linear.hip), so table k is bitwise
        `FilmTable.compute(cond_vector(conds[k], ...))`.  Returns K row-slices of one buffer."""
        assert act.dtype == torch.long and act.ndim == 2
        n, t = act.shape
        emb = self.act_emb[0].weight
        if _DEBUG_CHECKS:
            lo, hi = int(act.min()), int(act.max())
            if lo < 0 or hi >= emb.shape[0]:
                raise IndexError(f"action index out of range [0, {emb.shape[0]}): min {lo}, max {hi}")
        half = self.noise_emb.weight.shape[1]
        k = len(conds)
        x = torch.empty(k * n, 2 * half, device=act.device, dtype=torch.float32)
        act = act.contiguous()
        fw, ew = self._cache.f32(self.noise_emb.weight), self._cache.f32(emb)
        for i, (cond, stride) in enumerate(conds):
            xi = x[i * n:(i + 1) * n]
            nv.check(nv.lib().dmd_cond_embed(nv.fptr(cond), stride, nv.fptr(fw), nv.ptr(act), nv.fptr(ew), nv.fptr(xi), n,
                                             half, t, emb.shape[1], act_head, emb.shape[0], nv.stream()), "dmd_cond_embed")
        if self._film is None:
            self._film = FilmTable(self.unet)
        table = self._film.compute(self._cond_proj(x))
        return [table[i * n:(i + 1) * n] for i in range(k)]

    def run(self, packed_in: Tensor, cond: Optional[Tensor], naive: Optional[bool] = None, precision: Optional[str] = None,
            table: Optional[Tensor] = None, valid: Optional[Tuple[int, int]] = None) -> Tensor:
        """packed_in: NHWC16 [obs/sigma_data | noisy*c_in | 0]; returns F as NCHW (N,3,H,W).
        table: the batched FiLM table if the caller already has it (the training path computes it under autograd).
        valid = (h, w): the image is that part of the (H, W) buffer, the rest of which is ZERO (engine.padded_extent): the
        reference's forward for sizes whose U-Net levels are not multiples of the kernels' tiles, including its
        pad-to-2**num_down / crop (inner_model.py:44-49, blocks.py:227-229,247).  The returned (N, 3, H, W) is meaningful
        in [:h, :w] only."""
        if self._film is None:
            self._film = FilmTable(self.unet)
        if table is None:
            table = self._film.compute(cond)
        ctx = RunCtx(self._cache, self._film, table, naive, precision)
        x = E.conv2d([(E.Act(packed_in, needs_grad=False, valid=valid), nv.PROLOGUE_NONE, None)], self._cache.conv_weight(self.conv_in),
                     self._cache.conv_bias(self.conv_in), self.conv_in.out_channels, naive=naive, w_f16=ctx.w16(self.conv_in),
                     module=self.conv_in)
        padded = None
        if valid is not None:
            # UNet.forward pads conv_in's output with zeros to a multiple of 2**num_down (blocks.py:227-229) and treats the
            # padding as data from then on: zero those rows / columns (conv_in's partial sums are unaffected by zeros) and
            # widen the valid extent to the padded size
            m = 2 ** self.unet._num_down
            h, w = valid
            padded = ((h + m - 1) // m * m, (w + m - 1) // m * m)
            if padded != valid:
                x.t[:, h:padded[0], :padded[1]] = 0
                x.t[:, :h, w:padded[1]] = 0
                x.valid = padded
        x = self.unet.run(ctx, x)
        if padded is not None and padded != valid:
            # ... and crops before norm_out (blocks.py:247): GroupNorm statistics of the cropped tensor
            x = E.gn_stats(x.t, valid)
        co = self.conv_out
        w16 = self._cache.conv_weight_f16x2_head(co) if (ctx.precision == "f16x2" and not naive) else None
        cpad = 32 if w16 is not None else None  # the split kernel's 32-cout instance, real channels stored as NCHW
        return E.conv2d([(x, nv.PROLOGUE_NORM_SILU, self.norm_out.spec(ctx))], self._cache.conv_weight(co, cpad),
                        self._cache.conv_bias(co, cpad), co.out_channels, want_stats=False, out_nchw=True, naive=naive,
                        fast_math=ctx.fast_math, w_f16=w16, cout_padded=cpad, module=co).t
