"""Reward / termination model (reference models/rew_end_model.py): inference and training step on the HIP kernels.

`predict_rew_end` is on every WorldModelEnv.step (world_model_env.py:95-105) and in the
initial-condition burn-in (:123-124).  Encoder = the same fused conv / FiLM / attention
kernels as the denoiser at C = 32; the LSTM step is two MFMA GEMMs + one pointwise kernel.
`forward(batch)` is the training loss of trainer.py:365 (recorded forward + hand-written backward, see unet_train.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import engine as E
from . import native as nv
from .blocks import Downsample, FilmTable, ResBlocks, RunCtx, conv3x3


@dataclass
class RewEndModelConfig:
    lstm_dim: int
    img_channels: int
    img_size: int
    cond_channels: int
    depths: List[int]
    channels: List[int]
    attn_depths: List[int]
    num_actions: Optional[int] = None


def init_lstm(model: nn.Module) -> None:
    """xavier / orthogonal / forget-bias-1 (reference utils.py:184-196)."""
    for name, p in model.named_parameters():
        if "weight_ih" in name:
            nn.init.xavier_uniform_(p.data)
        elif "weight_hh" in name:
            nn.init.orthogonal_(p.data)
        elif "bias_ih" in name:
            p.data.fill_(0)
            n = p.size(0)
            p.data[(n // 4):(n // 2)].fill_(1)
        elif "bias_hh" in name:
            p.data.fill_(0)


class RewEndEncoder(nn.Module):
    def __init__(self, in_channels: int, cond_channels: int, depths: List[int], channels: List[int],
                 attn_depths: List[int]) -> None:
        super().__init__()
        assert len(depths) == len(channels) == len(attn_depths)
        self.conv_in = conv3x3(in_channels, channels[0])
        blocks = []
        for i, n in enumerate(depths):
            c1, c2 = channels[max(0, i - 1)], channels[i]
            blocks.append(ResBlocks([c1] + [c2] * (n - 1), [c2] * n, cond_channels, bool(attn_depths[i])))
        blocks.append(ResBlocks([channels[-1]] * 2, [channels[-1]] * 2, cond_channels, True))
        self.blocks = nn.ModuleList(blocks)
        self.downsamples = nn.ModuleList([nn.Identity()] + [Downsample(c) for c in channels[:-1]] + [nn.Identity()])

    def run(self, ctx: RunCtx, x_nhwc16: Tensor, valid: Optional[Tuple[int, int]] = None) -> E.Act:
        """valid = (h, w): the image is that part of the zero-padded buffer (engine.padded_extent; inference only)"""
        x = E.conv2d([(E.Act(x_nhwc16, needs_grad=False, valid=valid), nv.PROLOGUE_NONE, None)], ctx.cache.conv_weight(self.conv_in),
                     ctx.cache.conv_bias(self.conv_in), self.conv_in.out_channels, naive=ctx.naive, w_f16=ctx.w16(self.conv_in),
                     module=self.conv_in)
        tail = len(self.blocks) - 2  # the last level and the final attention group both run at the deepest resolution
        for i, (blocks, down) in enumerate(zip(self.blocks, self.downsamples)):
            if not isinstance(down, nn.Identity):
                x = down.run(ctx, x)
            if i == tail and self._chain_eligible(ctx, x):
                return self._run_lowres_chain(ctx, x)  # both groups as ONE launch (dmd_lowres.hip, 32-channel kernel)
            x, _ = blocks.run(ctx, x)
        return x

    def _chain_blocks(self):
        return list(self.blocks[-2].resblocks) + list(self.blocks[-1].resblocks)

    def _chain_eligible(self, ctx: RunCtx, x: E.Act) -> bool:
        from . import blocks as BL

        if not (int(BL.LOWRES_CHAIN) & 2) or ctx.precision != "f16x2" or ctx.naive or E.TAPE is not None or E._USE_NAIVE:
            return False
        if tuple(x.shape[1:]) != (8, 8, 32) or x.valid is not None:
            return False
        blks = self._chain_blocks()
        return len(blks) <= nv.CHAIN_MAX_BLOCKS and all(
            b.conv1.in_channels == 32 and b.conv1.out_channels == 32 and isinstance(b.proj, nn.Identity) for b in blks)

    def _run_lowres_chain(self, ctx: RunCtx, x: E.Act) -> E.Act:
        import ctypes as C

        cache = ctx.cache
        blks = self._chain_blocks()
        p = nv.LowresChainParams()
        n = x.shape[0]
        out = torch.empty_like(x.t)
        p.N, p.nblocks, p.input_save_slot = n, len(blks), -1
        p.x, p.out = nv.ptr(x.t), nv.ptr(out)
        p.table, p.table_stride = nv.ptr(ctx.table), ctx.table.stride(0)
        keep = []  # (the caches own these tensors; the list documents what the parameter block points at)
        for i, b in enumerate(blks):
            cb = p.blocks[i]
            cb.skip_slot = cb.save_slot = -1
            o1, o2 = ctx.film.offset[id(b.norm1)], ctx.film.offset[id(b.norm2)]
            cb.film1_mul[0], cb.film1_add[0] = o1, o1 + 32
            cb.film2_mul, cb.film2_add = o2, o2 + 32
            ts = [cache.conv_weight_f16x2(b.conv1), cache.conv_weight_f16x2(b.conv2), cache.conv_bias(b.conv1), cache.conv_bias(b.conv2)]
            cb.w1, cb.w2, cb.b1, cb.b2 = (nv.ptr(t) for t in ts)
            if not isinstance(b.attn, nn.Identity):
                a = b.attn
                cb.has_attn = 1
                qkv = [cache.get(a.qkv_proj.weight, f"qkv16[{k}]", lambda w, k=k: nv.pack_conv_weight_f16x2(w.detach().float()[32 * k:32 * k + 32].contiguous()))
                       for k in range(3)]
                ta = qkv + [cache.conv_weight_f16x2(a.out_proj), cache.f32(a.norm.norm.weight), cache.f32(a.norm.norm.bias),
                            cache.f32(a.qkv_proj.bias), cache.conv_bias(a.out_proj)]
                cb.wq, cb.wk, cb.wv, cb.wo, cb.gn_gamma, cb.gn_beta, cb.bqkv, cb.bo = (nv.ptr(t) for t in ta)
                ts += ta
            keep.append(ts)
        nv.check(nv.lib().dmd_lowres_chain32(C.byref(p), nv.stream()), "dmd_lowres_chain32")
        del keep
        return E.Act(out)


class RewEndModel(nn.Module):
    def __init__(self, cfg: RewEndModelConfig) -> None:
        super().__init__()
        self.cfg = cfg
        self.encoder = RewEndEncoder(2 * cfg.img_channels, cfg.cond_channels, cfg.depths, cfg.channels, cfg.attn_depths)
        self.act_emb = nn.Embedding(cfg.num_actions, cfg.cond_channels)
        input_dim_lstm = cfg.channels[-1] * (cfg.img_size // 2 ** (len(cfg.depths) - 1)) ** 2
        self.lstm = nn.LSTM(input_dim_lstm, cfg.lstm_dim, batch_first=True)
        self.head = nn.Sequential(nn.Linear(cfg.lstm_dim, cfg.lstm_dim), nn.SiLU(), nn.Linear(cfg.lstm_dim, 3 + 2, bias=False))
        init_lstm(self.lstm)
        self._cache = E.PackCache()
        self._film: Optional[FilmTable] = None
        self._train_params: Optional[List[nn.Parameter]] = None

    @property
    def device(self) -> torch.device:
        return self.act_emb.weight.device

    def _lstm_in_weight(self) -> Tensor:
        """weight_ih_l0 with its (e, h, w)-major columns (rew_end_model.py:52) permuted to the
        encoder's NHWC flatten order (h, w, e), so no activation transpose is needed."""
        def pack(w: Tensor) -> Tensor:
            e = self.cfg.channels[-1]
            s = int(round((w.shape[1] // e) ** 0.5))
            return w.detach().float().reshape(w.shape[0], e, s, s).permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
        return self._cache.get(self.lstm.weight_ih_l0, "nhwc_cols", pack)

    @torch.no_grad()
    def predict_rew_end(self, obs: Tensor, act: Tensor, next_obs: Tensor,
                        hx_cx: Optional[Tuple[Tensor, Tensor]] = None) -> Tuple[Tensor, Tensor, Tuple[Tensor, Tensor]]:
        b, t, c, h, w = obs.shape
        dev = obs.device
        nv.check_current_device(dev)  # (ctypes launches go to the CURRENT device's stream)
        x = torch.cat((obs.reshape(b * t, c, h, w), next_obs.reshape(b * t, c, h, w)), dim=1)
        # sizes whose levels leave the kernels' tile grid (the reference runs any size its three stride-2 convolutions halve
        # evenly, rew_end_model.py:33: 72 -> 36 -> 18 -> 9): the VALID EXTENT of a zero-padded buffer, like the denoiser
        nd = len(self.encoder.downsamples) - 2
        hp, wp = E.padded_extent(h, w, nd)
        valid = None if (hp, wp) == (h, w) else (h, w)
        if valid is not None:
            assert h % 2 ** nd == 0 and w % 2 ** nd == 0, f"RewEndModel: {h}x{w} is not a multiple of {2 ** nd} (the reference's own constraint)"
            x = torch.nn.functional.pad(x, (0, wp - w, 0, hp - h))
        x16 = E.nchw_to_nhwc(x, 16)
        cond = self._cache.f32(self.act_emb.weight)[act.reshape(b * t)].contiguous()  # embedding gather (plumbing)
        if self._film is None:
            self._film = FilmTable(self.encoder)
        ctx = RunCtx(self._cache, self._film, self._film.compute(cond))
        fa = self.encoder.run(ctx, x16, valid)
        ft = fa.t if fa.valid is None else fa.t[:, :fa.valid[0], :fa.valid[1]].contiguous()
        feat = ft.reshape(b, t, -1)  # NHWC flatten; weight columns permuted to match
        hd = self.cfg.lstm_dim
        if hx_cx is None:
            hx = torch.zeros(b, hd, device=dev)
            cx = torch.zeros(b, hd, device=dev)
        else:
            hx, cx = hx_cx[0][0].contiguous(), hx_cx[1][0].contiguous()
        w_ih, w_hh = self._lstm_in_weight(), self._cache.f32(self.lstm.weight_hh_l0)
        b_ih, b_hh = self._cache.f32(self.lstm.bias_ih_l0), self._cache.f32(self.lstm.bias_hh_l0)
        # input projections of all T steps in one GEMM, then the sequential recurrence
        gates_x = E.linear(feat.reshape(b * t, -1), w_ih, b_ih).reshape(b, t, 4 * hd)
        ys = []
        for i in range(t):
            g = gates_x[:, i].contiguous()
            E.linear(hx, w_hh, b_hh, out=g, accumulate=True)
            hn, cn = torch.empty_like(hx), torch.empty_like(cx)
            nv.check(nv.lib().dmd_lstm_pointwise(nv.fptr(g), nv.fptr(cx), nv.fptr(hn), nv.fptr(cn), b, hd, nv.stream()),
                     "dmd_lstm_pointwise")
            hx, cx = hn, cn
            ys.append(hx)
        y = torch.stack(ys, dim=1).reshape(b * t, hd)
        y = E.linear(y, self._cache.f32(self.head[0].weight), self._cache.f32(self.head[0].bias), silu=True)
        logits = E.linear(y, self._cache.f32(self.head[2].weight), None).reshape(b, t, -1)
        return logits[:, :, :-2], logits[:, :, -2:], (hx.unsqueeze(0), cx.unsqueeze(0))

    # -- training step (reference rew_end_model.py:57-90) ----------------------------------------------------------
    def logits_with_grad(self, obs: Tensor, act: Tensor, next_obs: Tensor, precision: Optional[str] = None) -> Tensor:
        """predict_rew_end's logits (B, T, 5), from a zero LSTM state, differentiable w.r.t. every parameter.  Encoder:
        the inference kernels under a recorded tape with the hand-written backward (unet_train.EncoderTrainFn);
        LSTM over the segment, the head and the FiLM-table GEMM: lstm_native.LinearFn / LstmStepFn (dmd_linear forward and
        backward); the action embedding (a gather) is a torch op under autograd."""
        import torch.nn.functional as F
        from . import unet_train as UT
        from .blocks import AdaGroupNorm
        from .lstm_native import LinearFn, LstmStepFn

        b, t, c, h, w = obs.shape
        x = torch.cat((obs.reshape(b * t, c, h, w), next_obs.reshape(b * t, c, h, w)), dim=1).detach()
        x16 = E.nchw_to_nhwc(x, 16)
        cond = self.act_emb(act.reshape(b * t))
        if self._film is None:
            self._film = FilmTable(self.encoder)
        film = self._film
        # (the FiLM table of all AdaGroupNorm layers as ONE GEMM, forward and backward on dmd_linear like the denoiser's
        #  training step; autograd carries the table gradient into the concatenated weights -- torch.cat's backward is a split)
        table = LinearFn.apply(self._cache, cond, torch.cat([m.linear.weight for m in film.norms], dim=0),
                               torch.cat([m.linear.bias for m in film.norms], dim=0))
        if self._train_params is None:
            skip = {id(p) for m in self.encoder.modules() if isinstance(m, AdaGroupNorm) for p in m.parameters()}
            self._train_params = [p for p in self.encoder.parameters() if id(p) not in skip]
        precision = precision or UT.TRAIN_PRECISION

        def run(tab: Tensor) -> Tensor:
            return self.encoder.run(RunCtx(self._cache, film, tab, precision=precision), x16).t

        feat = UT.EncoderTrainFn.apply(run, self._cache, table, precision, *self._train_params)  # (b t, s, s, e) NHWC
        hd, e = self.cfg.lstm_dim, self.cfg.channels[-1]
        w_ih = self.lstm.weight_ih_l0
        s_ = feat.shape[1]
        w_ih_nhwc = w_ih.reshape(4 * hd, e, s_, s_).permute(0, 2, 3, 1).reshape(4 * hd, -1)  # (e h w) -> (h w e) columns
        gx = LinearFn.apply(self._cache, feat.reshape(b * t, -1), w_ih_nhwc, self.lstm.bias_ih_l0).reshape(b, t, 4 * hd)
        hx = torch.zeros(b, hd, device=obs.device)
        cx = torch.zeros(b, hd, device=obs.device)
        ys = []
        for i in range(t):
            hx, cx = LstmStepFn.apply(self._cache, gx[:, i], hx, cx, self.lstm.weight_hh_l0, self.lstm.bias_hh_l0)
            ys.append(hx)
        y = torch.stack(ys, dim=1).reshape(b * t, hd)
        y = F.silu(LinearFn.apply(self._cache, y, self.head[0].weight, self.head[0].bias))
        return LinearFn.apply(self._cache, y, self.head[2].weight, None).reshape(b, t, -1)

    def forward(self, batch):
        """Cross-entropy losses of the reward (sign-clipped, 3 classes) and termination (2 classes) heads over the
        unpadded steps of a segment, with the true final observation put back where the episode ended (reference
        rew_end_model.py:57-90).  Returns (loss, metrics) with the reference's metric names."""
        import torch.nn.functional as F

        nv.check_current_device(batch.obs.device)
        obs = batch.obs[:, :-1]
        act = batch.act[:, :-1]
        next_obs = batch.obs[:, 1:]
        rew = batch.rew[:, :-1]
        end = batch.end[:, :-1]
        mask = batch.mask_padding[:, :-1]
        dead = end.bool().any(dim=1)
        if dead.any():
            final_obs = torch.stack([i["final_observation"] for i, d in zip(batch.info, dead) if d]).to(obs.device)
            next_obs[dead, end[dead].argmax(dim=1)] = final_obs  # writes through to batch.obs, like the reference
        logits = self.logits_with_grad(obs, act, next_obs)
        logits_rew, logits_end = logits[:, :, :-2][mask], logits[:, :, -2:][mask]
        target_rew = rew[mask].sign().long().add(1)
        target_end = end[mask]
        loss_rew = F.cross_entropy(logits_rew, target_rew)
        loss_end = F.cross_entropy(logits_end, target_end)
        loss = loss_rew + loss_end
        metrics = {
            "loss_rew": loss_rew.detach(),
            "loss_end": loss_end.detach(),
            "loss_total": loss.detach(),
            "confusion_matrix": {
                "rew": confusion_matrix(logits_rew, target_rew, 3),
                "end": confusion_matrix(logits_end, target_end, 2),
            },
        }
        return loss, metrics


def confusion_matrix(logits: Tensor, target: Tensor, num_classes: int) -> Tensor:
    """torcheval.metrics.functional.multiclass_confusion_matrix(logits, target, num_classes) (a pinned dependency of
    the reference, rew_end_model.py:8,84-85; not vendored): entry [i, j] counts samples of true class i predicted
    (argmax) as class j, int64."""
    pred = logits.detach().argmax(dim=1)
    return torch.bincount(target.detach().long() * num_classes + pred, minlength=num_classes ** 2).reshape(num_classes, num_classes)
