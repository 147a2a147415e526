"""Actor-critic conv encoder on hand-written HIP kernels, forward AND backward
(reference models/actor_critic.py:101-113 `ActorCriticEncoder`, blocks.py:116-123 `SmallResBlock`).

One `torch.autograd.Function` spans the whole encoder: Conv3x3 -> n x [skip(x) +
Conv3x3(SiLU(GroupNorm(x))), MaxPool2] -> flatten.  The forward saves only the block inputs
(with their GroupNorm statistics) and the pooling argmax; the backward recomputes the
activations inside the wgrad kernel's staging pass.  Kernel map:

  forward   dmd_conv2d (GN+SiLU prologue, bias + residual epilogue), dmd_maxpool2 (+ stats of the
            pooled tensor for the next GroupNorm)
  backward  dmd_maxpool2_bwd -> dmd_conv2d_wgrad (dW, db) -> dmd_conv2d on the flipped/transposed
            weight (dgrad) -> dmd_gn_silu_bwd (dx, dgamma, dbeta; adds the skip-branch gradient)

The LSTM cell and the two heads downstream run on dmd_linear / dmd_lstm_pointwise(_bwd): lstm_native.LstmHeadsFn.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import engine as E
from . import native as nv
from .engine import Act, NormSpec


def _maxpool(y: Tensor, valid: Optional[Tuple[int, int]] = None) -> Tuple[Act, Tensor]:
    """MaxPool2d(2) (floor: an odd valid extent loses its last row / column, like F.max_pool2d).  valid: the part of the
    buffer that exists; the pooled tensor is the (vh // 2, vw // 2) part of a buffer half the size, and its GroupNorm
    statistics count that part only."""
    n, h, w, c = y.shape
    out = torch.empty(n, h // 2, w // 2, c, device=y.device, dtype=torch.float32)
    arg = torch.empty(n, h // 2, w // 2, c, device=y.device, dtype=torch.uint8)
    stats = E.new_stats(n, c, 1, y.device) if (c % nv.GN_GROUP == 0 and valid is None) else None
    nv.check(nv.lib().dmd_maxpool2(nv.fptr(y), nv.fptr(out), nv.ptr(arg), nv.ptr(stats), n, h, w, c, nv.stream()), "dmd_maxpool2")
    if valid is not None:
        v2 = (valid[0] // 2, valid[1] // 2)
        return (E.gn_stats(out, v2) if c % nv.GN_GROUP == 0 else Act(out, valid=v2)), arg
    return Act(out, stats, 1 if stats is not None else 0), arg


def _maxpool_bwd(dp: Tensor, arg: Tensor) -> Tensor:
    n, ho, wo, c = dp.shape
    dx = torch.empty(n, ho * 2, wo * 2, c, device=dp.device, dtype=torch.float32)
    nv.check(nv.lib().dmd_maxpool2_bwd(nv.fptr(dp), nv.ptr(arg), nv.fptr(dx), n, ho * 2, wo * 2, c, nv.stream()), "dmd_maxpool2_bwd")
    return dx


class WgradBatch:
    """Weight gradients whose reductions wait for ONE launch per 32 of them (`dmd_wgrad_reduce_jobs`, ABI v10): the backward of a
    denoiser training step holds ~60 weight gradients nobody reads before it is over, and their reductions were ~140 launches
    of a few microseconds of work.  The sums are formed in the undeferred order: bit-identical gradients."""

    def __init__(self) -> None:
        self.jobs: List[nv.WgradReduceJob] = []
        self._keep: List[Tensor] = []  # the workspaces holding the partials (and the outputs) until flush()

    def add(self, job: "nv.WgradReduceJob", *keep: Tensor) -> None:
        self.jobs.append(job)
        self._keep.extend(keep)

    def flush(self) -> None:
        if self.jobs:
            table = (nv.WgradReduceJob * len(self.jobs))(*self.jobs)
            nv.check(nv.lib().dmd_wgrad_reduce_jobs(table, len(self.jobs), nv.stream()), "dmd_wgrad_reduce_jobs")
        self.jobs, self._keep = [], []


# (output channels / 16, input channels / 16) the weight-gradient kernel is instantiated for (csrc/dmd_backward.hip:
# dmd_conv2d_wgrad's dispatch): the shapes of the default configuration's networks
_WGRAD_INSTANCES = {9: {(2, 1), (4, 1), (1, 4), (2, 2), (4, 2), (4, 4)}, 1: {(4, 2), (2, 2), (4, 4)}}


def _wgrad_instance(cout: int, cin: int, taps: int) -> bool:
    return cout % 16 == 0 and cin % 16 == 0 and (cout // 16, cin // 16) in _WGRAD_INSTANCES[taps]


def _pad_channels(t: Tensor, c0: int, c1: int, to: int) -> Tensor:
    """Channels [c0, c1) of an NHWC tensor as a contiguous tensor of `to` channels (zeros behind the slice)."""
    if c1 - c0 == to:
        return t[..., c0:c1].contiguous()
    out = torch.zeros(t.shape[:-1] + (to,), device=t.device, dtype=t.dtype)
    out[..., :c1 - c0] = t[..., c0:c1]
    return out


def _norm_slice(x: Act, spec: NormSpec, c0: int, c1: int, to: int) -> Tuple[Tensor, int, NormSpec]:
    """Statistics and multiplicative / additive parameters of channels [c0, c1) of a normalised source, zero-padded to `to`
    channels: (partial sums, tiles, spec).  A padded group has sums 0 and parameters 0: its activated value is 0."""
    g = nv.GN_GROUP
    assert c0 % g == 0 and (c1 - c0) % g == 0 and to % g == 0 and x.stats is not None, \
        f"normalised source of {x.C} channels cut at [{c0}, {c1}): not whole GroupNorm groups"
    stats = x.stats[:, c0 // g:c1 // g]
    if to != c1 - c0:
        stats = torch.cat([stats, torch.zeros(stats.shape[0], (to - (c1 - c0)) // g, *stats.shape[2:], device=stats.device, dtype=stats.dtype)], 1)

    def cut(t: Optional[Tensor], stride: int):
        if t is None:
            return None, 0
        rows = t if t.ndim == 2 else t[None]
        rows = rows[:, c0:c1] if stride != 0 else rows[:1, c0:c1]
        out = torch.zeros(rows.shape[0], to, device=t.device, dtype=torch.float32)
        out[:, :c1 - c0] = rows
        return out, (to if stride != 0 else 0)

    mul, ms = cut(spec.mul, spec.mul_stride)
    add, as_ = cut(spec.add, spec.add_stride)
    return stats.contiguous(), x.tiles, NormSpec(mul, add, ms, as_, spec.plus_one)


def _wgrad_tiled(x: Act, prologue: int, spec: Optional[NormSpec], dy: Tensor, taps: int, cin_real: int, want_bias: bool, split: bool,
                 batch: Optional["WgradBatch"], dw_out: Optional[Tensor], c0: int, db_out: Optional[Tensor]):
    """Weight gradient of a convolution the kernel has no instance for (networks wider or narrower than the default
    configuration's 32 / 64 channels: the reference takes any `channels` list, blocks.py:183-222, actor_critic.py:101-113): the
    gradient of output-channel block i w.r.t. input-channel block j depends on those two blocks only, so the (Cout, Cin) plane is
    tiled with the 64 x 64 instance on contiguous, zero-padded channel slices (whole GroupNorm groups of a normalised source,
    with their partial sums and parameters).  Reduced per tile (nothing deferred): the correct path for such shapes, not a fast one."""
    n, h, w, cout = dy.shape
    k = 3 if taps == 9 else 1
    T = 64
    if batch is not None:
        dw, db = dw_out, db_out
    else:
        dw = torch.empty(cout, cin_real, k, k, device=dy.device, dtype=torch.float32)
        db = torch.empty(cout, device=dy.device, dtype=torch.float32) if want_bias else None
        c0 = 0
    for ci0 in range(0, cin_real, T):
        ci1 = min(cin_real, ci0 + T)
        cw = (ci1 - ci0 + 15) // 16 * 16  # (the source's own padding: conv_in's 15 channels travel as 16)
        if prologue == nv.PROLOGUE_NONE:
            xt, spec_t = Act(_pad_channels(x.t, ci0, min(x.C, ci0 + cw), T), valid=x.valid), None
        else:
            stats, tiles, spec_t = _norm_slice(x, spec, ci0, ci1, T)
            xt = Act(_pad_channels(x.t, ci0, ci1, T), stats, tiles, valid=x.valid)
        for co0 in range(0, cout, T):
            co1 = min(cout, co0 + T)
            dw_t, db_t = _wgrad(xt, prologue, spec_t, _pad_channels(dy, co0, co1, T), taps, ci1 - ci0,
                                want_bias=db is not None and ci0 == 0, split=split)
            dw[co0:co1, c0 + ci0:c0 + ci1] = dw_t[:co1 - co0]
            if db is not None and ci0 == 0:
                db[co0:co1] = db_t[:co1 - co0]
    return dw, db


def _wgrad(x: Act, prologue: int, spec: Optional[NormSpec], dy: Tensor, taps: int, cin_real: int,
           want_bias: bool = True, split: bool = False, batch: Optional[WgradBatch] = None, dw_out: Optional[Tensor] = None,
           c0: int = 0, db_out: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """dW, db of a convolution.  split: operands as split-fp16 pairs (needs dy pre-scaled to O(1): the callers' 2^k scaling);
    False: exact fp32 fma chain.  batch: the reduction of the partial sums is left to batch.flush(); then dw_out (rows of a
    contiguous OIHW tensor whose input-channel extent may be wider than this source: the gradient lands in channels
    [c0, c0 + cin_real)) and db_out (or None) name where the gradient goes."""
    n, h, w, cout = dy.shape
    k = 3 if taps == 9 else 1
    if not _wgrad_instance(cout, x.C, taps):  # (never at the default configuration)
        return _wgrad_tiled(x, prologue, spec, dy, taps, cin_real, want_bias, split, batch, dw_out, c0, db_out)
    p = nv.WgradParams()
    p.N, p.H, p.W, p.Cout, p.taps, p.cin_real = n, h, w, cout, taps, cin_real
    assert x.t.is_contiguous() and dy.is_contiguous() and tuple(x.shape[:3]) == (n, h, w)
    p.src.x = nv.ptr(x.t)
    p.src.C = x.C
    p.src.prologue = prologue
    if prologue != nv.PROLOGUE_NONE:
        p.src.norm = spec.to_native(x)
    if x.valid is not None:
        p.valid_h, p.valid_w = x.valid
    p.dy = nv.ptr(dy)
    p.precision = nv.PRECISION_F16X2 if split else nv.PRECISION_F32
    if nv.PROFILER is not None:  # (algorithmic work of a weight gradient: the forward conv's MACs; x and dy read once)
        def note():
            # (spelled like rocprofv3's kernel trace: bench.py's records and profiles/*pmc*.json share one key)
            # (the split-fp16 gradient runs on the producer / consumer kernel, csrc/dmd_backward.hip)
            geom = f"WgradGeom<{cout // 16}, {x.C // 16}, {taps}>"
            nv.PROFILER.annotate(f"wgrad_ps_kernel<{geom}, {'true' if prologue else 'false'}>" if split and os.environ.get("DIAMOND_WGRAD_PS", "1") != "0" else
                                 f"wgrad_kernel<{geom}, {'true' if split else 'false'}>",
                                 2.0 * taps * cin_real * cout * n * h * w, 4.0 * n * h * w * (x.C + cout))
    else:
        note = lambda: None
    if batch is not None:
        assert dw_out is not None and dw_out.is_contiguous() and dw_out.shape[0] >= cout and tuple(dw_out.shape[2:]) == (k, k) \
            and c0 + cin_real <= dw_out.shape[1] and (db_out is None or (db_out.is_contiguous() and db_out.numel() >= cout))
        p.dw, p.dbias, p.defer_reduce = nv.ptr(dw_out), nv.ptr(db_out), 1
        job = nv.WgradReduceJob()
        nv.check(nv.lib().dmd_wgrad_job(C.byref(p), C.byref(job)), "dmd_wgrad_job")
        # (the partials only, and they stay alive until the flush: the plan's workgroups, not the 1024 the query sizes for)
        ws = torch.empty(job.num_wg * (job.NB * job.NCO * 256 + job.NCO * 16), device=dy.device, dtype=torch.float32)
        p.workspace = job.partials = nv.ptr(ws)
        job.ld_cin, job.c0 = dw_out.shape[1], c0
        note()
        nv.check(nv.lib().dmd_conv2d_wgrad(C.byref(p), nv.stream()), "dmd_conv2d_wgrad")
        batch.add(job, ws, dw_out) if db_out is None else batch.add(job, ws, dw_out, db_out)
        return dw_out, db_out
    ws = torch.empty(int(nv.lib().dmd_wgrad_workspace_floats(C.byref(p))), device=dy.device, dtype=torch.float32)
    dw = torch.empty(cout, cin_real, k, k, device=dy.device, dtype=torch.float32)
    db = torch.empty(cout, device=dy.device, dtype=torch.float32) if want_bias else None
    p.workspace, p.dw, p.dbias = nv.ptr(ws), nv.ptr(dw), nv.ptr(db)
    note()
    nv.check(nv.lib().dmd_conv2d_wgrad(C.byref(p), nv.stream()), "dmd_conv2d_wgrad")
    return dw, db


def _gn_bwd_instance(c: int) -> bool:
    """channel counts dmd_gn_silu_bwd takes (csrc/dmd_backward.hip): 4 ... 256 in powers of two"""
    return c % 4 == 0 and c <= 256 and 256 % (c // 4) == 0 and (c % nv.GN_GROUP == 0 or c < nv.GN_GROUP)


def gn_bwd_sliced(fn, x: Act, spec: NormSpec, da: Tensor, dskip: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """GroupNorm backward over a channel count the kernel has no instance for (96, 160, 512 ...: networks wider than the default
    configuration): a group's backward involves its own 32 channels only, so the channels are cut into 256 / 128 / 64 / 32-wide
    runs of whole groups and `fn` (the kernel call for a supported count) runs on contiguous copies of each."""
    n, h, w, c = x.shape
    g = nv.GN_GROUP
    assert c % g == 0, f"GroupNorm backward over {c} channels"
    dx = torch.empty_like(x.t)
    dma = torch.empty(2, n, c, device=da.device, dtype=torch.float32)
    c0 = 0
    while c0 < c:
        step = next(s for s in (256, 128, 64, 32) if s <= c - c0)
        c1 = c0 + step
        cut = lambda t: None if t is None else t[..., c0:]
        xs = Act(x.t[..., c0:c1].contiguous(), x.stats[:, c0 // g:c1 // g].contiguous(), x.tiles, valid=x.valid)
        spec_s = NormSpec(cut(spec.mul), cut(spec.add), spec.mul_stride, spec.add_stride, spec.plus_one)
        dx_s, dmul_s, dadd_s = fn(xs, spec_s, da[..., c0:c1].contiguous(), None if dskip is None else dskip[..., c0:c1].contiguous())
        dx[..., c0:c1] = dx_s
        dma[0, :, c0:c1] = dmul_s
        dma[1, :, c0:c1] = dadd_s
        c0 = c1
    return dx, dma[0], dma[1]


def _gn_silu_bwd(x: Act, spec: NormSpec, da: Tensor, dskip: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    n, h, w, c = x.shape
    if not _gn_bwd_instance(c):  # (never at the default configuration)
        return gn_bwd_sliced(_gn_silu_bwd, x, spec, da, dskip)
    p = nv.GnBwdParams()
    p.N, p.HW, p.C = n, h * w, c
    if x.valid is not None:  # sums and the count over the valid extent, dx zero outside it
        p.W, p.valid_h, p.valid_w = w, x.valid[0], x.valid[1]
    p.x = nv.ptr(x.t)
    p.norm = spec.to_native(x)
    p.da = nv.fptr(da)
    p.dskip = nv.fptr(dskip)
    dx = torch.empty_like(x.t)
    ws = torch.empty(int(nv.lib().dmd_gn_bwd_workspace_bytes(n, h * w, c)), device=da.device, dtype=torch.uint8)
    dma = torch.empty(2, n, c, device=da.device, dtype=torch.float32)  # [dmul; dadd]: callers that only need the sums over the batch
    dmul, dadd = dma[0], dma[1]                                        # reduce both with one launch (dmul.sum(0) and dadd.sum(0) = dma.sum(1))
    p.dx, p.workspace, p.dmul, p.dadd = nv.ptr(dx), nv.ptr(ws), nv.ptr(dmul), nv.ptr(dadd)
    if nv.PROFILER is not None:  # (HBM-bound: x and da read, dx written, the skip gradient read when there is one)
        nv.PROFILER.annotate("dmd_gn_silu_bwd", 0.0, 4.0 * x.t.numel() * (3 + (dskip is not None)))
    nv.check(nv.lib().dmd_gn_silu_bwd(C.byref(p), nv.stream()), "dmd_gn_silu_bwd")
    return dx, dmul, dadd


# Arithmetic of the encoder's forward and dgrad convolutions: "f16x2" = split-fp32 on the f16 matrix cores where the
# shape is covered (fp32-class accuracy, see dmd_conv_f16ws.hip), "f32" = exact fp32 MFMA.  The weight gradient
# (contraction over pixels) runs on the split-fp16 instance of the wgrad kernel as well (its operands are pre-scaled to
# O(1) by the 2^k scaling of the backward), on the exact fp32 instance with "f32".
AC_PRECISION = os.environ.get("DIAMOND_AC_PRECISION", "f16x2")


def _transposed(w: Tensor) -> Tensor:
    return w.detach().flip(2, 3).transpose(0, 1).contiguous()


def _dgrad_weight(cache: E.PackCache, conv: nn.Conv2d) -> Tensor:
    """Packed weight of the transposed convolution: w_t[ci][co][ky][kx] = w[co][ci][2-ky][2-kx]."""
    return cache.dgrad_weight(conv, 0, conv.in_channels)


def _w16(cache: E.PackCache, conv: nn.Conv2d) -> Optional[Tensor]:
    return cache.conv_weight_f16x2(conv) if AC_PRECISION == "f16x2" else None


def _dgrad_w16(cache: E.PackCache, conv: nn.Conv2d) -> Optional[Tensor]:
    """Split-fp16 pieces of the transposed weight (its output channels = conv.in_channels)."""
    if AC_PRECISION != "f16x2" or conv.kernel_size != (3, 3) or conv.stride != (1, 1):
        return None
    cout_t, cin_t = conv.in_channels, conv.out_channels
    if cout_t not in (32, 64) or cin_t > (128 if cout_t == 64 else 64):
        return None
    return cache.dgrad_weight(conv, 0, conv.in_channels, f16x2=True)


class _Plan:
    """Static description of an ActorCriticEncoder: the layer list + the flat parameter order of
    the autograd.Function."""

    def __init__(self, encoder_seq: nn.Sequential) -> None:
        from .blocks import SmallResBlock

        layers = list(encoder_seq)
        assert isinstance(layers[0], nn.Conv2d), "encoder must start with Conv3x3 (actor_critic.py:105)"
        self.conv_in: nn.Conv2d = layers[0]
        self.blocks: List[Tuple[SmallResBlock, bool]] = []
        i = 1
        while i < len(layers):
            blk = layers[i]
            assert isinstance(blk, SmallResBlock), type(blk)
            pool = i + 1 < len(layers) and isinstance(layers[i + 1], nn.MaxPool2d)
            self.blocks.append((blk, pool))
            i += 2 if pool else 1
        # Image sizes the kernels take as they are: every convolution's level a multiple of the 8-pixel tiles.  A block's
        # convolution runs at the size left by the pools BEFORE it (the last pool's output only gets flattened): 64 for the
        # default encoder (levels 64 / 32 / 16 / 8 -> 4).  Other sizes: the valid extent of a padded buffer (_EncoderFn.forward).
        self.grid_multiple = 8 * 2 ** max(sum(1 for _, pool in self.blocks[:i] if pool) for i in range(len(self.blocks)))
        self.params: List[nn.Parameter] = [self.conv_in.weight, self.conv_in.bias]
        for blk, _ in self.blocks:
            gn, conv = blk.f[0].norm, blk.f[2]
            self.params += [gn.weight, gn.bias, conv.weight, conv.bias]
            if not isinstance(blk.skip_projection, nn.Identity):
                self.params += [blk.skip_projection.weight, blk.skip_projection.bias]


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan: _Plan, cache: E.PackCache, obs: Tensor, *params: Tensor) -> Tensor:
        n, cimg, h, w = obs.shape
        # Image sizes off the kernels' 8-pixel grid at some level (the reference runs any size, MaxPool2d floors:
        # actor_critic.py:45, 72 -> 36 -> 18 -> 9 -> 4) live as the VALID EXTENT of a zero-padded buffer whose levels are all
        # multiples of 8 (include/diamond_hip.h; the denoiser does the same, engine.padded_extent)
        m = plan.grid_multiple
        valid = None if (h % m == 0 and w % m == 0) else (h, w)
        obs_f = obs.detach().float()
        if valid is not None:
            obs_f = torch.nn.functional.pad(obs_f, (0, (w + m - 1) // m * m - w, 0, (h + m - 1) // m * m - h))
        x16 = E.nchw_to_nhwc(obs_f, 16)
        ci = plan.conv_in
        x = E.conv2d([(Act(x16, valid=valid), nv.PROLOGUE_NONE, None)], cache.conv_weight(ci), cache.conv_bias(ci), ci.out_channels,
                     w_f16=_w16(cache, ci))
        saved = []
        for blk, pool in plan.blocks:
            gn, conv = blk.f[0].norm, blk.f[2]
            spec = NormSpec(mul=cache.f32(gn.weight), add=cache.f32(gn.bias))
            sp = blk.skip_projection
            if isinstance(sp, nn.Identity):
                r = x
            else:
                r = E.conv2d([(x, nv.PROLOGUE_NONE, None)], cache.conv_weight(sp), cache.conv_bias(sp), sp.out_channels, taps=1,
                             want_stats=False, w_f16=_w16(cache, sp))
            y = E.conv2d([(x, nv.PROLOGUE_NORM_SILU, spec)], cache.conv_weight(conv), cache.conv_bias(conv), conv.out_channels,
                         residual=r, want_stats=not pool, w_f16=_w16(cache, conv))
            arg = None
            nxt = y
            if pool:
                nxt, arg = _maxpool(y.t, y.valid)
            saved.append((x, arg))
            x = nxt
        ctx.plan, ctx.cache, ctx.x16, ctx.saved = plan, cache, Act(x16, valid=valid), saved
        ctx.cimg = cimg
        ctx.out_valid, ctx.out_buf = x.valid, tuple(x.shape[1:3])
        # flatten in the reference's (c, h, w) order (actor_critic.py:71)
        feat = E.nhwc_to_nchw(x.t)
        if x.valid is not None:
            feat = feat[:, :, :x.valid[0], :x.valid[1]].contiguous()
        return feat.flatten(1)

    @staticmethod
    def backward(ctx, dfeat: Tensor):
        plan, cache = ctx.plan, ctx.cache
        split = AC_PRECISION == "f16x2"  # weight gradients in the split-fp16 form too (exact fp32 with DIAMOND_AC_PRECISION=f32)
        blk_last, _ = plan.blocks[-1]
        cl = blk_last.f[2].out_channels
        hb, wb = ctx.out_buf
        hl, wl = ctx.out_valid if ctx.out_valid is not None else (hb, wb)
        n = dfeat.shape[0]
        # The whole backward is linear in dfeat, so it runs on dfeat * 2^k (k chosen on the device so that the largest
        # entry is O(1)) and the parameter gradients are scaled back by 2^-k: exact in fp32, and it keeps the split-fp16
        # dgrad operands (absolute error floor 2^-25, dmd_conv_f16ws.hip) far above their floor however small the
        # loss scale is -- with loss = mean over B*T the raw gradients are ~1e-6.
        dfeat = dfeat.detach().float()
        amax = dfeat.abs().amax()
        k = torch.where(amax > 0, torch.floor(-torch.log2(amax.clamp_min(1e-37))), torch.zeros_like(amax)).clamp(-120, 120)
        inv_scale = torch.exp2(-k)
        dfeat = (dfeat * torch.exp2(k)).reshape(n, cl, hl, wl)
        if ctx.out_valid is not None:  # zero gradient outside the valid extent of the last buffer
            dfeat = torch.nn.functional.pad(dfeat, (0, wb - wl, 0, hb - hl))
        dcur = E.nchw_to_nhwc(dfeat.contiguous())
        grads_rev: List[Optional[Tensor]] = []
        # (the reductions of this backward's weight gradients as ONE launch at its end instead of two per gradient: same sums)
        batch = WgradBatch() if os.environ.get("DIAMOND_WGRAD_DEFER", "1") == "1" else None

        def wgrad(src, prologue, spec_, dy_, taps, cin):
            if batch is None:
                return _wgrad(src, prologue, spec_, dy_, taps, cin, split=split)
            kk = 3 if taps == 9 else 1
            return _wgrad(src, prologue, spec_, dy_, taps, cin, split=split, batch=batch,
                          dw_out=torch.empty(dy_.shape[-1], cin, kk, kk, device=dy_.device, dtype=torch.float32),
                          db_out=torch.empty(dy_.shape[-1], device=dy_.device, dtype=torch.float32))

        for (blk, pool), (x, arg) in zip(reversed(plan.blocks), reversed(ctx.saved)):
            gn, conv = blk.f[0].norm, blk.f[2]
            spec = NormSpec(mul=cache.f32(gn.weight), add=cache.f32(gn.bias))
            dy = _maxpool_bwd(dcur, arg) if pool else dcur
            dw, db = wgrad(x, nv.PROLOGUE_NORM_SILU, spec, dy, 9, conv.in_channels)
            vy = x.valid  # (a stride-1 block: its output exists where its input does; dy is zero elsewhere, and is treated so)
            da = E.conv2d([(Act(dy, valid=vy), nv.PROLOGUE_NONE, None)], _dgrad_weight(cache, conv), None, conv.in_channels,
                          want_stats=False, w_f16=_dgrad_w16(cache, conv)).t
            sp = blk.skip_projection
            g_skip: List[Optional[Tensor]] = []
            if isinstance(sp, nn.Identity):
                dskip = dy
            else:
                dws, dbs = wgrad(x, nv.PROLOGUE_NONE, None, dy, 1, sp.in_channels)
                dskip = E.conv2d([(Act(dy, valid=vy), nv.PROLOGUE_NONE, None)], _dgrad_weight(cache, sp), None, sp.in_channels, taps=1,
                                 want_stats=False).t
                g_skip = [dws, dbs]
            dx, dmul, dadd = _gn_silu_bwd(x, spec, da, dskip)
            # reversed order of [gn.weight, gn.bias, conv.weight, conv.bias, (skip.weight, skip.bias)]
            dms = dmul._base.sum(1)  # (2, C): both batch sums with one reduction (same per-column order as two .sum(0))
            grads_rev += list(reversed([dms[0], dms[1], dw, db] + g_skip))
            dcur = dx
        ci = plan.conv_in
        dw_in, db_in = wgrad(ctx.x16, nv.PROLOGUE_NONE, None, dcur, 9, ctx.cimg)
        if batch is not None:
            batch.flush()
        # (one multi-tensor launch instead of one broadcast multiplication per gradient: 18 per step, 270 per window)
        grads = torch._foreach_mul([dw_in, db_in] + list(reversed(grads_rev)), inv_scale)
        return (None, None, None, *grads)


class NativeEncoder:
    """Host-side launcher bound to one ActorCritic module (packed-weight cache + plan)."""

    def __init__(self, encoder_seq: nn.Sequential) -> None:
        self.plan = _Plan(encoder_seq)
        self.cache = E.PackCache()

    def __call__(self, obs: Tensor) -> Tensor:
        nv.require_gpu(obs)
        return _EncoderFn.apply(self.plan, self.cache, obs.contiguous(), *self.plan.params)
