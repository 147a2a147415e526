"""Training backward of the denoiser's U-Net on the hand-written HIP kernels (the
`loss.backward()` of `Denoiser.forward`, reference denoiser.py:93-122 / trainer.py:363-366, where ATen autograd
differentiates blocks.py / inner_model.py).

Design: the inference forward (`InnerModel.run`) is executed unchanged while `engine.TAPE` records every conv /
attention launch; ONE `torch.autograd.Function` spans the whole network and its backward walks the tape in reverse.
Per recorded convolution, with dOut the gradient of its output:
  * bias / weight gradient  -> `dmd_conv2d_wgrad` per source (the activated input SiLU(GN(x)*..) is recomputed from x and
    its statistics while staging: nothing but the layer inputs was saved);
  * data gradient           -> `dmd_conv2d` on the flipped / transposed weight slice of that source (split-fp16 MFMA
    where the shape is covered); a stride-2 convolution runs both on the zero-stuffed dOut, an upsampling one on the
    materialised nearest-x2 input followed by a 2x2 sum;
  * fused prologue          -> `dmd_gn_silu_bwd` (GroupNorm + FiLM/affine + SiLU or identity): dx plus per-(sample,
    channel) gradients of the multiplicative / additive terms, which are columns of the batched FiLM table
    (AdaGroupNorm) or the affine parameters of an nn.GroupNorm;
  * residual                -> accumulated into the gradient of the residual tensor (through the GroupNorm affine for
    the attention block's `x_normed + out_proj(y)`, blocks.py:72).
Attention core: `dmd_attention_bwd`.  The FiLM table itself (`cond @ W_cat^T + b_cat`), the 256-wide cond MLP and the
action embedding are a handful of tiny GEMMs: they run as torch ops under ordinary autograd, so the table gradient
returned here flows on into the 44 AdaGroupNorm linears, `cond_proj` and `act_emb`.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import engine as E
from . import native as nv
from .ac_native import WgradBatch, _gn_bwd_instance, _gn_silu_bwd, _transposed, _wgrad, gn_bwd_sliced
from .engine import Act, AttnRecord, ConvRecord, NormSpec

TRAIN_PRECISION = "f16x2"  # arithmetic of the forward, dgrad and wgrad convolutions (split-fp16 operands, fp32 accumulate); "f32" = exact


def _key(t: Tensor) -> int:
    return t.data_ptr()


class _Grads:
    """Gradient accumulator keyed by activation storage."""

    def __init__(self) -> None:
        self.g: Dict[int, Tensor] = {}

    def add(self, t: Tensor, g: Tensor) -> None:
        k = _key(t)
        if k in self.g:
            self.g[k] = self.g[k] + g
        else:
            self.g[k] = g

    def pop(self, t: Tensor) -> Optional[Tensor]:
        return self.g.pop(_key(t), None)

    def peek(self, t: Tensor) -> Optional[Tensor]:
        return self.g.get(_key(t))


def _dgrad_weights(cache: E.PackCache, conv: nn.Conv2d, c0: int, c1: int, cout_pad: Optional[int], use_f16: bool):
    """Packed weight of the transposed convolution restricted to input channels [c0, c1) of `conv`:
    w_t[ci - c0][co][ky][kx] = w[co][ci][K-1-ky][K-1-kx]; co zero-padded to `cout_pad` (conv_out: 3 -> 16)."""

    wp = cache.dgrad_weight(conv, c0, c1, cout_pad or 0)
    w16 = None
    ci, co = c1 - c0, cout_pad or conv.out_channels
    k = conv.kernel_size[0]
    if use_f16 and ci in (32, 64) and co <= (128 if ci == 64 else 64) and k in (1, 3):
        w16 = cache.dgrad_weight(conv, c0, c1, cout_pad or 0, f16x2=True)
    return wp, w16


def _zero_stuff(dy: Tensor) -> Tensor:
    """(N, H, W, C) -> (N, 2H, 2W, C) with dy at the even positions: the transposed stride-2 convolution and its weight
    gradient are the stride-1 ones of this tensor."""
    n, h, w, c = dy.shape
    z = torch.zeros(n, 2 * h, 2 * w, c, device=dy.device, dtype=dy.dtype)
    z[:, ::2, ::2] = dy
    return z


def _upsample_nearest(x: Tensor) -> Tensor:
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous()


def _sum_pool2(g: Tensor) -> Tensor:
    n, h, w, c = g.shape
    return g.reshape(n, h // 2, 2, w // 2, 2, c).sum(dim=(2, 4)).contiguous()


def _gn_bwd(x: Act, spec: NormSpec, da: Tensor, dskip: Optional[Tensor], identity: bool):
    import ctypes as C

    if not identity:
        return _gn_silu_bwd(x, spec, da, dskip)
    n, h, w, c = x.shape
    if not _gn_bwd_instance(c):  # (never at the default configuration)
        return gn_bwd_sliced(lambda *a: _gn_bwd(*a, True), x, spec, da, dskip)
    p = nv.GnBwdParams()
    p.N, p.HW, p.C = n, h * w, c
    p.identity_activation = 1
    p.x = nv.ptr(x.t)
    p.norm = spec.to_native(x)
    p.da = nv.fptr(da)
    p.dskip = nv.fptr(dskip)
    dx = torch.empty_like(x.t)
    ws = torch.empty(int(nv.lib().dmd_gn_bwd_workspace_bytes(n, h * w, c)), device=da.device, dtype=torch.uint8)
    dmul = torch.empty(n, c, device=da.device, dtype=torch.float32)
    dadd = torch.empty(n, c, device=da.device, dtype=torch.float32)
    p.dx, p.workspace, p.dmul, p.dadd = nv.ptr(dx), nv.ptr(ws), nv.ptr(dmul), nv.ptr(dadd)
    nv.check(nv.lib().dmd_gn_silu_bwd(C.byref(p), nv.stream()), "dmd_gn_silu_bwd")
    return dx, dmul, dadd


class _ParamGrads:
    def __init__(self, table: Tensor) -> None:
        self._table = table
        self._pieces: Dict[int, Tensor] = {}  # first column -> (N, c) gradient block of the FiLM table
        self._dtable: Optional[Tensor] = None
        self.by_param: Dict[int, Tensor] = {}

    def add_param(self, p: nn.Parameter, g: Tensor) -> None:
        k = id(p)
        self.by_param[k] = self.by_param[k] + g if k in self.by_param else g

    def norm_terms(self, spec: NormSpec, dmul: Tensor, dadd: Tensor) -> None:
        c = dmul.shape[1]
        if spec.film_cols is not None:  # AdaGroupNorm: y = xn * (1 + scale) + shift -> d scale = dmul, d shift = dadd
            mc, ac = spec.film_cols
            # every column block of the table belongs to ONE normalised source: the blocks are collected and the table
            # is assembled once (dtable) -- not 2 strided `+=` launches per normalisation
            for col, g in ((mc, dmul), (ac, dadd)):
                self._pieces[col] = self._pieces[col] + g if col in self._pieces else g
        elif spec.gn_module is not None:
            self.add_param(spec.gn_module.weight, dmul.sum(0))
            self.add_param(spec.gn_module.bias, dadd.sum(0))

    @property
    def dtable(self) -> Tensor:
        """Gradient of the batched FiLM table (N, total): the collected column blocks, zeros where no block was produced."""
        if self._dtable is None:
            n, total = self._table.shape
            parts, at = [], 0
            for col in sorted(self._pieces):
                g = self._pieces[col]
                assert col >= at, "overlapping FiLM column blocks"
                if col > at:
                    parts.append(torch.zeros(n, col - at, device=g.device, dtype=g.dtype))
                parts.append(g)
                at = col + g.shape[1]
            if at < total or not parts:
                parts.append(torch.zeros(n, total - at, device=self._table.device, dtype=self._table.dtype))
            self._dtable = torch.cat(parts, dim=1)
        return self._dtable


def backward_tape(tape: List, cache: E.PackCache, d_out: Tensor, table: Tensor, use_f16: bool,
                  out_nhwc: Optional[Tensor] = None) -> _ParamGrads:
    """Walk the recorded launches in reverse; returns the parameter / FiLM-table gradients.
    `out_nhwc` None: the tape ends with the NCHW head convolution (U-Net) and `d_out` is (N, 3, H, W); otherwise
    `out_nhwc` is the recorded NHWC activation the network returned (an encoder) and `d_out` its gradient."""
    grads = _Grads()
    pg = _ParamGrads(table)
    pending_norm: Dict[int, Tensor] = {}  # gradient w.r.t. GN_affine(x) handed over by a residual_norm consumer
    # the reductions of ALL weight gradients as one launch per 32 at the end (bit-identical sums), every source / 64-channel piece
    # of a convolution writing its slice of ONE OIHW tensor; DIAMOND_WGRAD_DEFER=0: reduced per call and concatenated, as before
    batch = WgradBatch() if os.environ.get("DIAMOND_WGRAD_DEFER", "1") == "1" else None
    if out_nhwc is None:
        last = tape[-1]
        assert isinstance(last, ConvRecord) and last.out_nchw, "the tape must end with the NCHW head convolution"
        # dF (N, 3, H, W) -> NHWC with the channels zero-padded to 16 (wgrad / dgrad work on 16-channel groups)
        grads.add(last.out.t, E.nchw_to_nhwc(d_out.contiguous().float(), 16))
    else:
        grads.add(out_nhwc, d_out.contiguous().float())
    for rec in reversed(tape):
        if isinstance(rec, AttnRecord):
            dy = grads.pop(rec.out)
            if dy is None:
                continue
            n, h, w, c3 = rec.qkv.shape
            dqkv = torch.empty_like(rec.qkv.t)
            ws = torch.empty(int(nv.lib().dmd_attention_bwd_workspace_floats(n, h * w, rec.c)), device=dy.device,
                             dtype=torch.float32)
            dyc = dy.contiguous()
            nv.check(nv.lib().dmd_attention_bwd(nv.fptr(rec.qkv.t), nv.fptr(rec.out), nv.fptr(dyc), nv.fptr(dqkv),
                                                nv.fptr(ws), n, h * w, rec.c, rec.head_dim, nv.stream()), "dmd_attention_bwd")
            grads.add(rec.qkv.t, dqkv)
            continue
        dout = grads.pop(rec.out.t)
        if dout is None:
            continue
        conv = rec.module
        cout = conv.out_channels
        head = rec.out_nchw  # conv_out: dout carries 16 channels (3 real)
        cpad = 16 if head else None
        # ---- residual branch
        if rec.residual is not None:
            if rec.residual_norm is not None:  # x_normed + out_proj(y): goes through GN_affine(x) with the qkv branch
                pending_norm[_key(rec.residual.t)] = dout
            else:
                grads.add(rec.residual.t, dout)
        # ---- geometry of the gradient the weight / data kernels consume
        dy_k = _zero_stuff(dout) if rec.stride == 2 else dout
        k = conv.kernel_size[0]
        dws: List[Tensor] = []
        db: Optional[Tensor] = None
        c0 = 0
        if batch is not None:  # (rows: the channel count dy carries -- the head's 3 output channels travel padded to 16)
            dw_all = torch.empty(dy_k.shape[-1], conv.in_channels, k, k, device=dy_k.device, dtype=torch.float32)
            db = torch.empty(dy_k.shape[-1], device=dy_k.device, dtype=torch.float32)
        for si, (a, prologue, spec) in enumerate(rec.srcs):
            ci_pad = a.C
            ci_real = min(ci_pad, conv.in_channels - c0)
            src = Act(_upsample_nearest(a.t)) if rec.upsample else a
            assert not (rec.upsample and prologue != nv.PROLOGUE_NONE)
            if batch is not None:
                step = 64 if cout > 64 and not head else dy_k.shape[-1]  # (the wgrad instances take at most 64 output channels: qkv)
                for o in range(0, dy_k.shape[-1], step):
                    dy_o = dy_k if step == dy_k.shape[-1] else dy_k[..., o:o + step].contiguous()
                    _wgrad(src, prologue, spec, dy_o, rec.taps, ci_real, split=use_f16, batch=batch, dw_out=dw_all[o:o + step], c0=c0,
                           db_out=db[o:o + step] if si == 0 else None)
            elif cout > 64 and not head:  # qkv (192 channels): the wgrad instances take at most 64 output channels
                parts = []
                for o in range(0, cout, 64):
                    dwp, dbp = _wgrad(src, prologue, spec, dy_k[..., o:o + 64].contiguous(), rec.taps, ci_real, split=use_f16)
                    parts.append((dwp, dbp))
                dw_i = torch.cat([p_[0] for p_ in parts], dim=0)
                db_i = torch.cat([p_[1] for p_ in parts], dim=0)
            else:
                dw_i, db_i = _wgrad(src, prologue, spec, dy_k, rec.taps, ci_real, split=use_f16)
            if batch is None:
                dws.append(dw_i[:cout])
                if si == 0:
                    db = db_i[:cout]
            if a.needs_grad:
                wp, w16 = _dgrad_weights(cache, conv, c0, c0 + ci_real, cpad, use_f16)
                da = E.conv2d([(Act(dy_k), nv.PROLOGUE_NONE, None)], wp, None, ci_real, taps=rec.taps, want_stats=False, w_f16=w16,
                              fast_math=use_f16).t
                if rec.upsample:
                    da = _sum_pool2(da)
                if prologue == nv.PROLOGUE_NONE:
                    grads.add(a.t, da)
                else:
                    identity = prologue == nv.PROLOGUE_NORM
                    if identity and _key(a.t) in pending_norm:
                        da = da + pending_norm.pop(_key(a.t))
                    dx, dmul, dadd = _gn_bwd(a, spec, da, grads.pop(a.t), identity)
                    grads.g[_key(a.t)] = dx  # dx already contains the gradient accumulated so far (dskip)
                    pg.norm_terms(spec, dmul, dadd)
            c0 += ci_real
        if batch is not None:
            if id(conv.weight) in pg.by_param:  # a convolution recorded twice: its gradients are ADDED, so they have to exist
                batch.flush()
            pg.add_param(conv.weight, dw_all[:cout])
            db = db[:cout]
        else:
            pg.add_param(conv.weight, torch.cat(dws, dim=1) if len(dws) > 1 else dws[0])
        if conv.bias is not None and db is not None:
            pg.add_param(conv.bias, db)
    if batch is not None:
        batch.flush()
    assert not pending_norm, "a normalised residual was never matched with its pre-norm consumer"
    return pg


class UNetTrainFn(torch.autograd.Function):
    """F = InnerModel.run(packed_in, table) with a hand-written backward.  Inputs that carry gradients: the FiLM
    table and every parameter in `params` (the convolutions, nn.GroupNorm affines and attention projections of
    conv_in / unet / norm_out / conv_out)."""

    @staticmethod
    def forward(ctx, inner, packed_in: Tensor, table: Tensor, precision: str, *params: Tensor) -> Tensor:
        assert E.TAPE is None, "nested recording"
        E.TAPE = []
        try:
            out = inner.run(packed_in, None, precision=precision, table=table.detach())
            tape = E.TAPE
        finally:
            E.TAPE = None
        ctx.inner, ctx.tape, ctx.table, ctx.precision = inner, tape, table.detach(), precision
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, d_out: Tensor):
        pg, inv = scaled_backward(ctx.tape, ctx.inner._cache, d_out, ctx.table, use_f16=ctx.precision == "f16x2")
        ctx.tape = None  # free the saved activations
        grads = _unscale(pg, ctx.params, inv)
        return (None, None, pg.dtable * inv, None, *grads)


def _unscale(pg: "_ParamGrads", params, inv: Tensor):
    """Gradients of the scaled problem times 2^-k.  One multiply per parameter would be 236 launches of 4 us for the
    denoiser (torch._foreach_mul with a device scalar falls back to exactly that): the gradients are gathered into one flat
    buffer (torch.cat: two launches), scaled there, and handed out as views of it."""
    have = [pg.by_param[id(p)] for p in params if pg.by_param.get(id(p)) is not None]
    if not have:
        return tuple(None for _ in params)
    flat = torch.cat([g.reshape(-1) for g in have])
    flat *= inv
    out, off = [], 0
    for g in have:
        out.append(flat[off:off + g.numel()].view(g.shape))
        off += g.numel()
    scaled = iter(out)
    return tuple(None if pg.by_param.get(id(p)) is None else next(scaled) for p in params)


def scaled_backward(tape: List, cache: E.PackCache, d_out: Tensor, table: Tensor, use_f16: bool,
                    out_nhwc: Optional[Tensor] = None) -> Tuple[_ParamGrads, Tensor]:
    """backward_tape on d_out * 2^k with the largest entry O(1); returns (gradients of the scaled problem, 2^-k).
    The backward is linear in d_out, so scaling back is exact in fp32.  Loss gradients are ~1e-5 and smaller; the
    split-fp16 dgrad operands have an absolute resolution floor of 2^-25 (dmd_conv_f16ws.hip) that unscaled gradients
    would sit on."""
    d_out = d_out.detach().float()
    amax = d_out.abs().amax()
    k = torch.where(amax > 0, torch.floor(-torch.log2(amax.clamp_min(1e-37))), torch.zeros_like(amax)).clamp(-120, 120)
    return backward_tape(tape, cache, d_out * torch.exp2(k), table, use_f16, out_nhwc), torch.exp2(-k)


class EncoderTrainFn(torch.autograd.Function):
    """feat = encoder.run(x, table) (an AdaGN-ResBlock encoder returning an NHWC activation: RewEndEncoder) with the
    same recorded-tape backward as the U-Net.  `run(table) -> Tensor` executes the forward."""

    @staticmethod
    def forward(ctx, run, cache: E.PackCache, table: Tensor, precision: str, *params: Tensor) -> Tensor:
        assert E.TAPE is None, "nested recording"
        E.TAPE = []
        try:
            out = run(table.detach())
            tape = E.TAPE
        finally:
            E.TAPE = None
        ctx.cache, ctx.tape, ctx.table, ctx.precision, ctx.out, ctx.params = cache, tape, table.detach(), precision, out, params
        return out

    @staticmethod
    def backward(ctx, d_out: Tensor):
        pg, inv = scaled_backward(ctx.tape, ctx.cache, d_out, ctx.table, ctx.precision == "f16x2", out_nhwc=ctx.out)
        ctx.tape = ctx.out = None
        grads = _unscale(pg, ctx.params, inv)
        return (None, None, pg.dtable * inv, None, *grads)


def trainable_unet_params(inner) -> List[nn.Parameter]:
    """Parameters whose gradients come out of UNetTrainFn (everything the recorded launches touch directly; the
    AdaGroupNorm linears, cond_proj and act_emb get theirs through the FiLM table under torch autograd)."""
    from .blocks import AdaGroupNorm

    skip = set()
    for m in inner.modules():
        if isinstance(m, AdaGroupNorm):
            skip.update(id(p) for p in m.parameters())
    skip.update(id(p) for p in inner.cond_proj.parameters())
    skip.update(id(p) for p in inner.act_emb.parameters())
    return [p for p in inner.parameters() if id(p) not in skip]
