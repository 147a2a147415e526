"""LSTMCell + actor / critic heads of the actor-critic on the hand-written kernels, forward AND backward
(reference models/actor_critic.py:46-48,72-73: `nn.LSTMCell(1024, 512)`, `actor_linear`, `critic_linear`, under
`loss.backward()`, trainer.py:366).

One `torch.autograd.Function` per policy step: gate GEMMs and head GEMM on `dmd_linear` (v_mfma_f32_16x16x4_f32, exact
fp32 fma chains), gate nonlinearity in `dmd_lstm_pointwise`; the backward is `dmd_lstm_pointwise_bwd` plus the
transposed GEMMs, again on `dmd_linear` -- its operands are K-contiguous ("NT"), so the data/weight gradients use
transposed copies of the (small) activation matrices and version-cached transposed weights.  BPTT over the 15-step
window is torch's ordinary chaining of these Functions through (hx, cx).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import engine as E
from . import native as nv


def _pad_k(t: Tensor) -> Tensor:
    """(M, K) -> contiguous (M, K rounded up to 16) (dmd_linear contracts over multiples of 16)."""
    k = t.shape[1]
    kp = (k + 15) // 16 * 16
    t = t.contiguous()
    return t if kp == k else F.pad(t, (0, kp - k))


def _mm_nt(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """a (M, K) @ w (N, K)^T on dmd_linear, K zero-padded to a multiple of 16."""
    a, w = _pad_k(a), _pad_k(w)
    return E.linear(a, w, bias, out=out, accumulate=accumulate)


class LstmHeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cache: E.PackCache, x: Tensor, hx: Tensor, cx: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor,
                b_hh: Tensor, w_heads: Tensor, b_heads: Tensor):
        x, hx, cx = x.detach().float().contiguous(), hx.detach().float().contiguous(), cx.detach().float().contiguous()
        n, hd = hx.shape
        gates = E.linear(x, w_ih.detach(), b_ih.detach())
        E.linear(hx, w_hh.detach(), b_hh.detach(), out=gates, accumulate=True)
        h = torch.empty_like(hx)
        c = torch.empty_like(cx)
        nv.check(nv.lib().dmd_lstm_pointwise(nv.fptr(gates), nv.fptr(cx), nv.fptr(h), nv.fptr(c), n, hd, nv.stream()),
                 "dmd_lstm_pointwise")
        heads = E.linear(h, w_heads.detach().contiguous(), b_heads.detach().contiguous())
        ctx.save_for_backward(x, hx, cx, gates, h, c, w_heads)
        # transposed weights for the data gradients: cached per parameter version (forward sees the Parameter objects)
        ctx.w_ih_t = cache.get(w_ih, "T", lambda w: w.detach().t().contiguous()) if ctx.needs_input_grad[1] else None
        ctx.w_hh_t = cache.get(w_hh, "T", lambda w: w.detach().t().contiguous()) if ctx.needs_input_grad[2] else None
        return heads, h, c

    @staticmethod
    def backward(ctx, dheads: Optional[Tensor], dh: Optional[Tensor], dc: Optional[Tensor]):
        x, hx, cx, gates, h, c, w_heads = ctx.saved_tensors
        n, hd = hx.shape
        need = ctx.needs_input_grad  # (cache, x, hx, cx, w_ih, w_hh, b_ih, b_hh, w_heads, b_heads)
        dw_heads = db_heads = None
        dh_total = None if dh is None else dh.detach().float().contiguous()
        if dheads is not None:
            dheads = dheads.detach().float().contiguous()
            # dh += dheads @ W_heads ; dW_heads = dheads^T @ h ; db_heads = sum dheads
            wt = w_heads.detach().t().contiguous()  # (hd, A + 1)
            dh_total = _mm_nt(dheads, wt, out=dh_total.clone() if dh_total is not None else None, accumulate=dh_total is not None)
            dw_heads = _mm_nt(dheads.t(), h.t())
            db_heads = dheads.sum(0)
        dgates = torch.empty_like(gates)
        dc_prev = torch.empty_like(cx)
        dcc = None if dc is None else dc.detach().float().contiguous()
        nv.check(nv.lib().dmd_lstm_pointwise_bwd(nv.fptr(gates), nv.fptr(cx), nv.fptr(c), nv.fptr(dh_total), nv.fptr(dcc),
                                                 nv.fptr(dgates), nv.fptr(dc_prev), n, hd, nv.stream()), "dmd_lstm_pointwise_bwd")
        dx = dhx = None
        if need[1]:
            dx = _mm_nt(dgates, ctx.w_ih_t)
        if need[2]:
            dhx = _mm_nt(dgates, ctx.w_hh_t)
        dgt = dgates.t().contiguous()
        dw_ih = _mm_nt(dgt, x.t())
        dw_hh = _mm_nt(dgt, hx.t())
        db = dgates.sum(0)
        return (None, dx, dhx, dc_prev if need[3] else None, dw_ih, dw_hh, db, db, dw_heads, db_heads)


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) on dmd_linear, forward and backward (the LSTM input projection over a whole segment and the
    two head linears of RewEndModel's training step, reference rew_end_model.py:35-40,53-54)."""

    @staticmethod
    def forward(ctx, cache: E.PackCache, x: Tensor, w: Tensor, b: Optional[Tensor]):
        x = x.detach().float().contiguous()
        y = _mm_nt(x, w.detach(), None if b is None else b.detach().contiguous())
        ctx.save_for_backward(x)
        # a Parameter is transposed once per version; a derived weight (the permuted weight_ih) on every call
        ctx.w_t = (cache.get(w, "T", lambda t: t.detach().t().contiguous()) if isinstance(w, torch.nn.Parameter)
                   else w.detach().t().contiguous()) if ctx.needs_input_grad[1] else None
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        (x,) = ctx.saved_tensors
        dy = dy.detach().float().contiguous()
        dx = _mm_nt(dy, ctx.w_t) if ctx.needs_input_grad[1] else None
        dw = _mm_nt(dy.t(), x.t())
        return None, dx, dw, dy.sum(0) if ctx.has_bias else None


class LstmStepFn(torch.autograd.Function):
    """One step of nn.LSTM given the precomputed input projection gx = x W_ih^T + b_ih:
    (h', c') = cell(gx + hx W_hh^T + b_hh, cx).  BPTT over the segment is torch chaining these through (hx, cx)."""

    @staticmethod
    def forward(ctx, cache: E.PackCache, gx: Tensor, hx: Tensor, cx: Tensor, w_hh: Tensor, b_hh: Tensor):
        hx, cx = hx.detach().float().contiguous(), cx.detach().float().contiguous()
        n, hd = hx.shape
        gates = gx.detach().float().clone(memory_format=torch.contiguous_format)
        E.linear(hx, w_hh.detach(), b_hh.detach(), out=gates, accumulate=True)
        h, c = torch.empty_like(hx), torch.empty_like(cx)
        nv.check(nv.lib().dmd_lstm_pointwise(nv.fptr(gates), nv.fptr(cx), nv.fptr(h), nv.fptr(c), n, hd, nv.stream()),
                 "dmd_lstm_pointwise")
        ctx.save_for_backward(hx, cx, gates, c)
        ctx.w_hh_t = cache.get(w_hh, "T", lambda w: w.detach().t().contiguous())
        return h, c

    @staticmethod
    def backward(ctx, dh: Optional[Tensor], dc: Optional[Tensor]):
        hx, cx, gates, c = ctx.saved_tensors
        n, hd = hx.shape
        dhc = None if dh is None else dh.detach().float().contiguous()
        dcc = None if dc is None else dc.detach().float().contiguous()
        dgates, dc_prev = torch.empty_like(gates), torch.empty_like(cx)
        nv.check(nv.lib().dmd_lstm_pointwise_bwd(nv.fptr(gates), nv.fptr(cx), nv.fptr(c), nv.fptr(dhc), nv.fptr(dcc),
                                                 nv.fptr(dgates), nv.fptr(dc_prev), n, hd, nv.stream()), "dmd_lstm_pointwise_bwd")
        need = ctx.needs_input_grad  # (cache, gx, hx, cx, w_hh, b_hh)
        dhx = _mm_nt(dgates, ctx.w_hh_t) if need[2] else None
        dw_hh = _mm_nt(dgates.t(), hx.t())
        return None, dgates, dhx, dc_prev if need[3] else None, dw_hh, dgates.sum(0)


class LstmBurnInFn(torch.autograd.Function):
    """The burn-in of the policy LSTM on the context frames of a NEW episode (reference env_loop.py:53-56: `tb` calls of
    predict_act_value from a zeroed state, WITH grad; only (hx, cx) survive, the heads are not needed) as ONE autograd node:
    x = the frames' encoder features, frame-major (tb * k, F).  Forward = the per-step arithmetic of LstmHeadsFn -- the input
    projection of all tb * k rows in one GEMM (a row's summation order in dmd_linear does not depend on the batch), then per
    frame the recurrent GEMM accumulated onto its rows and the gate kernel: bitwise the tb separate calls.  Backward = BPTT
    inside the node with ONE set of weight gradients (three GEMMs over all tb * k rows) instead of tb sets that autograd
    then adds up with one launch per parameter and call."""

    @staticmethod
    def forward(ctx, cache: E.PackCache, x: Tensor, tb: int, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor):
        x = x.detach().float().contiguous()
        k, hd = x.shape[0] // tb, w_hh.shape[1]
        gates = E.linear(x, w_ih.detach(), b_ih.detach())  # (tb * k, 4 hd)
        hs = torch.zeros(tb + 1, k, hd, device=x.device, dtype=torch.float32)  # hs[i] = state before frame i (hs[0] = 0: the gated row)
        cs = torch.zeros(tb + 1, k, hd, device=x.device, dtype=torch.float32)
        whh, bhh = w_hh.detach(), b_hh.detach()
        for i in range(tb):
            g = gates[i * k:(i + 1) * k]
            E.linear(hs[i], whh, bhh, out=g, accumulate=True)
            nv.check(nv.lib().dmd_lstm_pointwise(nv.fptr(g), nv.fptr(cs[i]), nv.fptr(hs[i + 1]), nv.fptr(cs[i + 1]), k, hd, nv.stream()),
                     "dmd_lstm_pointwise")
        ctx.save_for_backward(x, gates, hs, cs)
        ctx.tb = tb
        ctx.w_ih_t = cache.get(w_ih, "T", lambda w: w.detach().t().contiguous()) if ctx.needs_input_grad[1] else None
        ctx.w_hh_t = cache.get(w_hh, "T", lambda w: w.detach().t().contiguous())
        return hs[tb].clone(), cs[tb].clone()

    @staticmethod
    def backward(ctx, dh: Optional[Tensor], dc: Optional[Tensor]):
        x, gates, hs, cs = ctx.saved_tensors
        tb = ctx.tb
        k, hd = hs.shape[1], hs.shape[2]
        dgates = torch.empty_like(gates)
        dhc = torch.zeros(k, hd, device=x.device) if dh is None else dh.detach().float().contiguous()
        dcc = torch.zeros(k, hd, device=x.device) if dc is None else dc.detach().float().contiguous()
        for i in reversed(range(tb)):
            dg, dc_prev = dgates[i * k:(i + 1) * k], torch.empty_like(dcc)
            nv.check(nv.lib().dmd_lstm_pointwise_bwd(nv.fptr(gates[i * k:(i + 1) * k]), nv.fptr(cs[i]), nv.fptr(cs[i + 1]), nv.fptr(dhc), nv.fptr(dcc),
                                                     nv.fptr(dg), nv.fptr(dc_prev), k, hd, nv.stream()), "dmd_lstm_pointwise_bwd")
            dcc = dc_prev
            if i > 0:
                dhc = _mm_nt(dg, ctx.w_hh_t)
        dx = _mm_nt(dgates, ctx.w_ih_t) if ctx.needs_input_grad[1] else None
        dgt = dgates.t().contiguous()
        dw_ih = _mm_nt(dgt, x.t())
        # (frame 0 starts from the zero state: its rows contribute nothing to dW_hh)
        dw_hh = _mm_nt(dgt[:, k:].contiguous(), hs[1:tb].reshape((tb - 1) * k, hd).t()) if tb > 1 else torch.zeros_like(ctx.w_hh_t.t())
        db = dgates.sum(0)
        return None, dx, None, dw_ih, dw_hh, db, db


def lstm_burn_in(cache: E.PackCache, x: Tensor, tb: int, lstm) -> Tuple[Tensor, Tensor]:
    """(hx, cx) after stepping `lstm` from the zero state over tb frames' features x (tb * k, F), frame-major."""
    nv.require_gpu(x)
    return LstmBurnInFn.apply(cache, x, tb, lstm.weight_ih, lstm.weight_hh, lstm.bias_ih, lstm.bias_hh)


def lstm_heads(cache: E.PackCache, x: Tensor, hx: Tensor, cx: Tensor, lstm, actor_linear, critic_linear
               ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """(logits_act, val, hx', cx') of reference actor_critic.py:72-73."""
    nv.require_gpu(x)
    w_heads = torch.cat((actor_linear.weight, critic_linear.weight), dim=0)
    b_heads = torch.cat((actor_linear.bias, critic_linear.bias), dim=0)
    heads, h, c = LstmHeadsFn.apply(cache, x, hx, cx, lstm.weight_ih, lstm.weight_hh, lstm.bias_ih, lstm.bias_hh, w_heads, b_heads)
    a = actor_linear.out_features
    return heads[:, :a], heads[:, a], h, c


class MergeSlotsFn(torch.autograd.Function):
    """out[r] = base[r] for a live row, values[slot of r] for a row that was reset (dmd_merge_slots; env_loop: the burnt-in LSTM state
    of the new episodes merged into the batch's state, reference env_loop.py:51-56).  Rows of any trailing shape."""

    @staticmethod
    def forward(ctx, base: Tensor, values: Tensor, row_slot: Tensor, slot_row: Tensor):
        b, k = base.shape[0], values.shape[0]
        base2 = base.detach().float().contiguous().reshape(b, -1)
        val2 = values.detach().float().contiguous().reshape(k, -1)
        out = torch.empty_like(base2)
        nv.check(nv.lib().dmd_merge_slots(nv.fptr(base2), nv.fptr(val2), nv.ptr(row_slot), nv.fptr(out), b, base2.shape[1], nv.stream()),
                 "dmd_merge_slots")
        ctx.save_for_backward(row_slot, slot_row)
        ctx.shapes = (tuple(base.shape), tuple(values.shape))
        return out.reshape(base.shape)

    @staticmethod
    def backward(ctx, d_out: Tensor):
        row_slot, slot_row = ctx.saved_tensors
        (bs, vs) = ctx.shapes
        b, k = bs[0], vs[0]
        g = d_out.detach().float().contiguous().reshape(b, -1)
        d_base = torch.empty_like(g) if ctx.needs_input_grad[0] else None
        d_val = torch.empty(k, g.shape[1], device=g.device, dtype=torch.float32)
        nv.check(nv.lib().dmd_merge_slots_bwd(nv.fptr(g), nv.ptr(row_slot), nv.ptr(slot_row), nv.fptr(d_base), nv.fptr(d_val), b, k, g.shape[1],
                                              nv.stream()), "dmd_merge_slots_bwd")
        return (None if d_base is None else d_base.reshape(bs)), d_val.reshape(vs), None, None


def merge_slots(base: Tensor, values: Tensor, row_slot: Tensor, slot_row: Tensor) -> Tensor:
    nv.require_gpu(base)
    return MergeSlotsFn.apply(base, values, row_slot, slot_row)
