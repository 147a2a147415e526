"""Parameter containers + native execution of the DIAMOND building blocks.

Each class keeps the reference's attribute names and parameter shapes (models/blocks.py)
so that `state_dict()` keys, `Agent.load` (agent.py:48-62) and `configure_opt`
(utils.py:129-166) keep working, but none of them computes anything in PyTorch: `run()`
issues hand-written HIP kernels through `diamond_amd.engine` on NHWC activations.  Calling
`forward()` (the reference's eager NCHW entry point) converts layouts at the boundary and
calls `run()`; on a CPU tensor it raises -- there is no fallback path.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import engine as E
from . import native as nv
from .engine import Act, NormSpec

GN_GROUP_SIZE = 32
GN_EPS = 1e-5
ATTN_HEAD_DIM = 8
# DIAMOND_LOWRES_CHAIN: bit 0 = the 8x8 level of the denoiser's U-Net as one dmd_lowres_chain launch, bit 1 = the 8x8 tail of
# the reward / end encoder as one dmd_lowres_chain32 launch (default 3 = both; 0 = launch by launch).
LOWRES_CHAIN = int(os.environ.get("DIAMOND_LOWRES_CHAIN", "3"))


def conv3x3(cin: int, cout: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1)


def conv1x1(cin: int, cout: int) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=1, stride=1, padding=0)


def _groups(c: int) -> int:
    # The reference forms max(1, C // 32) groups of C / groups channels (blocks.py:27,38): groups of exactly 32 whenever C is a
    # multiple of 32 -- the only case the kernels implement (DMD_GN_GROUP).  Say so where the network is built, not at its first launch.
    assert c % GN_GROUP_SIZE == 0, (f"normalisation over {c} channels: diamond_amd's kernels normalise in groups of {GN_GROUP_SIZE} channels, "
                                    f"so every normalised width has to be a multiple of {GN_GROUP_SIZE} (the reference would form "
                                    f"{max(1, c // GN_GROUP_SIZE)} group(s) of {c / max(1, c // GN_GROUP_SIZE):g} here)")
    return c // GN_GROUP_SIZE


class RunCtx:
    """Per-forward state shared by the blocks: packed-weight cache and the batched FiLM
    table ((B, sum 2C) = every AdaGroupNorm.linear of the network applied to `cond` in ONE
    GEMM -- they all consume the same cond vector, blocks.py:171-177)."""

    def __init__(self, cache: E.PackCache, film: Optional["FilmTable"] = None, table: Optional[Tensor] = None,
                 naive: Optional[bool] = None, precision: Optional[str] = None):
        self.cache = cache
        self.film = film
        self.table = table
        self.naive = naive
        self.precision = precision or E.WORLD_MODEL_PRECISION
        # cheaper (1-ulp v_exp/v_rcp) prologue math also for the shapes the split kernel does not cover
        self.fast_math = self.precision == "f16x2" and not naive

    def w16(self, conv: nn.Conv2d) -> Optional[Tensor]:
        """Split-fp16 weight pieces when this forward runs in "f16x2" precision (else None -> exact fp32)."""
        if self.precision != "f16x2" or self.naive:
            return None
        return self.cache.conv_weight_f16x2(conv)

    def film_spec(self, norm: "AdaGroupNorm", c_start: int = 0) -> NormSpec:
        off = self.film.offset[id(norm)]
        c = norm.in_channels
        stride = self.table.stride(0)
        return NormSpec(mul=self.table[:, off + c_start:], add=self.table[:, off + c + c_start:], mul_stride=stride,
                        add_stride=stride, plus_one=True, film_cols=(off + c_start, off + c + c_start))


class FilmTable:
    """Concatenation of all AdaGroupNorm linears of a network, in module order."""

    def __init__(self, root: nn.Module):
        self.norms: List[AdaGroupNorm] = [m for m in root.modules() if isinstance(m, AdaGroupNorm)]
        self.offset: Dict[int, int] = {}
        off = 0
        for m in self.norms:
            self.offset[id(m)] = off
            off += 2 * m.in_channels
        self.total = off
        self._packed: Optional[Tuple[Tuple[int, ...], Tensor, Tensor]] = None
        self.stale_epoch = 0  # (see engine.PackCache)
        self.frees_epoch = 0
        self._audit = E.WeightAudit("FilmTable: the concatenated AdaGroupNorm linears")
        E._WEIGHT_CACHES.add(self)  # optimizer steps that do not bump `_version` (fused AdamW) invalidate through engine's hook

    def depends_on(self, param_ids) -> bool:
        return any(id(m.linear.weight) in param_ids or id(m.linear.bias) in param_ids for m in self.norms)

    def audits(self):
        return (self._audit,)

    def weights(self) -> Tuple[Tensor, Tensor]:
        ver = tuple((m.linear.weight._version, m.linear.weight.data_ptr(), m.linear.bias._version, m.linear.bias.data_ptr())
                    for m in self.norms)
        dev = self.norms[0].linear.weight.device
        if self._packed is None or self._packed[0] != ver or self._packed[1].device != dev:
            ws, bs = [m.linear.weight.detach().float() for m in self.norms], [m.linear.bias.detach().float() for m in self.norms]
            if self._packed is not None and self._packed[1].device == dev:
                # in place: a captured graph that reads the table keeps a valid pointer (engine.PackCache does the same)
                w, b = self._packed[1], self._packed[2]
                torch.cat(ws, dim=0, out=w)
                torch.cat(bs, dim=0, out=b)
            else:
                w, b = torch.cat(ws, dim=0).contiguous(), torch.cat(bs, dim=0).contiguous()
                self.frees_epoch += 1 if self._packed is not None else 0
            self._packed = (ver, w, b)
            srcs = [t for m in self.norms for t in (m.linear.weight, m.linear.bias)]
            if all(t.dtype == torch.float32 and t.is_contiguous() for t in srcs):
                self._audit.record(srcs)  # what the table was concatenated from (engine.WeightAudit)
        self._audit.tick()
        return self._packed[1], self._packed[2]

    def invalidate(self) -> None:
        """The concatenated copy is stale although no parameter version says so (engine.PackCache.invalidate)."""
        if self._packed is not None:
            self._packed = (None, self._packed[1], self._packed[2])
        self.stale_epoch += 1
        self._audit.forget()

    def refresh(self) -> None:
        self.weights()

    def compute(self, cond: Tensor) -> Tensor:
        w, b = self.weights()
        return E.linear(cond, w, b)


class GroupNorm(nn.Module):
    def __init__(self, in_channels: int) -> None:
        super().__init__()
        self.norm = nn.GroupNorm(_groups(in_channels), in_channels, eps=GN_EPS)

    def spec(self, ctx: RunCtx) -> NormSpec:
        return NormSpec(mul=ctx.cache.f32(self.norm.weight), add=ctx.cache.f32(self.norm.bias), gn_module=self.norm)


class AdaGroupNorm(nn.Module):
    """GroupNorm without affine followed by FiLM from `cond` (reference blocks.py:34-45).
    Never runs on its own: its statistics come from the producer's epilogue and its affine
    is applied inside the consumer convolution's load path."""

    def __init__(self, in_channels: int, cond_channels: int) -> None:
        super().__init__()
        self.in_channels = in_channels
        self.num_groups = _groups(in_channels)
        self.linear = nn.Linear(cond_channels, in_channels * 2)


class SelfAttention2d(nn.Module):
    def __init__(self, in_channels: int, head_dim: int = ATTN_HEAD_DIM) -> None:
        super().__init__()
        self.n_head = max(1, in_channels // head_dim)
        assert in_channels % self.n_head == 0
        self.norm = GroupNorm(in_channels)
        self.qkv_proj = conv1x1(in_channels, in_channels * 3)
        self.out_proj = conv1x1(in_channels, in_channels)
        nn.init.zeros_(self.out_proj.weight)
        nn.init.zeros_(self.out_proj.bias)

    def run(self, ctx: RunCtx, x: Act) -> Act:
        c = x.C
        spec = self.norm.spec(ctx)
        # GN affine fused into the qkv 1x1 conv's load; out_proj adds the NORMALISED input
        # back (reference blocks.py:64,72), recomputed from x + its statistics in the epilogue.
        qkv = E.conv2d([(x, nv.PROLOGUE_NORM, spec)], ctx.cache.conv_weight(self.qkv_proj), ctx.cache.conv_bias(self.qkv_proj),
                       3 * c, taps=1, want_stats=False, naive=ctx.naive, fast_math=ctx.fast_math, module=self.qkv_proj)
        y = E.attention(qkv, c, c // self.n_head)
        return E.conv2d([(Act(y, valid=x.valid), nv.PROLOGUE_NONE, None)], ctx.cache.conv_weight(self.out_proj),
                        ctx.cache.conv_bias(self.out_proj), c, taps=1, residual=x, residual_norm=spec, naive=ctx.naive,
                        fast_math=ctx.fast_math, module=self.out_proj)


class FourierFeatures(nn.Module):
    def __init__(self, cond_channels: int) -> None:
        super().__init__()
        assert cond_channels % 2 == 0
        self.register_buffer("weight", torch.randn(1, cond_channels // 2))


class Downsample(nn.Module):
    def __init__(self, in_channels: int) -> None:
        super().__init__()
        self.conv = conv3x3(in_channels, in_channels, stride=2)
        nn.init.orthogonal_(self.conv.weight)

    def run(self, ctx: RunCtx, x: Act) -> Act:
        # (w_f16: the generic stride-2 instance splits the fp32 pack on the fly and ignores it; the few-tile kernel reads it)
        return E.conv2d([(x, nv.PROLOGUE_NONE, None)], ctx.cache.conv_weight(self.conv), ctx.cache.conv_bias(self.conv),
                        self.conv.out_channels, stride=2, naive=ctx.naive, fast_math=ctx.fast_math, w_f16=ctx.w16(self.conv),
                        module=self.conv)


class Upsample(nn.Module):
    def __init__(self, in_channels: int) -> None:
        super().__init__()
        self.conv = conv3x3(in_channels, in_channels)

    def run(self, ctx: RunCtx, x: Act) -> Act:
        # nearest x2 is folded into the conv's gather: in[y >> 1][x >> 1]
        return E.conv2d([(x, nv.PROLOGUE_NONE, None)], ctx.cache.conv_weight(self.conv), ctx.cache.conv_bias(self.conv),
                        self.conv.out_channels, upsample=True, naive=ctx.naive, w_f16=ctx.w16(self.conv), module=self.conv)


class SmallResBlock(nn.Module):
    """skip(x) + Conv3x3(SiLU(GroupNorm(x)))  (reference blocks.py:116-123)."""

    def __init__(self, in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.f = nn.Sequential(GroupNorm(in_channels), nn.SiLU(inplace=True), conv3x3(in_channels, out_channels))
        self.skip_projection = nn.Identity() if in_channels == out_channels else conv1x1(in_channels, out_channels)


class ResBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, cond_channels: int, attn: bool) -> None:
        super().__init__()
        self.proj = conv1x1(in_channels, out_channels) if in_channels != out_channels else nn.Identity()
        self.norm1 = AdaGroupNorm(in_channels, cond_channels)
        self.conv1 = conv3x3(in_channels, out_channels)
        self.norm2 = AdaGroupNorm(out_channels, cond_channels)
        self.conv2 = conv3x3(out_channels, out_channels)
        self.attn = SelfAttention2d(out_channels) if attn else nn.Identity()
        nn.init.zeros_(self.conv2.weight)

    def run(self, ctx: RunCtx, xs: Sequence[Act]) -> Act:
        """xs: the channel-concatenated inputs (never materialised: the conv reads both)."""
        cout = self.conv1.out_channels
        r, proj = None, None
        if isinstance(self.proj, nn.Identity):
            assert len(xs) == 1
            r = xs[0]
        elif E.proj_fusable(xs, cout, ctx.precision, ctx.naive):
            # the up path's 128 -> 64 skip projection rides in conv2's write-out (no HBM round trip of its result)
            proj = (list(xs), ctx.w16(self.proj), ctx.cache.conv_bias(self.proj), self.proj)
        srcs, c0 = [], 0
        for a in xs:
            srcs.append((a, nv.PROLOGUE_NORM_SILU, ctx.film_spec(self.norm1, c0)))
            c0 += a.C
        h = E.conv2d(srcs, ctx.cache.conv_weight(self.conv1), ctx.cache.conv_bias(self.conv1), cout, naive=ctx.naive,
                     w_f16=ctx.w16(self.conv1), fast_math=ctx.fast_math, module=self.conv1)
        if r is None and proj is None:
            # the projection as its own launch (training, fp32).  BEHIND conv1 on purpose: the recorded backward walks the launches
            # in reverse, so the projection's input gradients exist when conv1's GroupNorm backward runs and are added inside that
            # kernel (its dskip operand) instead of by one torch addition per source afterwards -- 24 launches per training step
            r = E.conv2d([(a, nv.PROLOGUE_NONE, None) for a in xs], ctx.cache.conv_weight(self.proj),
                         ctx.cache.conv_bias(self.proj), cout, taps=1, want_stats=False, naive=ctx.naive,
                         w_f16=ctx.w16(self.proj), module=self.proj)
        h = E.conv2d([(h, nv.PROLOGUE_NORM_SILU, ctx.film_spec(self.norm2))], ctx.cache.conv_weight(self.conv2),
                     ctx.cache.conv_bias(self.conv2), cout, residual=r, naive=ctx.naive, w_f16=ctx.w16(self.conv2),
                     fast_math=ctx.fast_math, module=self.conv2, proj=proj)
        if not isinstance(self.attn, nn.Identity):
            h = self.attn.run(ctx, h)
        return h


class ResBlocks(nn.Module):
    def __init__(self, list_in_channels: List[int], list_out_channels: List[int], cond_channels: int, attn: bool) -> None:
        super().__init__()
        assert len(list_in_channels) == len(list_out_channels)
        self.in_channels = list_in_channels[0]
        self.resblocks = nn.ModuleList(
            [ResBlock(i, o, cond_channels, attn) for i, o in zip(list_in_channels, list_out_channels)])

    def run(self, ctx: RunCtx, x: Act, to_cat: Optional[List[Act]] = None) -> Tuple[Act, List[Act]]:
        outs = []
        for i, blk in enumerate(self.resblocks):
            x = blk.run(ctx, [x] if to_cat is None else [x, to_cat[i]])
            outs.append(x)
        return x, outs


class UNet(nn.Module):
    def __init__(self, cond_channels: int, depths: List[int], channels: List[int], attn_depths: List[int]) -> None:
        super().__init__()
        assert len(depths) == len(channels) == len(attn_depths)
        self._num_down = len(channels) - 1
        d_blocks, u_blocks = [], []
        for i, n in enumerate(depths):
            c1, c2 = channels[max(0, i - 1)], channels[i]
            d_blocks.append(ResBlocks([c1] + [c2] * (n - 1), [c2] * n, cond_channels, bool(attn_depths[i])))
            u_blocks.append(ResBlocks([2 * c2] * n + [c1 + c2], [c2] * n + [c1], cond_channels, bool(attn_depths[i])))
        self.d_blocks = nn.ModuleList(d_blocks)
        self.u_blocks = nn.ModuleList(reversed(u_blocks))  # deepest level first, like the reference
        self.mid_blocks = ResBlocks([channels[-1]] * 2, [channels[-1]] * 2, cond_channels, True)
        self.downsamples = nn.ModuleList([nn.Identity()] + [Downsample(c) for c in channels[:-1]])
        self.upsamples = nn.ModuleList([nn.Identity()] + [Upsample(c) for c in reversed(channels[:-1])])

    def run(self, ctx: RunCtx, x: Act) -> Act:
        m = 8 * 2 ** self._num_down  # every level must itself be a multiple of the kernels' 8-pixel tile
        if x.valid is None:
            assert x.shape[1] % m == 0 and x.shape[2] % m == 0, \
                (f"the native U-Net tiles every level in 8x8 pixel blocks: H, W must be multiples of {m} (8 x 2**num_down), got "
                 f"{x.shape[1]}x{x.shape[2]}; other sizes go in as the VALID EXTENT of a larger buffer "
                 "(engine.padded_extent, InnerModel.run(valid=...))")
        else:
            # the reference pads to a multiple of 2**num_down and crops (blocks.py:227-229,247): InnerModel.run has done the
            # padding (zero rows / columns inside the valid extent); every level halves the extent exactly
            m2 = 2 ** self._num_down
            assert x.valid[0] % m2 == 0 and x.valid[1] % m2 == 0 and x.shape[1] % (2 * m) == 0 and x.shape[2] % (2 * m) == 0, (x.valid, x.shape)
        skips: List[List[Act]] = []
        last = len(self.d_blocks) - 1
        chained = False
        for li, (blocks, down) in enumerate(zip(self.d_blocks, self.downsamples)):
            if not isinstance(down, nn.Identity):
                x = down.run(ctx, x)
            if li == last and self._chain_eligible(ctx, x):
                # deepest level at 8x8: its down blocks, the mid blocks and its up blocks as ONE launch (dmd_lowres.hip)
                x = self._run_lowres_chain(ctx, x)
                chained = True
                break
            x_down = x
            x, outs = blocks.run(ctx, x)
            skips.append([x_down] + outs)
        if not chained:
            x, _ = self.mid_blocks.run(ctx, x)
        for ui, (blocks, up) in enumerate(zip(self.u_blocks, self.upsamples)):
            if chained and ui == 0:
                continue
            skip = skips[len(self.u_blocks) - 1 - ui]
            if not isinstance(up, nn.Identity):
                x = up.run(ctx, x)
            x, _ = blocks.run(ctx, x, skip[::-1])
        return x

    # -- the 8x8 level as one launch ------------------------------------------------------------------------------
    def _chain_blocks(self) -> List["ResBlock"]:
        return list(self.d_blocks[-1].resblocks) + list(self.mid_blocks.resblocks) + list(self.u_blocks[0].resblocks)

    def _chain_eligible(self, ctx: RunCtx, x: Act) -> bool:
        """dmd_lowres_chain covers: split-fp16 inference (no recording for a backward), 8x8 x 64 channels at the deepest
        level with 64-channel neighbours, at most two down blocks (three skip slots) and eight blocks in all."""
        if not (int(LOWRES_CHAIN) & 1) or ctx.precision != "f16x2" or ctx.naive or E.TAPE is not None or E._USE_NAIVE:
            return False
        if len(self.d_blocks) < 2 or tuple(x.shape[1:]) != (8, 8, 64) or x.valid is not None:
            return False
        blks = self._chain_blocks()
        nd = len(self.d_blocks[-1].resblocks)
        if nd > 2 or len(blks) > nv.CHAIN_MAX_BLOCKS or len(self.u_blocks[0].resblocks) != nd + 1:
            return False
        for i, b in enumerate(blks):
            cat = i >= len(blks) - (nd + 1)
            if b.conv1.out_channels != 64 or b.conv1.in_channels != (128 if cat else 64) or isinstance(b.proj, nn.Identity) == cat:
                return False
        return True

    def _run_lowres_chain(self, ctx: RunCtx, x: Act) -> Act:
        import ctypes as C

        cache = ctx.cache
        blks = self._chain_blocks()
        nd = len(self.d_blocks[-1].resblocks)
        n_up = nd + 1
        p = nv.LowresChainParams()
        n = x.shape[0]
        out = torch.empty_like(x.t)
        p.N, p.nblocks, p.input_save_slot = n, len(blks), 0  # slot 0 = the level's input, slot j + 1 = output of down block j
        p.x, p.out = nv.ptr(x.t), nv.ptr(out)
        p.table, p.table_stride = nv.ptr(ctx.table), ctx.table.stride(0)
        keep = []  # tensors whose pointers are in `p` (the caches own them; this list only documents the lifetime)
        for i, b in enumerate(blks):
            cb = p.blocks[i]
            up_i = i - (len(blks) - n_up)
            cb.skip_slot = nd - up_i if up_i >= 0 else -1  # up block i concatenates skips[::-1][i]
            cb.save_slot = i + 1 if i < nd else -1
            o1, c1 = ctx.film.offset[id(b.norm1)], b.norm1.in_channels
            cb.film1_mul[0], cb.film1_add[0] = o1, o1 + c1
            cb.film1_mul[1], cb.film1_add[1] = o1 + 64, o1 + c1 + 64
            o2, c2 = ctx.film.offset[id(b.norm2)], b.norm2.in_channels
            cb.film2_mul, cb.film2_add = o2, o2 + c2
            ts = [cache.conv_weight_f16x2(b.conv1), cache.conv_weight_f16x2(b.conv2), cache.conv_bias(b.conv1), cache.conv_bias(b.conv2)]
            cb.w1, cb.w2, cb.b1, cb.b2 = (nv.ptr(t) for t in ts)
            if not isinstance(b.proj, nn.Identity):
                tp = [cache.conv_weight_f16x2(b.proj), cache.conv_bias(b.proj)]
                cb.wproj, cb.bproj = nv.ptr(tp[0]), nv.ptr(tp[1])
                ts += tp
            if not isinstance(b.attn, nn.Identity):
                a = b.attn
                cb.has_attn = 1
                qkv = [cache.get(a.qkv_proj.weight, f"qkv16[{k}]", lambda w, k=k: nv.pack_conv_weight_f16x2(w.detach().float()[64 * k:64 * k + 64].contiguous()))
                       for k in range(3)]
                ta = qkv + [cache.conv_weight_f16x2(a.out_proj), cache.f32(a.norm.norm.weight), cache.f32(a.norm.norm.bias),
                            cache.f32(a.qkv_proj.bias), cache.conv_bias(a.out_proj)]
                cb.wq, cb.wk, cb.wv, cb.wo, cb.gn_gamma, cb.gn_beta, cb.bqkv, cb.bo = (nv.ptr(t) for t in ta)
                ts += ta
            keep.append(ts)
        if nv.PROFILER is not None:
            flops = 0.0
            for b in blks:
                flops += 2.0 * n * 64 * 64 * 9 * (b.conv1.in_channels + 64)
                if not isinstance(b.proj, nn.Identity):
                    flops += 2.0 * n * 64 * 64 * 128
                if not isinstance(b.attn, nn.Identity):
                    flops += 2.0 * n * 64 * 64 * 64 * 4 + 4.0 * n * 64 * 64 * 64
            nv.PROFILER.annotate("lowres_chain_kernel", flops, 8.0 * x.t.numel())
        nv.check(nv.lib().dmd_lowres_chain(C.byref(p), nv.stream()), "dmd_lowres_chain")
        del keep
        return Act(out)
