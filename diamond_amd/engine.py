"""Host-side op layer over the C ABI: NHWC activations that carry their GroupNorm partial
statistics, packed-weight cache, and one Python call per kernel launch.

This is plumbing only -- every FLOP of the U-Net / encoders happens in libdiamond_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
import time
import weakref
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import native as nv

_USE_NAIVE = os.environ.get("DIAMOND_CONV_IMPL", "mfma") == "naive"  # debugging aid only (still HIP)
# Arithmetic of the no-grad world-model convolutions (denoiser / reward-end model):
#   "f16x2": split-fp32 operands on the f16 matrix cores where dmd_conv2d_f16x2_eligible (default);
#   "f32"  : exact fp32 MFMA everywhere.
# The actor-critic encoder has its own switch (ac_native.AC_PRECISION, DIAMOND_AC_PRECISION, default "f16x2" for its
# forward and dgrad convolutions; its weight gradients run on the split instance of the wgrad kernel too).
WORLD_MODEL_PRECISION = os.environ.get("DIAMOND_CONV_PRECISION", "f16x2")


LaunchProfiler = nv.LaunchProfiler  # per-launch HIP-event timing of every C-ABI launch: install with `native.PROFILER = ...`


@dataclass
class ConvRecord:
    """One dmd_conv2d launch of a recorded forward (training): everything its backward needs."""
    srcs: list  # [(Act, prologue, NormSpec | None)]
    module: nn.Module  # the nn.Conv2d whose weight / bias this launch used
    taps: int
    stride: int
    upsample: bool
    residual: Optional["Act"]
    residual_norm: Optional["NormSpec"]
    out: "Act"
    out_nchw: bool


@dataclass
class AttnRecord:
    qkv: "Act"
    out: Tensor
    c: int
    head_dim: int


# recording tape of the current forward (None: inference, nothing is recorded)
TAPE: Optional[list] = None


def kernel_key(p) -> str:
    """Name of the kernel instantiation dmd_conv2d launches for these parameters, spelled like rocprofv3's kernel
    trace (so bench.py's records, profiles/*_kernel_stats.csv and profiles/*pmc*.json share one key)."""
    buf = C.create_string_buffer(128)
    nv.check(nv.lib().dmd_conv2d_kernel_name(C.byref(p), buf, len(buf)), "dmd_conv2d_kernel_name")
    return buf.value.decode()


@dataclass
class Act:
    """NHWC fp32 activation (N, H, W, C) + fp64 partial GroupNorm sums (N, C/32, T, 2)."""
    t: Tensor
    stats: Optional[Tensor] = None
    tiles: int = 0
    needs_grad: bool = True  # False: network input (the training backward stops here)
    # (vh, vw): the part of the (H, W) buffer that exists (include/diamond_hip.h: VALID EXTENT) -- image sizes whose U-Net
    # levels are not multiples of the kernels' tiles live inside a larger buffer; None: all of it
    valid: Optional[Tuple[int, int]] = None

    @property
    def shape(self):
        return self.t.shape

    @property
    def C(self) -> int:
        return self.t.shape[3]


@dataclass
class NormSpec:
    """How a consumer normalises an Act: FiLM (mul/add rows of the batched AdaGN table) or
    GroupNorm affine parameters."""
    mul: Optional[Tensor]
    add: Optional[Tensor]
    mul_stride: int = 0
    add_stride: int = 0
    plus_one: bool = False
    # where the gradients of mul / add go in the training backward (unet_train.py): columns of the batched FiLM
    # table (mul_col, add_col), or the nn.GroupNorm module whose affine parameters these are
    film_cols: Optional[Tuple[int, int]] = None
    gn_module: Optional[nn.Module] = None

    def to_native(self, a: Act) -> nv.Norm:
        assert a.stats is not None, "normalising an activation that has no statistics"
        return nv.make_norm(a.stats, a.tiles, self.mul, self.add, self.mul_stride, self.add_stride, self.plus_one)


class _PackEntry:
    """One kernel-layout copy of a convolution parameter: the job that rebuilds it and the stamp it was last built from."""
    __slots__ = ("ref", "out", "stamp", "job", "elems")

    def __init__(self, p: Tensor, out: Tensor, job: "nv.PackJob", elems: int) -> None:
        self.ref, self.out, self.stamp, self.job, self.elems = weakref.ref(p), out, None, job, elems


def _stamp(p: Tensor) -> Tuple:
    # storage pointer: catches `p.data = other`; the entry's weakref: an id() re-used by a new tensor is not a hit
    return (p._version, p.data_ptr(), p.device)


# Every live cache of kernel-layout weight copies (PackCache below, blocks.FilmTable).  Staleness is normally read off
# `Tensor._version`, which torch's in-place ops bump -- but not everything that updates a parameter does: the FUSED optimizers
# (`torch.optim.AdamW(fused=True)`, Adam, SGD: `torch._fused_adamw_` writes through TensorLists) leave `_version` alone, and a
# model trained with one would keep running on the weights of its first forward, silently.  So a process-wide optimizer
# post-step hook marks stale every cache that holds a copy of a parameter of the optimizer that just stepped (a Python set
# intersection per step; the copies are rebuilt in place by their next use, as after a version bump).
_WEIGHT_CACHES: "weakref.WeakSet" = weakref.WeakSet()


def _optimizer_stepped(optimizer, args, kwargs) -> None:
    ids = {id(p) for group in optimizer.param_groups for p in group["params"]}
    for cache in list(_WEIGHT_CACHES):
        if cache.depends_on(ids):
            cache.invalidate()


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_post_hook  # noqa: E402

_register_step_post_hook(_optimizer_stepped)



class WeightAudit:
    """Always-on guard of the packed weight copies of ONE cache group (ABI v9, `dmd_checksums`).

    A copy is rebuilt when its parameter's (`_version`, storage pointer) stamp changes, or when the optimizer hook above says
    so.  A write that shows in neither -- `p.data.copy_(...)`, `p.data.fill_`, a collective on `.data` -- used to be a
    documented hole: the kernels would go on computing with the old weights.  Now every (re)build is followed by a launch that
    records an exact fingerprint of the SOURCE values the copy was built from (device-resident, next to the copy), and every
    AUDIT_EVERY lookups one launch fingerprints the live parameters and compares: a mismatch on a row whose stamp still says
    "unchanged" is a silent write.  The comparison's answer travels to a pinned buffer asynchronously and is looked at by the
    next tick / by `check_weight_audits()` (WorldModelEnv calls it right after its per-step host synchronisation): the culprit
    RAISES instead of training on stale weights.  Cost: one small launch per rebuild, one per few thousand lookups.
    Capturable: a replayed training step (GraphedTrainStep) rebuilds copies and re-records fingerprints inside the graph."""

    AUDIT_EVERY = 16384  # (lookups of one cache group between audits: about one audit per imagined window; 14 us per launch)
    # ... and a byte budget: an audit reads every audited parameter once (eight workgroups per tensor), which is 14 us for the 64x64
    # networks but 6.5 ms where `lstm.weight_ih` is 2048 x 32768 (configs[4]: 2 % of a window) -- at most AUDIT_BYTES_PER_S of
    # parameter bytes are re-read per second of wall time, i.e. large models are audited every few windows instead of every window
    AUDIT_BYTES_PER_S = 256e6
    AUDIT_FREE_BYTES = 64e6  # (cache groups below this size -- every one of the 64x64 networks -- are audited by lookup count alone)
    CAPACITY = 4096

    def __init__(self, what: str) -> None:
        self.what = what
        self._refs: List["weakref.ref"] = []
        self._row: Dict[int, int] = {}
        self._ptr: List[int] = []
        self._stamps: List[Optional[Tuple]] = []
        self._table: Optional[Tensor] = None
        self._rec: Optional[Tensor] = None
        self._live: Optional[Tensor] = None
        self._pending = None
        self.ticks = 0
        self.audits = 0
        self._last_run = 0.0  # time.monotonic() of the last tick-driven audit

    @staticmethod
    def _capturing() -> bool:
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()

    def _ensure_row(self, p: Tensor) -> int:
        i = self._row.get(id(p))
        if i is not None and self._refs[i]() is p and self._ptr[i] == p.data_ptr() and self._table is not None and self._table.device == p.device:
            return i
        assert not self._capturing(), (f"WeightAudit({self.what}): a parameter that is new, or whose storage moved, inside a hipGraph capture "
                                       "(its job-table row is a host upload: warm the step up eagerly first)")
        if self._table is None or self._table.device != p.device:
            self._table = torch.zeros(self.CAPACITY * C.sizeof(nv.ChecksumJob), dtype=torch.uint8, device=p.device)
            self._rec = torch.zeros(self.CAPACITY, nv.CHECKSUM_PARTS, dtype=torch.int64, device=p.device)
            self._live = torch.zeros_like(self._rec)
            self._refs, self._row, self._ptr, self._stamps, self._pending = [], {}, [], [], None
            i = None
            # (the comparison's two torch kernels are loaded NOW, while the caches are built: their first use costs 20-100 ms of
            #  lazy code-object loading, which otherwise shows as one hole in the middle of some later window -- profiles/r06d_*)
            (self._live[:1] != self._rec[:1]).any(dim=1)
        if i is None or self._refs[i]() is not p:
            # a row of its own: one whose parameter is gone is taken over (replaced parameter objects must not use the table up)
            i = next((j for j, ref in enumerate(self._refs) if ref() is None), None)
            if i is None:
                i = len(self._refs)
                assert i < self.CAPACITY, f"WeightAudit({self.what}): more than {self.CAPACITY} live parameters"
                self._refs.append(None)
                self._ptr.append(0)
                self._stamps.append(None)
            self._row = {k: v for k, v in self._row.items() if v != i}
            self._refs[i] = weakref.ref(p)
            self._row[id(p)] = i
        assert p.element_size() == 4 and p.is_contiguous(), "audited parameters are contiguous 32-bit tensors"
        job = nv.ChecksumJob()
        job.src, job.words = p.data_ptr(), p.numel()
        sz = C.sizeof(nv.ChecksumJob)
        self._table[i * sz:(i + 1) * sz].copy_(torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8))
        self._ptr[i] = p.data_ptr()
        self._stamps[i] = None
        return i

    def _launch(self, first: int, count: int, out: Tensor) -> None:
        sz = C.sizeof(nv.ChecksumJob)
        nv.check(nv.lib().dmd_checksums(self._table.data_ptr() + first * sz, count, out.data_ptr() + first * nv.CHECKSUM_PARTS * 8,
                                        nv.stream()), "dmd_checksums")

    def record(self, params) -> None:
        """The copies of these parameters were (re)built just now, on the current stream: record what they were built from."""
        rows = sorted(self._ensure_row(p) for p in params)
        if not rows:
            return
        # (contiguous runs: the jobs of a PackCache refresh are all of its rows -> one launch)
        start = prev = rows[0]
        for r in rows[1:] + [None]:
            if r is None or r != prev + 1:
                self._launch(start, prev - start + 1, self._rec)
                start = r
            prev = r
        for p in params:
            self._stamps[self._row[id(p)]] = _stamp(p)

    def forget(self) -> None:
        """The cache was invalidated by someone who knows the values changed (optimizer hook, PackCache.invalidate): nothing is
        believed fresh until it is rebuilt and recorded again."""
        self._stamps = [None] * len(self._stamps)
        self._pending = None

    def tick(self) -> None:
        self.ticks += 1
        if self.ticks % self.AUDIT_EVERY == 0:
            now = time.monotonic()
            nbytes = 4 * sum(p.numel() for p in (r() for r in self._refs) if p is not None)
            if nbytes < self.AUDIT_FREE_BYTES or now - self._last_run >= nbytes / self.AUDIT_BYTES_PER_S:
                self._last_run = now
                self.run()

    def run(self) -> None:
        """Fingerprint the live parameters and compare with what the copies were built from (asynchronous; `check` reads it)."""
        if self._table is None or not self._refs or self._capturing():
            return
        self.check()
        n = len(self._refs)
        believed = []
        for i, ref in enumerate(self._refs):
            p = ref()
            believed.append(p is not None and self._stamps[i] is not None and self._stamps[i] == _stamp(p))
        if not any(believed):
            return
        # only runs of believed rows are read: the job of a row whose parameter is gone (or was replaced unseen) points at
        # memory that may have been returned to the driver since
        start = None
        for i, b in enumerate(believed + [False]):
            if b and start is None:
                start = i
            elif not b and start is not None:
                self._launch(start, i - start, self._live)
                start = None
        bad = (self._live[:n] != self._rec[:n]).any(dim=1)
        self.audits += 1
        if bad.is_cuda:
            host = torch.empty(n, dtype=torch.bool).pin_memory()
            host.copy_(bad, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = bad.clone(), None
        self._pending = (ev, host, believed)

    def check(self, wait: bool = False) -> None:
        if self._pending is None:
            return
        ev, host, believed = self._pending
        if ev is not None:
            if wait:
                ev.synchronize()
            elif not ev.query():
                return
        self._pending = None
        rows = [i for i, (b, ok) in enumerate(zip(host.tolist(), believed)) if b and ok]
        # (a row counts only if its stamp STILL says unchanged: a visible update between audit and check is not a silent one)
        rows = [i for i in rows if self._refs[i]() is not None and self._stamps[i] is not None and self._stamps[i] == _stamp(self._refs[i]())]
        if rows:
            shapes = [tuple(self._refs[i]().shape) for i in rows[:4]]
            self.forget()
            raise RuntimeError(
                f"stale packed weights ({self.what}): {len(rows)} parameter(s), e.g. of shape {shapes}, changed WITHOUT a version bump "
                "(`p.data.copy_`, `p.data.fill_`, a collective on `.data`, ...) after their kernel-layout copies were built, and the kernels "
                "have been reading the old values since.  Write through the parameter itself (`with torch.no_grad(): p.copy_(x)`) or call "
                "`PackCache.invalidate()` / `FilmTable.invalidate()` after such a write.")


def check_weight_audits(wait: bool = False) -> None:
    """Raise if any cache's last audit found a silent parameter write (cheap: an event query per cache with an audit in flight)."""
    for cache in list(_WEIGHT_CACHES):
        for a in cache.audits():
            a.check(wait)


def run_weight_audits() -> None:
    for cache in list(_WEIGHT_CACHES):
        for a in cache.audits():
            a.run()


class PackCache:
    """Kernel-layout copies of nn.Module parameters, refreshed when a parameter changes
    (optimizer steps bump `Tensor._version`, or are seen by the hook above).  Parameters keep the reference's OIHW / (out,in)
    layouts so checkpoints stay interchangeable (agent.py:48-62).

    The convolution copies (packed fp32, split-fp16 pieces, padded biases, and the transposed / sliced weights of the data
    gradient) are JOBS of one table in device memory, all rebuilt together by ONE launch (dmd_pack_jobs) as soon as any of
    them is stale: a training step changes every parameter, and packing copy by copy was ~640 launches per denoiser step.
    The outputs are rebuilt in place (stable pointers: a captured training step keeps reading the same buffers)."""

    def __init__(self) -> None:
        self._store: Dict[Tuple[int, str], Tuple[Optional[Tuple], "weakref.ref", Any, Any]] = {}  # (stamp, param, copy, builder)
        self._jobs: Dict[Tuple, _PackEntry] = {}
        self._table: Optional[Tensor] = None  # (njobs * sizeof(dmd_pack_job)) bytes on the device
        self._table_entries: List[_PackEntry] = []
        self._max_elems = 0
        # What the owner of a captured graph that reads these copies directly (DiffusionSampler.sample_ring_graphed) watches:
        # `stale_epoch` counts invalidations no `Tensor._version` shows -- refresh() before the next replay; `frees_epoch`
        # counts copies whose BUFFER was replaced or dropped -- graphs holding the old pointer are void.  (Rebuilding a copy
        # in place changes neither: the graph reads the new values through the same pointer.)
        self.stale_epoch = 0
        self.frees_epoch = 0
        self._audit_jobs, self._audit_store = WeightAudit("PackCache: convolution copies"), WeightAudit("PackCache: other copies")
        _WEIGHT_CACHES.add(self)

    def audits(self):
        return (self._audit_jobs, self._audit_store)

    def depends_on(self, param_ids) -> bool:
        """Does this cache hold a copy of any of these parameters (ids)?"""
        return any(k[0] in param_ids for k in self._jobs) or any(k[0] in param_ids for k in self._store)

    def invalidate(self) -> None:
        """Every packed copy is stale.  Needed after writes that do not bump `Tensor._version` (anything done through
        `p.data`: `.data.copy_`, `.data.fill_`, collectives on `.data`; optimizer steps are covered by the hook above).  Buffers and the job table are kept: the next use
        rebuilds the copies IN PLACE (a captured graph that reads them keeps valid pointers; it is the weights EPOCH that
        tells its owner that what it captured may be out of date)."""
        for key, hit in list(self._store.items()):
            self._store[key] = (None, hit[1], hit[2], hit[3])
        for ent in self._jobs.values():
            ent.stamp = None
        self.stale_epoch += 1
        self._audit_jobs.forget()
        self._audit_store.forget()

    def get(self, p: Tensor, kind: str, fn):
        key = (id(p), kind)
        hit = self._store.get(key)
        if hit is None or hit[0] != _stamp(p) or hit[1]() is not p:
            hit = self._rebuild(key, p, fn, hit)
        self._audit_store.tick()
        return hit[2]

    def _rebuild(self, key, p: Tensor, fn, hit):
        new = fn(p)
        old = hit[2] if (hit is not None and hit[1]() is p) else None
        if isinstance(old, Tensor) and isinstance(new, Tensor) and new.data_ptr() != p.data_ptr() and old.data_ptr() != p.data_ptr() \
                and old.shape == new.shape and old.dtype == new.dtype and old.device == new.device and old.is_contiguous():
            old.copy_(new)  # same buffer: whoever holds its pointer (a captured graph, a device-side table) sees the update
            new = old
        elif isinstance(old, Tensor) and isinstance(new, Tensor) and old.data_ptr() == new.data_ptr():
            pass  # a "copy" that IS the parameter's storage (f32() of a contiguous fp32 parameter): nothing was replaced
        elif hit is not None:
            self.frees_epoch += 1
        hit = (_stamp(p), weakref.ref(p), new, fn)
        self._store[key] = hit
        if isinstance(new, Tensor) and new.data_ptr() != p.data_ptr() and p.element_size() == 4 and p.is_contiguous():
            self._audit_store.record([p])  # (a copy that IS the parameter cannot be stale)
        return hit

    def refresh(self) -> None:
        """Rebuild every stale copy NOW (on the current stream, capturable): the captured training step calls this behind its
        optimizer update, so that a replay leaves the packed copies equal to the parameters it leaves."""
        if self._jobs and any(e.ref() is not None and e.stamp != _stamp(e.ref()) for e in self._jobs.values()):
            self._refresh()
        for key, hit in list(self._store.items()):
            p = hit[1]()
            if p is None:
                del self._store[key]
                self.frees_epoch += 1
            elif hit[0] != _stamp(p):
                self._rebuild(key, p, hit[3], hit)

    # -- convolution copies: one table, one launch --------------------------------------------------------------------
    def _conv_job(self, p: Tensor, kind: int, cout_pad: int, transposed: bool = False, c0: int = 0, c1: int = 0,
                  cin_pad_to: int = 0) -> Tensor:
        key = (id(p), kind, cout_pad, transposed, c0, c1, cin_pad_to)
        ent = self._jobs.get(key)
        if ent is not None and ent.ref() is p and ent.stamp == _stamp(p):
            self._audit_jobs.tick()
            return ent.out
        if ent is None or ent.ref() is not p or ent.out.device != p.device:
            assert p.dtype == torch.float32 and p.is_contiguous(), "convolution parameters are contiguous fp32"
            job = nv.PackJob()
            job.src, job.kind, job.transposed, job.c0, job.c1 = p.data_ptr(), kind, int(transposed), c0, c1
            if kind == nv.PACK_BIAS:
                job.Cout, job.Cin, job.k, job.CoutPad, job.CinPad = p.numel(), 1, 1, cout_pad, 16
                out = torch.empty(cout_pad, device=p.device, dtype=torch.float32)
                elems = cout_pad
            else:
                cout, cin, k, _ = p.shape
                job.Cout, job.Cin, job.k = cout, cin, k
                cin_l = max(cout, cin_pad_to) if transposed else cin  # input channels of the (transposed) weight
                job.CoutPad, job.CinPad = cout_pad, (cin_l + 15) // 16 * 16
                elems = job.CinPad * k * k * cout_pad
                out = torch.empty(elems * (2 if kind == nv.PACK_F16X2 else 1), device=p.device,
                                  dtype=torch.float16 if kind == nv.PACK_F16X2 else torch.float32)
            job.dst = out.data_ptr()
            if ent is not None:
                self.frees_epoch += 1
            ent = _PackEntry(p, out, job, elems)
            self._jobs[key] = ent
            self._table = None  # (rebuilt with all rows by the first full refresh)
            self._pack_one(ent)  # a new copy alone: registration is O(1) launches, not a rebuild of everything so far
            return ent.out
        self._refresh()  # stale: an optimizer step / load changed every parameter -> all copies, one launch
        return ent.out

    @staticmethod
    def _no_capture(what: str) -> None:
        assert not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing(), \
            f"PackCache: {what} inside a hipGraph capture (a synchronous upload; warm the step up eagerly first)"

    def _pack_one(self, e: _PackEntry) -> None:
        self._no_capture("a new packed copy was requested")
        table = torch.frombuffer(bytearray(bytes(e.job)), dtype=torch.uint8).to(e.out.device)
        nv.check(nv.lib().dmd_pack_jobs(nv.ptr(table), 1, e.elems, nv.stream()), "dmd_pack_jobs")
        e.stamp = _stamp(e.ref())
        self._audit_jobs.record([e.ref()])

    def _refresh(self) -> None:
        """Rebuild every registered copy with one launch (all of them: whoever changed one parameter changed them all)."""
        if self._table is not None:
            # a parameter whose storage was swapped (`p.data = other`, `module.to(...)`) or that died: its rows still name
            # the old storage -- patch them before anything is rebuilt from it
            for e in self._table_entries:
                p = e.ref()
                if p is None or e.job.src != p.data_ptr() or e.out.device != p.device:
                    self._table = None
                    break
        if self._table is None:
            for k, e in list(self._jobs.items()):
                p = e.ref()
                if p is None or e.out.device != p.device:
                    del self._jobs[k]  # (a moved parameter registers a new job on its next lookup)
                    self.frees_epoch += 1
                elif e.job.src != p.data_ptr():
                    e.job.src = p.data_ptr()
            self._table_entries = list(self._jobs.values())
            if not self._table_entries:
                return
            self._no_capture("the job table has to be uploaded")
            raw = b"".join(bytes(e.job) for e in self._table_entries)
            dev = self._table_entries[0].out.device
            self._table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
            self._max_elems = max(e.elems for e in self._table_entries)
        nv.check(nv.lib().dmd_pack_jobs(nv.ptr(self._table), len(self._table_entries), self._max_elems, nv.stream()), "dmd_pack_jobs")
        live = {}
        for e in self._table_entries:
            p = e.ref()
            e.stamp = None if p is None else _stamp(p)
            if p is not None:
                live[id(p)] = p
        self._audit_jobs.record(list(live.values()))

    def conv_weight(self, conv: nn.Conv2d, cout_padded: Optional[int] = None) -> Tensor:
        return self._conv_job(conv.weight, nv.PACK_F32, cout_padded or nv.cout_pad(conv.out_channels))

    def conv_weight_f16x2(self, conv: nn.Conv2d) -> Optional[Tensor]:
        """Split-fp16 pieces of a 3x3 / 1x1 weight with 32 or 64 output channels (None for other shapes); stride 2 (Downsample)
        has the same layout -- only the few-tile kernel reads it there."""
        if conv.out_channels not in (32, 64) or conv.kernel_size not in ((3, 3), (1, 1)) or conv.stride not in ((1, 1), (2, 2)):
            return None
        if conv.in_channels > (128 if conv.out_channels == 64 else 64):
            return None
        return self._conv_job(conv.weight, nv.PACK_F16X2, conv.out_channels)

    def conv_weight_f16x2_head(self, conv: nn.Conv2d) -> Optional[Tensor]:
        """Split-fp16 pieces of a few-output-channel 3x3 head (conv_out: 64 -> 3), zero-padded to 32 couts."""
        if conv.out_channels > 4 or conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.in_channels > 64:
            return None
        return self._conv_job(conv.weight, nv.PACK_F16X2, 32)

    def conv_bias(self, conv: nn.Conv2d, cout_padded: Optional[int] = None) -> Optional[Tensor]:
        if conv.bias is None:
            return None
        n = cout_padded or nv.cout_pad(conv.out_channels)
        b = conv.bias
        if n == b.numel() and b.dtype == torch.float32 and b.is_contiguous():
            return b.detach()  # nothing to pad: the kernels read the parameter itself
        return self._conv_job(b, nv.PACK_BIAS, n)

    def dgrad_weight(self, conv: nn.Conv2d, c0: int, c1: int, cin_pad_to: int = 0, f16x2: bool = False) -> Tensor:
        """Packed weight of the data-gradient (transposed) convolution restricted to input channels [c0, c1) of `conv`:
        W'[ci - c0][co][ky][kx] = W[co][ci][K-1-ky][K-1-kx], its input channels (= conv's outputs) zero-padded to
        `cin_pad_to` (conv_out: 3 -> 16)."""
        cout_t = c1 - c0
        return self._conv_job(conv.weight, nv.PACK_F16X2 if f16x2 else nv.PACK_F32, cout_t if f16x2 else nv.cout_pad(cout_t),
                              True, c0, c1, cin_pad_to)

    def f32(self, p: Tensor) -> Tensor:
        return self.get(p, "f32", lambda t: t.detach().float().contiguous())


def new_stats(n: int, c: int, tiles: int, device) -> Tensor:
    return torch.empty(n, max(1, c // nv.GN_GROUP), tiles, 2, device=device, dtype=torch.float64)


CONV_CIN_MAX = 256  # input channels (all sources together) of ONE dmd_conv2d launch: DMD_CIN_MAX, csrc/dmd_conv.hip


def _channel_slice(a: Act, prologue: int, norm: Optional[NormSpec], c0: int, c1: int) -> Tuple[Act, int, Optional[NormSpec]]:
    """Channels [c0, c1) of a source as a source of its own: a contiguous copy of the slice, the partial GroupNorm sums of its
    groups (whole 32-channel groups: the statistics of a group do not depend on the others) and the matching columns of its
    multiplicative / additive parameters (views: the kernels take pointer + row stride)."""
    sub = Act(a.t[..., c0:c1].contiguous(), None, 0, a.needs_grad, a.valid)
    if prologue == nv.PROLOGUE_NONE:
        return sub, prologue, None
    g = nv.GN_GROUP
    assert c0 % g == 0 and c1 % g == 0 and a.stats is not None, f"normalised source of {a.C} channels cut at [{c0}, {c1}): not whole GroupNorm groups"
    sub.stats, sub.tiles = a.stats[:, c0 // g:c1 // g].contiguous(), a.tiles
    cut = lambda t: None if t is None else t[..., c0:]
    return sub, prologue, NormSpec(cut(norm.mul), cut(norm.add), norm.mul_stride, norm.add_stride, norm.plus_one)


def _conv2d_wide(srcs, w_packed, bias, cout, *, taps, stride, upsample, residual, residual_norm, want_stats, out_nchw, cout_padded,
                 naive, fast_math, module) -> Act:
    """A convolution over more than CONV_CIN_MAX input channels (U-Nets wider than the default configuration: the reference's
    UNet takes any `channels` list, blocks.py:183-222, and its up path concatenates two of them, :174) as a chain of launches
    over <= CONV_CIN_MAX channels each: the contraction is a sum over input channels, so
        out = conv(piece_0) + bias + residual;   out = conv(piece_i) + out   (i = 1 ...),
    the last launch emitting the GroupNorm partial sums of the finished output.  The packed weight is K-major
    ([Cin / 16][taps][CoutPad][16], dmd_pack_conv_weight): a piece's weights are a contiguous range of it, nothing is re-packed.
    A source wider than the limit on its own goes in as contiguous channel slices (_channel_slice).  The pieces run on the generic
    kernel (conv_mfma_kernel: its split-fp16 instance where the caller's precision asks for it, exact fp32 otherwise -- the
    wave-specialised kernel covers Cin <= 128); one launch per piece -- correctness for wide configurations, not their fast path."""
    global TAPE
    assert not out_nchw, "wide convolution with an NCHW result"
    pieces = []  # (source triple, channels)
    for a, prologue, norm in srcs:
        if a.C <= CONV_CIN_MAX:
            pieces.append(((a, prologue, norm), a.C))
        else:
            for c0 in range(0, a.C, CONV_CIN_MAX):
                c1 = min(a.C, c0 + CONV_CIN_MAX)
                pieces.append((_channel_slice(a, prologue, norm, c0, c1), c1 - c0))
    launches, k0 = [], 0  # ([source triples], first input channel): consecutive pieces, at most two sources per launch
    for triple, c in pieces:
        assert c % 16 == 0, f"source of {c} channels inside a wide convolution (multiples of 16)"
        if launches and len(launches[-1][0]) < 2 and launches[-1][2] + c <= CONV_CIN_MAX:
            launches[-1][0].append(triple)
            launches[-1][2] += c
        else:
            launches.append([[triple], k0, c])
        k0 += c
    cp = cout_padded or nv.cout_pad(cout)
    tape, TAPE = TAPE, None  # the chain is ONE convolution to the recorded backward (below)
    try:
        acc = None
        for i, (group, first, _) in enumerate(launches):
            last = i == len(launches) - 1
            acc = conv2d(group, w_packed[(first // 16) * taps * cp * 16:], bias if i == 0 else None, cout, taps=taps, stride=stride,
                         upsample=upsample, residual=residual if i == 0 else acc, residual_norm=residual_norm if i == 0 else None,
                         want_stats=want_stats and last, cout_padded=cout_padded, naive=naive, fast_math=fast_math)
    finally:
        TAPE = tape
    if TAPE is not None:
        assert module is not None, "recording a conv launch that does not name its nn.Conv2d"
        TAPE.append(ConvRecord(list(srcs), module, taps, stride, upsample, residual, residual_norm, acc, out_nchw))
    return acc


def conv2d(
    srcs: Sequence[Tuple[Act, int, Optional[NormSpec]]],  # (activation, prologue, norm)
    w_packed: Tensor,
    bias: Optional[Tensor],
    cout: int,
    *,
    taps: int = 9,
    stride: int = 1,
    upsample: bool = False,
    residual: Optional[Act] = None,
    residual_norm: Optional[NormSpec] = None,
    want_stats: bool = True,
    out_nchw: bool = False,
    cout_padded: Optional[int] = None,
    naive: Optional[bool] = None,
    w_f16: Optional[Tensor] = None,
    fast_math: bool = False,
    module: Optional[nn.Module] = None,
    proj: Optional[Tuple[Sequence[Act], Tensor, Optional[Tensor]]] = None,  # fused skip projection: (sources, w_f16 (k = 1), bias)
) -> Act:
    if sum(a.C for a, _, _ in srcs) > CONV_CIN_MAX:  # (never at the default configuration: 128 channels at most)
        assert proj is None, "fused skip projection on a wide convolution"
        return _conv2d_wide(srcs, w_packed, bias, cout, taps=taps, stride=stride, upsample=upsample, residual=residual,
                            residual_norm=residual_norm, want_stats=want_stats, out_nchw=out_nchw, cout_padded=cout_padded, naive=naive,
                            fast_math=fast_math or w_f16 is not None, module=module)
    a0 = srcs[0][0]
    n, hs, ws, _ = a0.shape
    if upsample:
        h, w = hs * 2, ws * 2
    else:
        h, w = hs // stride, ws // stride
    dev = a0.t.device
    p = nv.ConvParams()
    p.N, p.H, p.W = n, h, w
    p.Cout = cout
    p.CoutPad = cout_padded or nv.cout_pad(cout)
    p.taps, p.stride, p.upsample, p.nsrc = taps, stride, int(upsample), len(srcs)
    for i, (a, prologue, norm) in enumerate(srcs):
        assert a.t.is_contiguous() and a.t.dtype == torch.float32 and tuple(a.shape[:3]) == (n, hs, ws)
        p.src[i].x = nv.ptr(a.t)
        p.src[i].C = a.C
        p.src[i].prologue = prologue
        if prologue != nv.PROLOGUE_NONE:
            p.src[i].norm = norm.to_native(a)
    # valid extent of the output from the sources' (all sources, and the residual, cover the same part)
    valid = None
    src_valid = {a.valid for a, _, _ in srcs}
    assert len(src_valid) == 1, f"sources with different valid extents: {src_valid}"
    sv = src_valid.pop()
    if sv is not None:
        assert TAPE is None, "valid extents are an inference-path feature (no recorded backward)"
        valid = (sv[0] * 2, sv[1] * 2) if upsample else ((sv[0] // stride, sv[1] // stride))
        assert stride == 1 or (sv[0] % 2 == 0 and sv[1] % 2 == 0), f"stride-2 conv over an odd valid extent {sv}"
        p.valid_h, p.valid_w = valid
        assert residual is None or residual.valid == valid
    p.w = nv.ptr(w_packed)
    if w_f16 is not None:
        p.w_f16 = nv.ptr(w_f16)
    if w_f16 is not None or fast_math:
        p.precision = nv.PRECISION_F16X2  # shapes the split kernel does not cover still get its cheaper prologue math
    p.bias = nv.ptr(bias)
    if residual is not None:
        assert tuple(residual.shape) == (n, h, w, cout) and residual.t.is_contiguous()
        p.residual = nv.ptr(residual.t)
        if residual_norm is not None:
            p.residual_norm = residual_norm.to_native(residual)
    if out_nchw:
        out = torch.empty(n, cout, h, w, device=dev, dtype=torch.float32)
    else:
        out = torch.empty(n, h, w, cout, device=dev, dtype=torch.float32)
    p.out = nv.ptr(out)
    p.out_nchw = int(out_nchw)
    if proj is not None:
        # `proj(cat(sources)) + conv(...)` in one launch (ResBlock.forward, blocks.py:147); dmd_conv2d fails loudly
        # on parameters dmd_conv2d_proj_eligible() rejects -- the caller asks proj_fusable() first
        p_srcs, p_w16, p_bias = proj[:3]  # (+ the projection's nn.Conv2d as a fourth element when the launch is recorded)
        assert residual is None and len(p_srcs) == 2 and (TAPE is None or len(proj) == 4)
        p.proj_nsrc = len(p_srcs)
        for i, a in enumerate(p_srcs):
            assert a.t.is_contiguous() and a.t.dtype == torch.float32 and tuple(a.shape[:3]) == (n, h, w)
            p.proj_x[i] = nv.ptr(a.t)
            p.proj_C[i] = a.C
        p.proj_w_f16 = nv.ptr(p_w16)
        p.proj_bias = nv.ptr(p_bias)
    stats, tiles = None, 0
    if want_stats:
        tiles = nv.conv_stat_tiles(h, w)
        stats = new_stats(n, cout, tiles, dev)
        p.out_stats = nv.ptr(stats)
    use_naive = _USE_NAIVE if naive is None else naive
    fn = nv.lib().dmd_conv2d_naive if use_naive else nv.lib().dmd_conv2d
    if nv.PROFILER is not None:
        cin = sum(a.C for a, _, _ in srcs)
        flops = 2.0 * n * h * w * cout * cin * taps  # algorithmic: MAC = 2, real channels
        nbytes = 4.0 * (sum(a.t.numel() for a, _, _ in srcs) + out.numel() + (residual.t.numel() if residual is not None else 0))
        if proj is not None:
            flops += 2.0 * n * h * w * cout * sum(a.C for a in proj[0])
            nbytes += 4.0 * sum(a.t.numel() for a in proj[0])
        nv.PROFILER.annotate(kernel_key(p), flops, nbytes)
    nv.check(fn(C.byref(p), nv.stream()), "dmd_conv2d")
    result = Act(out, stats, tiles, valid=valid)
    if TAPE is not None:
        assert module is not None, "recording a conv launch that does not name its nn.Conv2d"
        if proj is not None:
            # recorded as what it computes -- the projection as a launch of its own in front of this one, its result this launch's
            # residual.  The backward of a convolution needs its inputs and the gradient of its output, never the output: a
            # one-element tensor stands for the result that was not materialised (the gradient dictionary is keyed by storage).
            ghost = Act(torch.empty(1, device=dev, dtype=torch.float32))
            TAPE.append(ConvRecord([(a, nv.PROLOGUE_NONE, None) for a in proj[0]], proj[3], 1, 1, False, None, None, ghost, False))
            residual = ghost
        TAPE.append(ConvRecord(list(srcs), module, taps, stride, upsample, residual, residual_norm, result, out_nchw))
    return result


FUSE_PROJ = True  # skip projections inside conv2's launch (dmd_conv_f16ws.hip: PROJECTION); a module attribute for the tests


def proj_fusable(xs: Sequence[Act], cout: int, precision: str, naive: Optional[bool]) -> bool:
    """Mirror of dmd_conv2d_proj_eligible() for a ResBlock whose conv2 is cout -> cout: split-fp16 launch (inference, or a
    recorded training forward), cout == 64, H, W multiples of 16, two 64-channel projection sources."""
    if not FUSE_PROJ or precision != "f16x2" or naive or _USE_NAIVE:
        return False
    if TAPE is not None and os.environ.get("DIAMOND_TRAIN_FUSE_PROJ", "1") != "1":  # (A/B switch: recorded forwards unfused, as before)
        return False
    n, hh, ww, _ = xs[0].shape
    if any(a.valid is not None for a in xs):
        return False
    return (cout == 64 and hh % 16 == 0 and ww % 16 == 0 and len(xs) == 2 and all(a.C == 64 for a in xs)
            and n * hh * ww * 256 < 2 ** 32)


def gn_stats(t: Tensor, valid: Optional[Tuple[int, int]] = None) -> Act:
    """Attach (single-tile) GroupNorm statistics to an NHWC tensor no dmd kernel produced (valid: of that part of it)."""
    n, h, w, c = t.shape
    stats = new_stats(n, c, 1, t.device)
    if valid is None:
        nv.check(nv.lib().dmd_gn_stats(nv.fptr(t), nv.ptr(stats), n, h * w, c, nv.stream()), "dmd_gn_stats")
    else:
        nv.check(nv.lib().dmd_gn_stats_valid(nv.fptr(t), nv.ptr(stats), n, h, w, valid[0], valid[1], c, nv.stream()), "dmd_gn_stats_valid")
    return Act(t, stats, 1, valid=valid)


def padded_extent(h: int, w: int, num_down: int) -> Tuple[int, int]:
    """Buffer size for an (h, w) image through a U-Net with num_down stride-2 levels: (h, w) itself when every level is a
    multiple of the kernels' 8-pixel tiles, else the next multiple of 16 * 2**num_down (every level on the 16x16 tiles), in
    which the image is the VALID EXTENT."""
    m = 8 * 2 ** num_down
    if h % m == 0 and w % m == 0:
        return h, w
    m *= 2
    return (h + m - 1) // m * m, (w + m - 1) // m * m


def nchw_to_nhwc(x: Tensor, cpad: Optional[int] = None) -> Tensor:
    n, c, h, w = x.shape
    cp = cpad or c
    x = x.contiguous()
    out = torch.empty(n, h, w, cp, device=x.device, dtype=torch.float32)
    nv.check(nv.lib().dmd_nchw_to_nhwc(nv.fptr(x), nv.fptr(out), n, c, h, w, cp, nv.stream()), "dmd_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: Tensor, c: Optional[int] = None) -> Tensor:
    n, h, w, cp = x.shape
    c = c or cp
    out = torch.empty(n, c, h, w, device=x.device, dtype=torch.float32)
    nv.check(nv.lib().dmd_nhwc_to_nchw(nv.fptr(x), nv.fptr(out), n, c, h, w, cp, nv.stream()), "dmd_nhwc_to_nchw")
    return out


def attention(qkv: Act, c: int, head_dim: int = 8) -> Tensor:
    n, h, w, c3 = qkv.shape
    assert c3 == 3 * c
    out = torch.empty(n, h, w, c, device=qkv.t.device, dtype=torch.float32)
    if qkv.valid is not None:  # keys outside the valid extent stay out of the softmax
        assert TAPE is None
        nv.check(nv.lib().dmd_attention_valid(nv.fptr(qkv.t), nv.fptr(out), n, h, w, qkv.valid[0], qkv.valid[1], c, head_dim, nv.stream()),
                 "dmd_attention_valid")
        return out
    if nv.PROFILER is not None:  # QK^T and PV: 2 x (2 T^2 d) per head
        t = h * w
        nv.PROFILER.annotate("attention_f16x2_kernel" if t % 256 == 0 else "attention_kernel", 4.0 * n * t * t * c, 4.0 * n * t * 4 * c)
    nv.check(nv.lib().dmd_attention(nv.fptr(qkv.t), nv.fptr(out), n, h * w, c, head_dim, nv.stream()), "dmd_attention")
    if TAPE is not None:
        TAPE.append(AttnRecord(qkv, out, c, head_dim))
    return out


def linear(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, silu: bool = False, out: Optional[Tensor] = None,
           accumulate: bool = False) -> Tensor:
    if out is None:
        out = torch.empty(a.shape[0], w.shape[0], device=a.device, dtype=torch.float32)
    return nv.linear(a, w, bias, out, accumulate=accumulate, silu=silu)
