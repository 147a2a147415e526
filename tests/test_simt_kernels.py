"""The kernels' own source, run lane by lane on the CPU (tests/simt: a SIMT interpreter, TEST INFRASTRUCTURE), against numpy
restatements of what each entry point of include/diamond_hip.h promises.  This checks index arithmetic, LDS layouts,
synchronisation structure and MFMA operand layouts without a GPU; the `-m gpu` tests remain the parity tests proper (the
interpreter's MFMA accumulation order is not the hardware's, loads are synchronous and waits are no-ops here: nothing about
memory ordering, in-flight LDS-DMA or speed is established)."""
import os

import numpy as np
import pytest

from tests.simt import loader as S
from tests.simt.fence import fenced as G  # arrays between inaccessible pages: an out-of-bounds access of a kernel faults


def test_product_refuses_the_host_build(monkeypatch):
    """diamond_amd has no CPU path: pointing it at the interpreter build must fail loudly."""
    import importlib
    S.lib()
    import diamond_amd.native as nv
    monkeypatch.setattr(nv, "LIB_PATH", S.LIB_PATH)
    monkeypatch.setattr(nv, "_lib", None)
    with pytest.raises(nv.NativeLibraryMissing, match="no CPU path"):
        nv.lib()


# ---- schedule independence: a correctly synchronised kernel computes the SAME BITS in whatever order the interpreter resumes
#      its waves and lanes (SIMT_SCHEDULE 0: ascending, 1: descending, 2: odd waves first); a missing barrier or a read of LDS
#      that another wave is still writing shows up as a difference.  Results of schedule 0 are kept and compared by test id. ----
_BITS = {}
_COMPARED = [0]
SCHEDULES = [0, 1, 2]


def _same_bits_as_schedule_0(request, schedule, *arrays):
    key = request.node.name.replace(f"[{schedule}-", "[").replace(f"-{schedule}]", "]")
    blob = b"".join(np.ascontiguousarray(a).tobytes() for a in arrays if a is not None)
    if schedule == 0:
        _BITS[key] = blob
    elif key in _BITS:  # (absent when a single schedule was selected with -k)
        _COMPARED[0] += 1
        assert _BITS[key] == blob, f"{key}: results depend on the wave schedule ({schedule} vs 0)"



def _group_sums(x, hv=None, wv=None):
    n, h, w, c = x.shape
    v = x[:, :hv, :wv].astype(np.float64).reshape(n, -1, c // 32, 32)
    return np.stack([v.sum(axis=(1, 3)), (v * v).sum(axis=(1, 3))], axis=-1)  # (N, G, 2)


@pytest.mark.parametrize("shape", [(2, 8, 8, 64), (1, 5, 7, 32), (3, 16, 4, 96)])
def test_gn_stats(shape):
    rng = np.random.default_rng(0)
    x = rng.standard_normal(shape).astype(np.float32)
    n, h, w, c = shape
    stats = np.full((n, c // 32, 2), np.nan)
    S.check(S.lib().dmd_gn_stats(S.ptr(x), S.ptr(stats), n, h * w, c, None), "dmd_gn_stats")
    np.testing.assert_allclose(stats, _group_sums(x), rtol=1e-12, atol=1e-10)


def test_gn_stats_valid_extent():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 16, 16, 64)).astype(np.float32)
    stats = np.full((2, 2, 2), np.nan)
    S.check(S.lib().dmd_gn_stats_valid(S.ptr(x), S.ptr(stats), 2, 16, 16, 9, 13, 64, None), "dmd_gn_stats_valid")
    np.testing.assert_allclose(stats, _group_sums(x, 9, 13), rtol=1e-12, atol=1e-10)
    assert S.lib().dmd_gn_stats_valid(S.ptr(x), S.ptr(stats), 2, 16, 16, 17, 13, 64, None) != 0  # extent beyond the buffer


# ---- dmd_conv2d: conv_mfma_kernel instances and the streaming 1x1 (no w_f16: nothing here is eligible for conv_f16ws) ----------------
from diamond_amd import native as nv  # struct layouts only


def _norm(stats=None, tiles=0, mul=None, add=None, plus_one=False):
    n = nv.Norm()
    n.stats, n.stat_tiles, n.mul_plus_one = S.ptr(stats), tiles, int(plus_one)
    n.mul, n.add = S.ptr(mul), S.ptr(add)
    n.mul_stride = 0 if mul is None else mul.shape[-1]
    n.add_stride = 0 if add is None else add.shape[-1]
    return n


def _apply_norm(x, hv, wv, mul, add, plus_one, silu):
    """GroupNorm(32-channel groups, eps 1e-5, statistics of the valid extent) * (1 +) mul + add [, SiLU], fp64."""
    n, h, w, c = x.shape
    v = x[:, :hv, :wv].astype(np.float64).reshape(n, hv * wv, c // 32, 32)
    mean = v.mean(axis=(1, 3))
    var = (v * v).mean(axis=(1, 3)) - mean * mean
    rstd = 1.0 / np.sqrt(np.maximum(var, 0) + 1e-5)
    m = np.ones((n, c)) if mul is None else mul.astype(np.float64)
    if plus_one:
        m = 1.0 + m
    a = np.repeat(rstd, 32, axis=1) * m
    y = (x.astype(np.float64) - np.repeat(mean, 32, axis=1)[:, None, None]) * a[:, None, None]
    if add is not None:
        y = y + add.astype(np.float64)[:, None, None]
    return y / (1.0 + np.exp(-y)) if silu else y


def _partial_stats(x, hv, wv, tiles, rng):
    """(N, G, T, 2) partial sums whose total is the valid extent's; split unevenly over T tiles on purpose."""
    tot = _group_sums(x, hv, wv)  # (N, G, 2)
    wgt = rng.random((1, 1, tiles, 1)) + 0.1
    wgt = wgt / wgt.sum()
    return np.ascontiguousarray(tot[:, :, None, :] * wgt)


def _ref_conv(xs, w, bias, k, stride, upsample, hv_out, wv_out):
    """fp64 F.conv2d(cat(xs), w, bias, stride, padding=k//2) on NHWC arrays already in the conv's input domain; input
    positions beyond the sources' valid extent are zero."""
    x = np.concatenate(xs, axis=-1)
    if upsample:
        x = x.repeat(2, axis=1).repeat(2, axis=2)
    n, h, w_, c = x.shape
    hv_in, wv_in = (hv_out * stride, wv_out * stride)
    x = x.copy()
    x[:, hv_in:] = 0
    x[:, :, wv_in:] = 0
    pad = k // 2
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    ho, wo = h // stride, w_ // stride
    out = np.zeros((n, ho, wo, w.shape[0]))
    for dy in range(k):
        for dx in range(k):
            patch = xp[:, dy:dy + h:stride, dx:dx + w_:stride][:, :ho, :wo]
            out += patch @ w[:, :, dy, dx].astype(np.float64).T
    return out + (0 if bias is None else bias.astype(np.float64))


CONV_CASES = [
    # N, H, W (output), [Cin...], Cout, k, stride, upsample, prologues, residual, precision, valid
    dict(n=2, h=8, w=16, cin=[32], cout=64, k=3),
    dict(n=1, h=16, w=8, cin=[64], cout=32, k=3, prologue=[1], film=True),                       # W % 16 != 0 geometry
    dict(n=1, h=8, w=16, cin=[32, 64], cout=64, k=3, prologue=[1, 1], film=True, residual="raw", stats=True),
    dict(n=2, h=8, w=8, cin=[32], cout=16, k=3, stride=2),
    dict(n=1, h=16, w=16, cin=[64], cout=64, k=3, upsample=1, prologue=[1]),
    dict(n=1, h=8, w=16, cin=[64], cout=96, k=1, prologue=[2], residual="norm"),                  # attention-style 1x1s
    dict(n=1, h=8, w=16, cin=[32], cout=3, k=3, prologue=[1], nchw=True),                         # few-channel NCHW head
    dict(n=1, h=16, w=16, cin=[32], cout=64, k=3, prologue=[1], valid=(9, 13), stats=True),       # valid extent
    dict(n=1, h=16, w=16, cin=[64], cout=64, k=3, stride=2, valid=(5, 7), stats=True),
    dict(n=1, h=8, w=16, cin=[32, 32], cout=64, k=3, prologue=[1, 0], precision=1, stats=True),   # split-fp16 conv_mfma instance
    dict(n=1, h=16, w=8, cin=[64], cout=32, k=3, stride=2, precision=1, residual="raw"),
    dict(n=1, h=8, w=16, cin=[128], cout=64, k=1),                                                # streaming 1x1 kernel
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k not in ("n",)).replace(" ", ""))
def test_conv2d(case):
    rng = np.random.default_rng(7)
    L = S.lib()
    n, h, w, cins, cout, k = case["n"], case["h"], case["w"], case["cin"], case["cout"], case["k"]
    stride, up = case.get("stride", 1), case.get("upsample", 0)
    prol = case.get("prologue", [0] * len(cins))
    hv, wv = case.get("valid", (h, w))
    hs, ws = (h * stride, w * stride) if not up else (h // 2, w // 2)         # source buffer size
    hvs, wvs = (hv * stride, wv * stride) if not up else (hv // 2, wv // 2)   # source valid extent
    cout_pad = (cout + 15) // 16 * 16
    cin = sum(cins)

    p = nv.ConvParams()
    p.N, p.H, p.W, p.Cout, p.CoutPad, p.taps, p.stride, p.upsample, p.nsrc = n, h, w, cout, cout_pad, k * k, stride, up, len(cins)
    p.precision = case.get("precision", 0)
    if "valid" in case:
        p.valid_h, p.valid_w = hv, wv
    keep, xs_ref = [], []
    for i, c in enumerate(cins):
        x = G((rng.standard_normal((n, hs, ws, c)) * 1.5 + 0.3).astype(np.float32), tight="start" if (n + hs + c) % 2 else "end")
        p.src[i].x, p.src[i].C, p.src[i].prologue = S.ptr(x), c, prol[i]
        if prol[i]:
            tiles = 3
            st = _partial_stats(x, hvs, wvs, tiles, rng)
            mul = (rng.standard_normal((n, c)) * 0.3).astype(np.float32) if case.get("film") or prol[i] == 2 else None
            add = (rng.standard_normal((n, c)) * 0.3).astype(np.float32) if case.get("film") or prol[i] == 2 else None
            plus_one = bool(case.get("film"))
            p.src[i].norm = _norm(st, tiles, mul, add, plus_one)
            xs_ref.append(_apply_norm(x, hvs, wvs, mul, add, plus_one, silu=prol[i] == 1))
            keep += [st, mul, add]
        else:
            xs_ref.append(x.astype(np.float64))
        keep.append(x)

    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    bias = rng.standard_normal(cout_pad).astype(np.float32)
    bias[cout:] = 0
    packed = np.full((cin // 16) * k * k * cout_pad * 16, np.nan, dtype=np.float32)
    S.check(L.dmd_pack_conv_weight(S.ptr(wt), S.ptr(packed), cout, cin, k, cout_pad, cin, None), "dmd_pack_conv_weight")
    assert np.isfinite(packed).all()
    p.w, p.bias = S.ptr(packed), S.ptr(bias)

    ref = _ref_conv(xs_ref, wt, bias[:cout], k, stride, up, hv, wv)
    if case.get("residual"):
        r = G(rng.standard_normal((n, h, w, cout)).astype(np.float32))
        p.residual = S.ptr(r)
        if case["residual"] == "norm":
            rst = _partial_stats(r, hv, wv, 2, rng)
            g, b = rng.standard_normal((1, cout)).astype(np.float32), rng.standard_normal((1, cout)).astype(np.float32)
            g, b = np.ascontiguousarray(np.repeat(g, n, 0)), np.ascontiguousarray(np.repeat(b, n, 0))
            p.residual_norm = _norm(rst, 2, g, b)
            ref = ref + _apply_norm(r, hv, wv, g, b, False, silu=False)
            keep += [rst, g, b]
        else:
            ref = ref + r
        keep.append(r)

    nchw = bool(case.get("nchw"))
    out = G(np.full((n, cout, h, w) if nchw else (n, h, w, cout), np.nan, dtype=np.float32))
    p.out, p.out_nchw = S.ptr(out), int(nchw)
    tiles = L.dmd_conv_stat_tiles(h, w)
    stats = G(np.full((n, cout // 32, tiles, 2), np.nan)) if case.get("stats") else None
    p.out_stats = S.ptr(stats)

    name = (b" " * 160)
    S.check(L.dmd_conv2d(p, None), "dmd_conv2d")
    got = out.transpose(0, 2, 3, 1) if nchw else out
    # the exact instance is an fp32 fma chain, the split instance carries fp32-class error (2^-22 per operand)
    err = np.abs(got[:, :hv, :wv] - ref[:, :hv, :wv]).max()
    assert err <= 2e-5 * max(1.0, np.abs(ref).max()), err
    if stats is not None:
        want = _group_sums(np.ascontiguousarray(got.astype(np.float32)), hv, wv)
        np.testing.assert_allclose(stats.sum(axis=2), want, rtol=1e-9, atol=1e-6)


# ---- dmd_conv2d_wgrad ------------------------------------------------------------------------------------------------------------
WGRAD_CASES = [
    dict(n=2, h=8, w=16, cin=64, cout=64, k=3),
    dict(n=3, h=8, w=8, cin=32, cout=64, k=3, prologue=1, film=True),          # odd number of 8x8 sub-blocks: a half-empty tile
    dict(n=1, h=16, w=16, cin=64, cout=64, k=3, prologue=1, precision=1),
    dict(n=2, h=8, w=8, cin=16, cin_real=15, cout=64, k=3),                     # denoiser conv_in: 15 real input channels
    dict(n=1, h=8, w=16, cin=64, cout=16, k=3, prologue=1, precision=1),        # denoiser conv_out (dy padded to 16)
    dict(n=2, h=8, w=8, cin=32, cout=64, k=1, prologue=2),
    dict(n=1, h=16, w=8, cin=64, cout=64, k=1, precision=1),
    dict(n=2, h=8, w=8, cin=32, cout=32, k=3, no_bias=True),
    dict(n=3, h=16, w=16, cin=64, cout=64, k=3, prologue=1, film=True, precision=1),  # 6 tiles, three images: table refreshes
    dict(n=5, h=8, w=8, cin=32, cout=64, k=3, prologue=1, precision=1),                # 2.5 tiles: a tile spanning two images
    dict(n=3, h=8, w=16, cin=16, cin_real=15, cout=64, k=3, precision=1),              # conv_in on the producer / consumer kernel: 240 patch items for 256 producer threads
    dict(n=2, h=16, w=8, cin=32, cout=32, k=1, prologue=2, film=True, precision=1),    # 1x1, normalised without SiLU
]


# the launcher's plan: at most DIAMOND_WGRAD_MAX_WG workgroups (default 256; 512 for the 32-output-channel 3x3 shapes), each walking a contiguous range of tiles with its
# accumulators in registers and the next tile's loads in flight under the current tile's MFMAs
WGRAD_PLANS = [dict(), dict(max_wg=1), dict(max_wg=3), dict(max_wg=1024)]


@pytest.mark.parametrize("schedule", SCHEDULES)
@pytest.mark.parametrize("plan", WGRAD_PLANS, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()) or "default")
@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_conv2d_wgrad(case, plan, schedule, monkeypatch, request, dmd_env):
    monkeypatch.setenv("SIMT_SCHEDULE", str(schedule))
    dmd_env(DIAMOND_WGRAD_MAX_WG=plan.get("max_wg"))
    rng = np.random.default_rng(11)
    L = S.lib()
    n, h, w, cin, cout, k = case["n"], case["h"], case["w"], case["cin"], case["cout"], case["k"]
    cin_real, prol = case.get("cin_real", cin), case.get("prologue", 0)
    x = (rng.standard_normal((n, h, w, cin)) * 1.2 - 0.2).astype(np.float32)
    x[..., cin_real:] = 0
    x = G(x, tight="start" if (n + h) % 2 else "end")
    dy = G(rng.standard_normal((n, h, w, cout)).astype(np.float32))
    p = nv.WgradParams()
    p.N, p.H, p.W, p.Cout, p.taps, p.cin_real, p.precision = n, h, w, cout, k * k, cin_real, case.get("precision", 0)
    p.src.x, p.src.C, p.src.prologue = S.ptr(x), cin, prol
    a = x.astype(np.float64)
    keep = []
    if prol:
        st = _partial_stats(x, h, w, 2, rng)
        film = case.get("film") or prol == 2
        mul = (rng.standard_normal((n, cin)) * 0.3).astype(np.float32) if film else None
        add = (rng.standard_normal((n, cin)) * 0.3).astype(np.float32) if film else None
        p.src.norm = _norm(st, 2, mul, add, bool(case.get("film")))
        a = _apply_norm(x, h, w, mul, add, bool(case.get("film")), silu=prol == 1)
        keep += [st, mul, add]
    p.dy = S.ptr(dy)
    ws = G(np.full(L.dmd_wgrad_workspace_floats(p), np.nan, dtype=np.float32))  # sized by the library: an overrun faults
    dw = G(np.full((cout, cin_real, k, k), np.nan, dtype=np.float32))
    db = None if case.get("no_bias") else np.full(cout, np.nan, dtype=np.float32)
    p.workspace, p.dw, p.dbias = S.ptr(ws), S.ptr(dw), S.ptr(db)
    S.check(L.dmd_conv2d_wgrad(p, None), "dmd_conv2d_wgrad")
    _same_bits_as_schedule_0(request, schedule, dw, db)

    pad = k // 2
    ap = np.pad(a, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    want = np.zeros((cout, cin_real, k, k))
    g = dy.astype(np.float64).reshape(-1, cout)
    for ky in range(k):
        for kx in range(k):
            want[:, :, ky, kx] = g.T @ ap[:, ky:ky + h, kx:kx + w, :cin_real].reshape(-1, cin_real)
    scale = np.abs(want).max()
    assert np.abs(dw - want).max() <= 3e-6 * scale * np.sqrt(n * h * w / 64), np.abs(dw - want).max() / scale
    if db is not None:
        np.testing.assert_allclose(db, g.sum(axis=0), rtol=0, atol=2e-5 * np.abs(g.sum(axis=0)).max() + 1e-5)


@pytest.mark.parametrize("max_wg", [None, 2])
@pytest.mark.parametrize("shape", [(3, 16, 16, 64, 64, 3), (5, 8, 8, 32, 32, 3), (2, 8, 16, 32, 64, 1)], ids=str)
def test_conv2d_wgrad_two_roles_same_bits(shape, max_wg, dmd_env):
    """wgrad_ps_kernel (producer / consumer waves, the default of the split-fp16 gradient) against wgrad_kernel<G, true>
    (DIAMOND_WGRAD_PS=0) on a source without prologue: the same plan, the same pixels at the same k of every MFMA, the same order of
    accumulation -- the same bits of the weight gradient."""
    rng = np.random.default_rng(21)
    L = S.lib()
    n, h, w, cin, cout, k = shape
    x = G((rng.standard_normal((n, h, w, cin)) * 1.2 - 0.2).astype(np.float32))
    dy = G(rng.standard_normal((n, h, w, cout)).astype(np.float32))
    got = []
    for ps in (1, 0):
        dmd_env(DIAMOND_WGRAD_PS=ps, DIAMOND_WGRAD_MAX_WG=max_wg)
        p = nv.WgradParams()
        p.N, p.H, p.W, p.Cout, p.taps, p.cin_real, p.precision = n, h, w, cout, k * k, cin, 1
        p.src.x, p.src.C, p.src.prologue = S.ptr(x), cin, 0
        p.dy = S.ptr(dy)
        ws = G(np.full(L.dmd_wgrad_workspace_floats(p), np.nan, dtype=np.float32))
        dw = G(np.full((cout, cin, k, k), np.nan, dtype=np.float32))
        db = np.full(cout, np.nan, dtype=np.float32)
        p.workspace, p.dw, p.dbias = S.ptr(ws), S.ptr(dw), S.ptr(db)
        S.check(L.dmd_conv2d_wgrad(p, None), "dmd_conv2d_wgrad")
        got.append((np.array(dw), db.copy()))
    # (the bias gradient is summed by the producer threads over pixel PAIRS: another order of fp32 additions)
    assert np.isfinite(got[0][0]).all() and np.array_equal(got[0][0], got[1][0])
    np.testing.assert_allclose(got[0][1], got[1][1], rtol=0, atol=2e-5 * np.abs(got[1][1]).max())


def test_conv2d_wgrad_many_partials(dmd_env):
    """more partials than the single-pass reduction takes (64): the two-pass reduction, against the single pass of a small plan"""
    rng = np.random.default_rng(12)
    L = S.lib()
    n, h, w, c = 20, 16, 16, 32  # 80 8x8 blocks = 40 tiles... times 4 images more: 100 workgroups
    n = 50
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    dy = rng.standard_normal((n, h, w, c)).astype(np.float32)
    g = dy.astype(np.float64).reshape(-1, c)
    ap = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    want = np.zeros((c, c, 3, 3))
    for ky in range(3):
        for kx in range(3):
            want[:, :, ky, kx] = g.T @ ap[:, ky:ky + h, kx:kx + w].reshape(-1, c)
    got = {}
    for single in ("0", "256"):
        dmd_env(DIAMOND_WGRAD_MAX_WG=1024 if single == "0" else 32)  # 200 tiles: 200 partials (two passes) / 32 (one pass)
        p = nv.WgradParams()
        p.N, p.H, p.W, p.Cout, p.taps, p.cin_real, p.precision = n, h, w, c, 9, c, 1
        p.src.x, p.src.C, p.src.prologue, p.dy = S.ptr(x), c, 0, S.ptr(dy)
        ws = G(np.full(L.dmd_wgrad_workspace_floats(p), np.nan, dtype=np.float32))
        dw, db = np.full((c, c, 3, 3), np.nan, dtype=np.float32), np.full(c, np.nan, dtype=np.float32)
        p.workspace, p.dw, p.dbias = S.ptr(ws), S.ptr(dw), S.ptr(db)
        S.check(L.dmd_conv2d_wgrad(p, None), "dmd_conv2d_wgrad")
        assert np.abs(dw - want).max() <= 2e-6 * np.abs(want).max() and np.abs(db - g.sum(0)).max() <= 2e-5 * np.abs(g.sum(0)).max()
        got[single] = dw
    assert np.abs(got["0"] - got["256"]).max() <= 1e-6 * np.abs(want).max()


@pytest.mark.parametrize("max_wg", [32, 1024], ids=["single_pass", "two_passes"])
def test_wgrad_deferred_reductions_are_bit_identical(max_wg, dmd_env):
    """dmd_conv2d_wgrad(defer_reduce = 1) + dmd_wgrad_reduce_jobs (ABI v10) against the undeferred call: the same bits, for few
    partials (one fp64 pass) and many (fp32 slices, then fp64), with the gradients of two sources landing in the channel slices of
    ONE OIHW tensor, a 64-row piece of a wider output, 15 real input channels, a missing bias -- and 33 jobs, so that the table
    takes two launches."""
    rng = np.random.default_rng(14)
    L = S.lib()
    dmd_env(DIAMOND_WGRAD_MAX_WG=max_wg)
    n, h, w = 34, 16, 16  # 136 8x8 blocks = 68 tiles: 68 partials at max_wg = 1024 (two passes), 32 at 32 (one)
    keep, jobs, checks = [], [], []

    def one(cin, cin_real, cout, k, dw_all, row0, c0, db_all):
        x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
        x[..., cin_real:] = 0
        dy = rng.standard_normal((n, h, w, cout)).astype(np.float32)
        p = nv.WgradParams()
        p.N, p.H, p.W, p.Cout, p.taps, p.cin_real, p.precision = n, h, w, cout, k * k, cin_real, 1
        p.src.x, p.src.C, p.src.prologue, p.dy = S.ptr(x), cin, 0, S.ptr(dy)
        ws = np.full(L.dmd_wgrad_workspace_floats(p), np.nan, dtype=np.float32)
        dw, db = np.full((cout, cin_real, k, k), np.nan, dtype=np.float32), np.full(cout, np.nan, dtype=np.float32)
        p.workspace, p.dw, p.dbias = S.ptr(ws), S.ptr(dw), S.ptr(db)
        S.check(L.dmd_conv2d_wgrad(p, None), "dmd_conv2d_wgrad")  # the undeferred gradient
        view = dw_all[row0:row0 + cout]
        p.workspace, p.dw, p.dbias, p.defer_reduce = None, view.ctypes.data, None if db_all is None else db_all[row0:].ctypes.data, 1
        job = nv.WgradReduceJob()
        S.check(L.dmd_wgrad_job(p, job), "dmd_wgrad_job")
        assert (job.num_wg > 64) == (max_wg == 1024) and job.ld_cin == cin_real and job.c0 == 0
        # the partials alone (less than dmd_wgrad_workspace_floats), against a guard page: a write past them faults
        ws2 = G(np.full(job.num_wg * (job.NB * job.NCO * 256 + job.NCO * 16), np.nan, dtype=np.float32))
        assert ws2.size < ws.size
        p.workspace = job.partials = S.ptr(ws2)
        job.ld_cin, job.c0 = dw_all.shape[1], c0
        S.check(L.dmd_conv2d_wgrad(p, None), "dmd_conv2d_wgrad")
        keep.extend([x, dy, ws2])
        jobs.append(job)
        checks.append((dw, db, view, c0, cin_real, None if db_all is None else db_all[row0:row0 + cout]))

    two = np.full((64, 96, 3, 3), np.nan, dtype=np.float32)  # a convolution over cat(64, 32) channels
    db_two = np.full(64, np.nan, dtype=np.float32)
    one(64, 64, 64, 3, two, 0, 0, db_two)
    one(32, 32, 64, 3, two, 0, 64, None)
    wide = np.full((128, 32, 1, 1), np.nan, dtype=np.float32)  # two 64-row pieces of a 1x1 convolution with 128 outputs
    db_wide = np.full(128, np.nan, dtype=np.float32)
    one(32, 32, 64, 1, wide, 0, 0, db_wide)
    one(32, 32, 64, 1, wide, 64, 0, db_wide)
    head = np.full((16, 64, 3, 3), np.nan, dtype=np.float32)
    one(64, 64, 16, 3, head, 0, 0, np.full(16, np.nan, dtype=np.float32))
    first = np.full((64, 15, 3, 3), np.nan, dtype=np.float32)
    one(16, 15, 64, 3, first, 0, 0, None)
    small = [np.full((32, 32, 3, 3), np.nan, dtype=np.float32) for _ in range(27)]
    for t in small:
        one(32, 32, 32, 3, t, 0, 0, np.full(32, np.nan, dtype=np.float32))
    assert len(jobs) == 33
    table = (nv.WgradReduceJob * len(jobs))(*jobs)
    S.check(L.dmd_wgrad_reduce_jobs(table, len(jobs), None), "dmd_wgrad_reduce_jobs")
    for dw, db, view, c0, cin_real, dbv in checks:
        assert np.isfinite(dw).all() and np.array_equal(view[:, c0:c0 + cin_real], dw)
        if dbv is not None:
            assert np.array_equal(dbv, db)
    assert np.isfinite(two).all() and np.isfinite(wide).all() and np.isfinite(db_wide).all()
    bad = nv.WgradReduceJob.from_buffer_copy(bytes(jobs[1]))
    bad.c0 = 80  # channels [80, 112) of a 96-channel row
    assert L.dmd_wgrad_reduce_jobs((nv.WgradReduceJob * 1)(bad), 1, None) != 0


# ---- attention -------------------------------------------------------------------------------------------------------------------
def _ref_attention(qkv, c, mask=None):
    n, t, _ = qkv.shape
    q, k, v = (qkv[..., i * c:(i + 1) * c].astype(np.float64).reshape(n, t, c // 8, 8).transpose(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(0, 1, 3, 2) / np.sqrt(8.0)
    if mask is not None:
        s = np.where(mask[None, None, None, :], s, -np.inf)
    p = np.exp(s - s.max(axis=-1, keepdims=True))
    p /= p.sum(axis=-1, keepdims=True)
    return (p @ v).transpose(0, 2, 1, 3).reshape(n, t, c), p, (q, k, v)


@pytest.mark.parametrize("t", [64, 128, 256, 512])  # 64 / 128: attention_kernel; 256 / 512: the split-fp16 two-pass kernel
def test_attention(t):
    rng = np.random.default_rng(3)
    n, c = 2, 16
    qkv = (rng.standard_normal((n, t, 3 * c)) * 1.5).astype(np.float32)
    out = np.full((n, t, c), np.nan, dtype=np.float32)
    S.check(S.lib().dmd_attention(S.ptr(qkv), S.ptr(out), n, t, c, 8, None), "dmd_attention")
    ref, _, _ = _ref_attention(qkv, c)
    assert np.abs(out - ref).max() <= 5e-6 * np.abs(ref).max(), np.abs(out - ref).max()


def test_attention_valid_extent():
    rng = np.random.default_rng(4)
    n, h, w, c, hv, wv = 1, 8, 16, 8, 5, 11
    qkv = (rng.standard_normal((n, h * w, 3 * c)) * 1.5).astype(np.float32)
    out = np.full((n, h * w, c), np.nan, dtype=np.float32)
    S.check(S.lib().dmd_attention_valid(S.ptr(qkv), S.ptr(out), n, h, w, hv, wv, c, 8, None), "dmd_attention_valid")
    yy, xx = np.divmod(np.arange(h * w), w)
    mask = (yy < hv) & (xx < wv)
    ref, _, _ = _ref_attention(qkv, c, mask)
    assert np.abs(out[:, mask] - ref[:, mask]).max() <= 5e-6 * np.abs(ref).max()


def test_attention_bwd():
    rng = np.random.default_rng(5)
    n, t, c = 2, 64, 16
    qkv = rng.standard_normal((n, t, 3 * c)).astype(np.float32)
    dy = rng.standard_normal((n, t, c)).astype(np.float32)
    y64, p, (q, k, v) = _ref_attention(qkv, c)
    y = y64.astype(np.float32)
    L = S.lib()
    ws = G(np.full(L.dmd_attention_bwd_workspace_floats(n, t, c), np.nan, dtype=np.float32))
    dqkv = G(np.full_like(qkv, np.nan))
    S.check(L.dmd_attention_bwd(S.ptr(qkv), S.ptr(y), S.ptr(dy), S.ptr(dqkv), S.ptr(ws), n, t, c, 8, None), "dmd_attention_bwd")
    g = dy.astype(np.float64).reshape(n, t, c // 8, 8).transpose(0, 2, 1, 3)
    dp = g @ v.transpose(0, 1, 3, 2)
    ds = p * (dp - (dp * p).sum(axis=-1, keepdims=True))
    dq, dk, dv = ds @ k / np.sqrt(8.0), ds.transpose(0, 1, 3, 2) @ q / np.sqrt(8.0), p.transpose(0, 1, 3, 2) @ g
    want = np.concatenate([a.transpose(0, 2, 1, 3).reshape(n, t, c) for a in (dq, dk, dv)], axis=-1)
    assert np.abs(dqkv - want).max() <= 1e-5 * np.abs(want).max()


# ---- dmd_linear ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,k,acc,silu", [(5, 70, 48, 0, 0), (64, 64, 256, 1, 0), (3, 33, 512, 0, 1), (37, 96, 1024, 1, 0)])
def test_linear(m, n, k, acc, silu):
    rng = np.random.default_rng(m + n)
    a = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    c0 = rng.standard_normal((m, n)).astype(np.float32)
    c = c0.copy()
    p = nv.LinearParams()
    p.M, p.N, p.K, p.A, p.lda, p.W, p.ldw, p.bias, p.C, p.ldc = m, n, k, S.ptr(a), k, S.ptr(w), k, S.ptr(b), S.ptr(c), n
    p.accumulate, p.silu = acc, silu
    S.check(S.lib().dmd_linear(p, None), "dmd_linear")
    ref = a.astype(np.float64) @ w.astype(np.float64).T + b + (c0 if acc else 0)
    if silu:
        ref = ref / (1 + np.exp(-ref))
    assert np.abs(c - ref).max() <= 3e-6 * max(1.0, np.abs(ref).max())


# ---- dmd_gn_silu_bwd: fp64 finite-difference-free truth from the closed form in the header --------------------------------------
@pytest.mark.parametrize("n,hw,c,identity,skip", [(2, 64, 64, 0, True), (1, 300, 32, 0, False), (2, 256, 128, 1, True), (1, 64, 16, 0, False),
                                                  (3, 1024, 64, 0, True)])
def test_gn_silu_bwd(n, hw, c, identity, skip, dmd_env):
    """hw <= 256: the one-launch form (a workgroup holds an image) -- and, bitwise, the two launches it replaces"""
    rng = np.random.default_rng(hw + c)
    L = S.lib()
    x = (rng.standard_normal((n, hw, 1, c)) * 1.4 + 0.3).astype(np.float32)
    da = rng.standard_normal((n, hw, 1, c)).astype(np.float32)
    dskip = rng.standard_normal((n, hw, 1, c)).astype(np.float32) if skip else None
    mul = (rng.standard_normal((n, c)) * 0.3).astype(np.float32)
    add = (rng.standard_normal((n, c)) * 0.3).astype(np.float32)
    gsz = min(c, 32)
    g = c // gsz
    v = x.astype(np.float64).reshape(n, hw, g, gsz)
    tot = np.stack([v.sum(axis=(1, 3)), (v * v).sum(axis=(1, 3))], axis=-1)
    st = np.ascontiguousarray(np.stack([tot * 0.25, tot * 0.75], axis=2))  # (N, G, 2 tiles, 2)
    p = nv.GnBwdParams()
    p.N, p.HW, p.C, p.identity_activation = n, hw, c, identity
    p.x, p.norm, p.da, p.dskip = S.ptr(x), _norm(st, 2, mul, add, plus_one=True), S.ptr(da), S.ptr(dskip)
    dx = G(np.full_like(x, np.nan))
    dmul, dadd = G(np.full((n, c), np.nan, dtype=np.float32)), G(np.full((n, c), np.nan, dtype=np.float32))
    ws = G(np.zeros(L.dmd_gn_bwd_workspace_bytes(n, hw, c), dtype=np.uint8))
    p.dx, p.workspace, p.dmul, p.dadd = S.ptr(dx), S.ptr(ws), S.ptr(dmul), S.ptr(dadd)
    S.check(L.dmd_gn_silu_bwd(p, None), "dmd_gn_silu_bwd")

    mean = v.mean(axis=(1, 3), keepdims=True)
    rstd = 1 / np.sqrt((v * v).mean(axis=(1, 3), keepdims=True) - mean ** 2 + 1e-5)
    xh = ((v - mean) * rstd).reshape(n, hw, c)
    m1 = 1.0 + mul.astype(np.float64)[:, None]
    u = xh * m1 + add.astype(np.float64)[:, None]
    sg = 1 / (1 + np.exp(-u))
    du = da.astype(np.float64).reshape(n, hw, c) * (1 if identity else sg * (1 + u * (1 - sg)))
    dxh = (du * m1).reshape(n, hw, g, gsz)
    xhg = xh.reshape(n, hw, g, gsz)
    want = rstd * (dxh - dxh.mean(axis=(1, 3), keepdims=True) - xhg * (dxh * xhg).mean(axis=(1, 3), keepdims=True))
    want = want.reshape(n, hw, 1, c) + (dskip if skip else 0)
    assert np.abs(dx - want).max() <= 1e-5 * np.abs(want).max()
    np.testing.assert_allclose(dmul, (du * xh).sum(axis=1), rtol=0, atol=2e-5 * np.abs(du * xh).sum(axis=1).max())
    np.testing.assert_allclose(dadd, du.sum(axis=1), rtol=0, atol=2e-5 * np.abs(du).sum(axis=1).max())
    if hw <= 256:
        # the other two forms of the same sums: the single launch for many images (2; the default took the register-resident one
        # for few) and the two launches (0) -- the same bits
        for form in (2, 3, 0):
            dmd_env(DIAMOND_GN_BWD_FUSED=form)
            dx2 = G(np.full_like(x, np.nan))
            dmul2, dadd2 = G(np.full((n, c), np.nan, dtype=np.float32)), G(np.full((n, c), np.nan, dtype=np.float32))
            p.dx, p.dmul, p.dadd = S.ptr(dx2), S.ptr(dmul2), S.ptr(dadd2)
            assert not ws.any()  # (a single launch leaves the workspace untouched)
            S.check(L.dmd_gn_silu_bwd(p, None), "dmd_gn_silu_bwd")
            assert np.array_equal(dx, dx2) and np.array_equal(dmul, dmul2) and np.array_equal(dadd, dadd2), form
        assert ws.any()  # (the two launches went through the workspace)
    else:
        assert ws.any()


def test_gn_silu_bwd_one_launch_on_a_valid_extent_is_bitwise_the_two_launches(dmd_env):
    """the 9x9 valid part of a 16x16 buffer (the actor-critic encoder's last level at 72x72 frames): one launch against two"""
    rng = np.random.default_rng(3)
    L = S.lib()
    n, h, w, c, vh, vw = 3, 16, 16, 64, 9, 9
    x = (rng.standard_normal((n, h * w, 1, c)) * 1.4 + 0.3).astype(np.float32)
    da = rng.standard_normal((n, h * w, 1, c)).astype(np.float32)
    dskip = rng.standard_normal((n, h * w, 1, c)).astype(np.float32)
    mul, add = (rng.standard_normal((n, c)) * 0.3).astype(np.float32), (rng.standard_normal((n, c)) * 0.3).astype(np.float32)
    v = x.astype(np.float64).reshape(n, h, w, 2, 32)[:, :vh, :vw]
    st = np.ascontiguousarray(np.stack([v.sum(axis=(1, 2, 4)), (v * v).sum(axis=(1, 2, 4))], axis=-1)[:, :, None, :])  # (N, G, 1 tile, 2)
    got = {}
    for fused in (1, 2, 0):
        dmd_env(DIAMOND_GN_BWD_FUSED=fused)
        p = nv.GnBwdParams()
        p.N, p.HW, p.C, p.identity_activation, p.W, p.valid_h, p.valid_w = n, h * w, c, 0, w, vh, vw
        p.x, p.norm, p.da, p.dskip = S.ptr(x), _norm(st, 1, mul, add, plus_one=True), S.ptr(da), S.ptr(dskip)
        dx = G(np.full_like(x, np.nan))
        dmul, dadd = G(np.full((n, c), np.nan, dtype=np.float32)), G(np.full((n, c), np.nan, dtype=np.float32))
        ws = G(np.zeros(L.dmd_gn_bwd_workspace_bytes(n, h * w, c), dtype=np.uint8))
        p.dx, p.workspace, p.dmul, p.dadd = S.ptr(dx), S.ptr(ws), S.ptr(dmul), S.ptr(dadd)
        S.check(L.dmd_gn_silu_bwd(p, None), "dmd_gn_silu_bwd")
        assert bool(ws.any()) == (fused == 0)
        got[fused] = (dx.copy(), dmul.copy(), dadd.copy())
    assert all(np.array_equal(a_, b_) for a_, b_ in zip(got[0], got[1])) and all(np.array_equal(a_, b_) for a_, b_ in zip(got[0], got[2]))
    dxv = got[1][0].reshape(n, h, w, c)
    assert np.isfinite(dxv).all() and not dxv[:, vh:].any() and not dxv[:, :, vw:].any() and dxv[:, :vh, :vw].any()


# ---- dmd_pack_jobs: every packed layout against its definition -------------------------------------------------------------------
def _packed_f32(wl, cout_pad, cin_pad):
    """[CinPad/16][taps][CoutPad][16] of a logical (co, c, tap) weight"""
    co, c, taps = wl.shape
    full = np.zeros((cout_pad, cin_pad, taps), dtype=np.float32)
    full[:co, :c] = wl
    return np.ascontiguousarray(full.reshape(cout_pad, cin_pad // 16, 16, taps).transpose(1, 3, 0, 2))


def _packed_f16x2(wl, cout, cin_pad):
    """[CinPad/16][taps][h|l][2][Cout][8] split-fp16 pieces"""
    co, c, taps = wl.shape
    full = np.zeros((cout, cin_pad, taps), dtype=np.float32)
    full[:co, :c] = wl
    h = full.astype(np.float16)
    l = (full - h.astype(np.float32)).astype(np.float16)
    out = np.stack([h, l])  # (2, Cout, CinPad, taps)
    out = out.reshape(2, cout, cin_pad // 16, 2, 8, taps).transpose(2, 5, 0, 3, 1, 4)
    return np.ascontiguousarray(out)


def test_pack_jobs():
    rng = np.random.default_rng(21)
    L = S.lib()
    w3 = rng.standard_normal((64, 40, 3, 3)).astype(np.float32)     # Cin 40 -> padded to 48
    w1 = rng.standard_normal((64, 128, 1, 1)).astype(np.float32)
    wsm = rng.standard_normal((3, 64, 3, 3)).astype(np.float32)     # conv_out: Cout 3
    bias = rng.standard_normal(3).astype(np.float32)
    jobs, outs, wants = [], [], []

    def add(src, dst_shape, dtype, want, **kw):
        dst = np.full(dst_shape, np.nan if dtype == np.float32 else 0, dtype=dtype)
        if dtype == np.float16:
            dst[...] = np.float16(np.nan)
        j = nv.PackJob()
        j.src, j.dst = S.ptr(src), S.ptr(dst)
        for k, v in kw.items():
            setattr(j, k, v)
        jobs.append(j)
        outs.append(dst)
        wants.append(want)

    lw3 = w3.reshape(64, 40, 9)
    add(w3, (3, 9, 64, 16), np.float32, _packed_f32(lw3, 64, 48), Cout=64, Cin=40, k=3, kind=nv.PACK_F32, CoutPad=64, CinPad=48)
    add(w3, (3, 9, 2, 2, 64, 8), np.float16, _packed_f16x2(lw3, 64, 48), Cout=64, Cin=40, k=3, kind=nv.PACK_F16X2, CoutPad=64, CinPad=48)
    add(w1, (8, 1, 2, 2, 64, 8), np.float16, _packed_f16x2(w1.reshape(64, 128, 1), 64, 128), Cout=64, Cin=128, k=1, kind=nv.PACK_F16X2,
        CoutPad=64, CinPad=128)
    add(wsm, (4, 9, 16, 16), np.float32, _packed_f32(wsm.reshape(3, 64, 9), 16, 64), Cout=3, Cin=64, k=3, kind=nv.PACK_F32, CoutPad=16, CinPad=64)
    add(bias, (16,), np.float32, np.concatenate([bias, np.zeros(13, np.float32)]), Cout=3, Cin=0, k=1, kind=nv.PACK_BIAS, CoutPad=16, CinPad=0)
    # data-gradient weights: W'[co][c][tap] = W[c][c0 + co][taps - 1 - tap], co < c1 - c0 input channels of the slice, c < Cout
    for c0, c1 in ((0, 40), (8, 40)):
        wt = lw3[:, c0:c1, ::-1].transpose(1, 0, 2)  # (c1 - c0, 64, 9)
        cp = (c1 - c0 + 15) // 16 * 16
        add(w3, (4, 9, cp, 16), np.float32, _packed_f32(wt, cp, 64), Cout=64, Cin=40, k=3, kind=nv.PACK_F32, transposed=1, c0=c0, c1=c1,
            CoutPad=cp, CinPad=64)
    wt = w1.reshape(64, 128, 1)[:, 64:128].transpose(1, 0, 2)  # second half of a concatenated input: (64, 64, 1)
    add(w1, (4, 1, 2, 2, 64, 8), np.float16, _packed_f16x2(wt, 64, 64), Cout=64, Cin=128, k=1, kind=nv.PACK_F16X2, transposed=1, c0=64, c1=128,
        CoutPad=64, CinPad=64)

    table = (nv.PackJob * len(jobs))(*jobs)
    S.check(L.dmd_pack_jobs(table, len(jobs), max(o.size for o in outs), None), "dmd_pack_jobs")
    for i, (got, want) in enumerate(zip(outs, wants)):
        assert got.shape == want.shape, (i, got.shape, want.shape)
        assert np.array_equal(got.view(np.uint16 if got.dtype == np.float16 else np.uint32),
                              want.astype(got.dtype).view(np.uint16 if got.dtype == np.float16 else np.uint32)), f"job {i}"
    # dmd_pack_conv_weight is the same layout
    single = np.full((3, 9, 64, 16), np.nan, dtype=np.float32)
    S.check(L.dmd_pack_conv_weight(S.ptr(w3), S.ptr(single), 64, 40, 3, 64, 48, None), "dmd_pack_conv_weight")
    assert np.array_equal(single, wants[0])


# ---- pointwise family --------------------------------------------------------------------------------------------------------------
f32 = np.float32


def _quantise(d):
    """denoiser.py:81-83 in fp32, op by op: clamp(-1, 1).add(1).div(2).mul(255).byte().div(255).mul(2).sub(1)"""
    d = np.clip(d, f32(-1), f32(1))
    d = ((d + f32(1)) / f32(2)) * f32(255)
    q = d.astype(np.uint8).astype(np.float32)  # truncation
    return (q / f32(255)) * f32(2) - f32(1)


def test_edm_pointwise_ops_are_bit_exact():
    rng = np.random.default_rng(31)
    L = S.lib()
    n, cx, frames, h, w = 3, 3, 4, 8, 8
    cobs = frames * 3
    x = rng.standard_normal((n, cx, h, w)).astype(f32)
    ring = (rng.random((n, frames, 3, h, w)) * 2 - 1).astype(f32)
    cond = rng.random((n, 4)).astype(f32) + f32(0.1)
    head, sd, cpad = 3, f32(0.5), 16
    out = np.full((n, h, w, cpad), np.nan, dtype=f32)
    S.check(L.dmd_edm_pack_input(S.ptr(x), S.ptr(ring), S.ptr(cond), 4, float(sd), S.ptr(out), n, cx, cobs, h, w, cpad, frames, head, None), "pack")
    logical = np.roll(ring, -head, axis=1).reshape(n, cobs, h, w)  # logical frame t is stored at slot (head + t) % T
    want = np.zeros((n, cpad, h, w), dtype=f32)
    want[:, :cobs] = logical / sd
    want[:, cobs:cobs + cx] = x * cond[:, 0, None, None, None]
    assert np.array_equal(out, want.transpose(0, 2, 3, 1))

    f = rng.standard_normal((n, cx, h, w)).astype(f32)
    den = np.full_like(x, np.nan)
    S.check(L.dmd_edm_denoised(S.ptr(x), S.ptr(f), S.ptr(cond), 4, S.ptr(den), n, cx * h * w, None), "denoised")
    assert np.array_equal(den, _quantise(cond[:, 2, None, None, None] * x + cond[:, 1, None, None, None] * f))

    sh, sn, dt = f32(2.5), f32(1.25), f32(-1.25)
    xo = np.full_like(x, np.nan)
    S.check(L.dmd_euler_step(S.ptr(x), S.ptr(den), float(sh), float(dt), S.ptr(xo), x.size, None), "euler")
    assert np.array_equal(xo, x + ((x - den) / sh) * dt)
    x2, den2 = xo.copy(), _quantise(xo * f32(0.7))
    xh = np.full_like(x, np.nan)
    S.check(L.dmd_heun_step(S.ptr(x), S.ptr(den), S.ptr(x2), S.ptr(den2), float(sh), float(sn), float(dt), S.ptr(xh), x.size, None), "heun")
    assert np.array_equal(xh, x + ((((x - den) / sh) + ((x2 - den2) / sn)) / f32(2)) * dt)


def test_cond_embed_ring_and_layout_transposes():
    rng = np.random.default_rng(32)
    L = S.lib()
    n, half, t, a_rows = 3, 32, 4, 6
    e = 2 * half // t
    cond = rng.random((n, 4)).astype(f32)
    fw = rng.standard_normal(half).astype(f32)
    act = rng.integers(0, a_rows, (n, t)).astype(np.int64)
    emb = rng.standard_normal((a_rows, e)).astype(f32)
    out = np.full((n, 2 * half), np.nan, dtype=f32)
    S.check(L.dmd_cond_embed(S.ptr(cond), 4, S.ptr(fw), S.ptr(act), S.ptr(emb), S.ptr(out), n, half, t, e, 1, a_rows, None), "cond_embed")
    fr = (f32(2.0 * np.pi) * cond[:, 3:4]) * fw[None]
    want = np.concatenate([np.cos(fr), np.sin(fr)], axis=1) + emb[np.roll(act, -1, axis=1)].reshape(n, -1)
    assert np.abs(out - want).max() < 2e-6  # cosf / sinf: libm here, ocml on the device

    x = rng.standard_normal((2, 5, 4, 6)).astype(f32)
    nhwc = np.full((2, 4, 6, 8), np.nan, dtype=f32)
    S.check(L.dmd_nchw_to_nhwc(S.ptr(x), S.ptr(nhwc), 2, 5, 4, 6, 8, None), "nchw_to_nhwc")
    assert np.array_equal(nhwc[..., :5], x.transpose(0, 2, 3, 1)) and not nhwc[..., 5:].any()
    back = np.full_like(x, np.nan)
    S.check(L.dmd_nhwc_to_nchw(S.ptr(nhwc), S.ptr(back), 2, 5, 4, 6, 8, None), "nhwc_to_nchw")
    assert np.array_equal(back, x)


def test_uint8_pool_round_trip():
    rng = np.random.default_rng(33)
    L = S.lib()
    p_, t, per = 5, 4, 48
    levels = rng.integers(0, 256, (p_, t, per)).astype(np.uint8)
    frames = (levels.astype(f32) / f32(255)) * f32(2) - f32(1)
    q = np.zeros_like(levels)
    off = np.zeros(1, dtype=np.int32)
    S.check(L.dmd_quantize_u8(S.ptr(frames), S.ptr(q), S.ptr(off), frames.size, None), "quantize")
    assert np.array_equal(q, levels) and off[0] == 0
    bad = frames.copy()
    bad[2, 1, 7] += f32(1e-3)
    S.check(L.dmd_quantize_u8(S.ptr(bad), S.ptr(q), S.ptr(off), bad.size, None), "quantize")
    assert off[0] == 1  # a frame off the 256-level grid cannot live in the uint8 pool
    idx, rows = np.array([4, 0, 2], dtype=np.int64), np.array([1, 3, 0], dtype=np.int64)
    dst = np.full((4, t, per), np.nan, dtype=f32)
    S.check(L.dmd_dequant_gather(S.ptr(levels), S.ptr(idx), S.ptr(rows), S.ptr(dst), 3, t, per, 2, None), "dequant_gather")
    for i in range(3):
        assert np.array_equal(np.roll(dst[rows[i]], -2, axis=0), frames[idx[i]])  # logical frame k at slot (head + k) % T
    assert np.isnan(dst[2]).all()


def _resolve_np(end, ep_len, horizon, K):
    """numpy restatement of dmd_resolve_deaths (reference world_model_env.py:71-72,77-83 + the slot assignment in row order)"""
    ep = ep_len + 1
    trunc = (ep >= horizon).astype(np.int64)
    dead = (end != 0) | (trunc != 0)
    rows = np.flatnonzero(dead)
    slot_row = np.full(K, -1, dtype=np.int64)
    slot_row[:min(K, rows.size)] = rows[:K]
    row_slot = np.full(end.size, -1, dtype=np.int32)
    row_slot[rows[:K]] = np.arange(min(K, rows.size))
    report = np.concatenate([dead.astype(np.int32), np.array([rows.size, int((end != 0).sum()), int(rows.size > K), K], dtype=np.int32)])
    return np.where(dead, 0, ep), trunc, dead.astype(np.uint8), slot_row, row_slot, report


@pytest.mark.parametrize("b,k,p_end", [(9, 4, 0.2), (256, 24, 0.02), (256, 0, 0.0), (300, 300, 0.5), (700, 16, 0.05), (5, 8, 1.0)])
def test_resolve_deaths_matches_the_row_order_of_the_reference(b, k, p_end):
    """the step's deaths resolved on the device: truncation, dead mask, episode lengths, slots in ascending row order, unused slots,
    the report the host reads a step late -- overflow (more deaths than slots) included; batches above one 256-row chunk"""
    rng = np.random.default_rng(b * 131 + k)
    L = S.lib()
    horizon = 7
    end = (rng.random(b) < p_end).astype(np.int64)
    ep_len = rng.integers(0, horizon, b).astype(np.int64)
    want = _resolve_np(end, ep_len, horizon, k)
    ep = ep_len.copy()
    trunc, dead = np.full(b, -1, dtype=np.int64), np.full(b, 7, dtype=np.uint8)
    slot_row, row_slot, report = np.full(max(k, 1), -9, dtype=np.int64), np.full(b, -9, dtype=np.int32), np.full(b + 4, -9, dtype=np.int32)
    S.check(L.dmd_resolve_deaths(S.ptr(end), S.ptr(ep), horizon, S.ptr(trunc), S.ptr(dead), b, k, S.ptr(slot_row), S.ptr(row_slot),
                                 S.ptr(report), None), "resolve_deaths")
    for got, w, name in zip((ep, trunc, dead, slot_row[:k], row_slot, report), want, ("ep_len", "trunc", "dead", "slot_row", "row_slot", "report")):
        assert np.array_equal(got, w), name
    assert (report[b + 2] == 1) == (int(want[2].sum()) > k)


@pytest.mark.parametrize("u8,pad", [(True, False), (True, True), (False, False)])
def test_reset_slots_is_the_ring_advance_plus_the_reference_reset(u8, pad):
    """dmd_reset_slots against the sequence it replaces: ring advance (the oldest slot receives the imagined frame), the reference's
    reset_dead for the dead rows (context frames, actions, reward/end LSTM state from pool rows served in row order), and the policy's
    next input [newest frames | final observations | burn-in frames, frame-major]; unused slots touch nothing"""
    from diamond_amd import native as nv
    import ctypes as C

    rng = np.random.default_rng(77 + u8 + 2 * pad)
    L = S.lib()
    b, k, t, per, hd, p_, head_old, base = 7, 5, 4, 24, 12, 11, 2, 3
    levels = rng.integers(0, 256, (p_, t, per)).astype(np.uint8)
    padm = (rng.random((p_, t)) < 0.3).astype(np.uint8) if pad else None
    frames = (levels.astype(f32) / f32(255)) * f32(2) - f32(1)
    if pad:
        frames[padm.astype(bool)] = 0.0
    pool = levels if u8 else frames.copy()
    pool_act = rng.integers(0, 18, (p_, t)).astype(np.int64)
    pool_hx, pool_cx = rng.standard_normal((p_, hd)).astype(f32), rng.standard_normal((p_, hd)).astype(f32)
    ctx = rng.standard_normal((b, t, per)).astype(f32)
    act = rng.integers(0, 18, (b, t)).astype(np.int64)
    hx, cx = rng.standard_normal((b, hd)).astype(f32), rng.standard_normal((b, hd)).astype(f32)
    nxt = rng.standard_normal((b, per)).astype(f32)
    dead_rows = np.array([1, 4, 6])
    slot_row = np.full(k, -1, dtype=np.int64)
    slot_row[:3] = dead_rows
    row_slot = np.full(b, -1, dtype=np.int32)
    row_slot[dead_rows] = np.arange(3)
    head = (head_old + 1) % t
    # the reference's order of operations on a ring
    w_ctx, w_act, w_hx, w_cx = ctx.copy(), act.copy(), hx.copy(), cx.copy()
    w_ctx[:, head_old] = nxt
    cols = (head + np.arange(t)) % t
    idx = base + np.arange(3)
    w_ctx[dead_rows[:, None], cols[None, :]] = frames[idx]
    w_act[dead_rows[:, None], cols[None, :]] = pool_act[idx]
    w_hx[dead_rows], w_cx[dead_rows] = pool_hx[idx], pool_cx[idx]
    obs = nxt.copy()
    obs[dead_rows] = frames[idx, t - 1]
    fin = np.repeat(nxt[:1], k, 0)
    fin[:3] = nxt[dead_rows]
    burn = np.repeat(nxt[None, :1], t - 1, 0).repeat(k, 1)  # (t - 1, k, per) frame-major
    burn[:, :3] = frames[idx, :t - 1].transpose(1, 0, 2)
    want_enc = np.concatenate([obs, fin, burn.reshape(-1, per)])
    enc = np.full((b + t * k, per), np.nan, dtype=f32)
    def round_(frames_, pad_, act_, hx_, cx_, f32_):
        r = nv.PoolRound()
        r.frames, r.pad, r.act, r.hx, r.cx, r.is_f32, r.rows = S.ptr(frames_), S.ptr(pad_), S.ptr(act_), S.ptr(hx_), S.ptr(cx_), int(f32_), frames_.shape[0]
        return r

    p = nv.ResetSlotsParams()
    p.B, p.K, p.T, p.head, p.per_frame, p.hd, p.pool_base = b, k, t, head, per, hd, base
    p.pool[0] = round_(pool, padm if (pad and u8) else None, pool_act, pool_hx, pool_cx, not u8)
    p.slot_row, p.row_slot, p.next_obs, p.ctx, p.act_ring, p.hx, p.cx, p.enc_in = (S.ptr(a) for a in (slot_row, row_slot, nxt, ctx, act, hx, cx, enc))
    start = [a.copy() for a in (ctx, act, hx, cx)]
    S.check(L.dmd_reset_slots(C.byref(p), None), "reset_slots")
    for got, w, name in zip((ctx, act, hx, cx, enc), (w_ctx, w_act, w_hx, w_cx, want_enc), ("ctx", "act", "hx", "cx", "enc_in")):
        assert np.array_equal(got, w), name
    # two rounds: the step's deaths fit what is left of the first one (rows base .. base + 3 <= p_) -> the same result; they do
    # not (a first round of base + 2 rows) -> the SECOND round's rows 0.. (the reference drops the remainder and preloads, :133-139)
    n_dead = np.array([3], dtype=np.int32)
    other = (rng.integers(0, 256, (p_, t, per)).astype(np.uint8), rng.integers(0, 18, (p_, t)).astype(np.int64),
             rng.standard_normal((p_, hd)).astype(f32), rng.standard_normal((p_, hd)).astype(f32))
    p.num_dead = S.ptr(n_dead)
    for first_is_short in (False, True):
        for a, a0 in zip((ctx, act, hx, cx), start):
            a[...] = a0
        enc[...] = np.nan
        if first_is_short:  # the round used above is now the SECOND one: served from its row 0
            p.pool[1] = round_(pool[base:], padm[base:] if (pad and u8) else None, pool_act[base:], pool_hx[base:], pool_cx[base:], not u8)
            p.pool[0] = round_(other[0][:base + 2], None, other[1], other[2], other[3], False)
        else:
            p.pool[1] = round_(*((other[0], None) + other[1:]), False)
        S.check(L.dmd_reset_slots(C.byref(p), None), "reset_slots, two rounds")
        for got, w, name in zip((ctx, act, hx, cx, enc), (w_ctx, w_act, w_hx, w_cx, want_enc), ("ctx", "act", "hx", "cx", "enc_in")):
            assert np.array_equal(got, w), (first_is_short, name)
    p.pool[0] = round_(pool, padm if (pad and u8) else None, pool_act, pool_hx, pool_cx, not u8)
    p.pool[1] = nv.PoolRound()
    p.num_dead = None
    # no slots at all: the ring advance and a copy of the imagined frames
    ctx2, enc2 = rng.standard_normal((b, t, per)).astype(f32), np.full((b, per), np.nan, dtype=f32)
    w2 = ctx2.copy()
    w2[:, head_old] = nxt
    none = np.full(b, -1, dtype=np.int32)
    p.K, p.row_slot, p.ctx, p.enc_in = 0, S.ptr(none), S.ptr(ctx2), S.ptr(enc2)
    S.check(L.dmd_reset_slots(C.byref(p), None), "reset_slots K=0")
    assert np.array_equal(ctx2, w2) and np.array_equal(enc2, nxt)


def test_merge_slots_and_its_backward():
    rng = np.random.default_rng(5)
    L = S.lib()
    b, k, d = 9, 4, 40
    base, slots, g = (rng.standard_normal(sh).astype(f32) for sh in ((b, d), (k, d), (b, d)))
    slot_row = np.array([2, 7, -1, -1], dtype=np.int64)
    row_slot = np.full(b, -1, dtype=np.int32)
    row_slot[[2, 7]] = [0, 1]
    out = np.full((b, d), np.nan, dtype=f32)
    S.check(L.dmd_merge_slots(S.ptr(base), S.ptr(slots), S.ptr(row_slot), S.ptr(out), b, d, None), "merge_slots")
    want = base.copy()
    want[[2, 7]] = slots[:2]
    assert np.array_equal(out, want)
    d_base, d_slots = np.full((b, d), np.nan, dtype=f32), np.full((k, d), np.nan, dtype=f32)
    S.check(L.dmd_merge_slots_bwd(S.ptr(g), S.ptr(row_slot), S.ptr(slot_row), S.ptr(d_base), S.ptr(d_slots), b, k, d, None), "merge_slots_bwd")
    wb = g.copy()
    wb[[2, 7]] = 0
    ws = np.zeros((k, d), dtype=f32)
    ws[:2] = g[[2, 7]]
    assert np.array_equal(d_base, wb) and np.array_equal(d_slots, ws)


def test_reset_state_and_parameter_checksums():
    """dmd_reset_state (the action ring, reward/end LSTM state and episode length of the reset rows in one launch: reference
    world_model_env.py:56-62) against the indexed assignments it replaces; dmd_checksums (fingerprints of parameter storage for
    the audit of the packed copies) against 64-bit sums of the 32-bit words, and its sensitivity to a single flipped bit"""
    rng = np.random.default_rng(35)
    L = S.lib()
    b, p_, t, hd, head = 6, 9, 4, 40, 3
    pool_act = rng.integers(0, 18, (p_, t)).astype(np.int64)
    pool_hx, pool_cx = rng.standard_normal((p_, hd)).astype(f32), rng.standard_normal((p_, hd)).astype(f32)
    act = rng.integers(0, 18, (b, t)).astype(np.int64)
    hx, cx = rng.standard_normal((b, hd)).astype(f32), rng.standard_normal((b, hd)).astype(f32)
    ep = rng.integers(1, 9, b).astype(np.int64)
    idx, rows = np.array([7, 2, 8], dtype=np.int64), np.array([4, 0, 5], dtype=np.int64)
    want = [a.copy() for a in (act, hx, cx, ep)]
    cols = (head + np.arange(t)) % t
    want[0][rows[:, None], cols[None, :]] = pool_act[idx]
    want[1][rows], want[2][rows], want[3][rows] = pool_hx[idx], pool_cx[idx], 0
    S.check(L.dmd_reset_state(S.ptr(idx), S.ptr(rows), 3, S.ptr(pool_act), S.ptr(act), t, head, S.ptr(pool_hx), S.ptr(pool_cx), S.ptr(hx),
                              S.ptr(cx), hd, S.ptr(ep), None), "reset_state")
    for got, w in zip((act, hx, cx, ep), want):
        assert np.array_equal(got, w)

    from diamond_amd import native as nv
    import ctypes as C

    tensors = [rng.standard_normal(n).astype(f32) for n in (1, 255, 256, 4097, 70000)]
    jobs = (nv.ChecksumJob * len(tensors))()
    for j, a in zip(jobs, tensors):
        j.src, j.words = a.ctypes.data, a.size
    table = np.frombuffer(bytes(jobs), dtype=np.uint8).copy()
    out = np.full((len(tensors), nv.CHECKSUM_PARTS), -1, dtype=np.int64)
    S.check(L.dmd_checksums(S.ptr(table), len(tensors), S.ptr(out), None), "checksums")
    for a, o in zip(tensors, out):
        w = a.view(np.int32).astype(np.int64)
        part = (np.arange(a.size) // 256) % nv.CHECKSUM_PARTS
        assert np.array_equal(o, np.array([w[part == k].sum() for k in range(nv.CHECKSUM_PARTS)]))
    tensors[3].view(np.int32)[1234] ^= 1  # one bit
    out2 = np.zeros_like(out)
    S.check(L.dmd_checksums(S.ptr(table), len(tensors), S.ptr(out2), None), "checksums")
    assert (out2 != out).any(axis=1).tolist() == [False, False, False, True, False]


def test_maxpool_lstm_categorical():
    rng = np.random.default_rng(34)
    L = S.lib()
    n, h, w, c = 2, 8, 8, 32
    x = rng.standard_normal((n, h, w, c)).astype(f32)
    x[0, 0, 0, 0] = x[0, 0, 1, 0] = 9.0  # a tie: the FIRST maximum in scan order wins
    out = np.full((n, h // 2, w // 2, c), np.nan, dtype=f32)
    arg = np.full((n, h // 2, w // 2, c), 255, dtype=np.uint8)
    stats = np.full((n, 1, 1, 2), np.nan)
    S.check(L.dmd_maxpool2(S.ptr(x), S.ptr(out), S.ptr(arg), S.ptr(stats), n, h, w, c, None), "maxpool2")
    win = x.reshape(n, h // 2, 2, w // 2, 2, c).transpose(0, 1, 3, 5, 2, 4).reshape(n, h // 2, w // 2, c, 4)
    assert np.array_equal(out, win.max(axis=-1)) and np.array_equal(arg, win.argmax(axis=-1))
    np.testing.assert_allclose(stats[:, 0, 0], _group_sums(out)[:, 0], rtol=1e-12)
    dp = rng.standard_normal(out.shape).astype(f32)
    dx = np.full_like(x, np.nan)
    S.check(L.dmd_maxpool2_bwd(S.ptr(dp), S.ptr(arg), S.ptr(dx), n, h, w, c, None), "maxpool2_bwd")
    want = np.zeros_like(win)
    np.put_along_axis(want, win.argmax(axis=-1)[..., None], dp[..., None], axis=-1)
    assert np.array_equal(dx, want.reshape(n, h // 2, w // 2, c, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(n, h, w, c))

    b, hd = 3, 40
    gates = (rng.standard_normal((b, 4 * hd)) * 2).astype(f32)
    c0 = rng.standard_normal((b, hd)).astype(f32)
    hh, cc = np.full((b, hd), np.nan, dtype=f32), np.full((b, hd), np.nan, dtype=f32)
    S.check(L.dmd_lstm_pointwise(S.ptr(gates), S.ptr(c0), S.ptr(hh), S.ptr(cc), b, hd, None), "lstm")
    sig = lambda v: 1 / (1 + np.exp(-v.astype(np.float64)))
    i, f, g, o = (gates[:, k * hd:(k + 1) * hd] for k in range(4))
    c_ref = sig(f) * c0 + sig(i) * np.tanh(g.astype(np.float64))
    h_ref = sig(o) * np.tanh(c_ref)
    assert np.abs(cc - c_ref).max() < 1e-6 and np.abs(hh - h_ref).max() < 1e-6
    dh, dc = rng.standard_normal((b, hd)).astype(f32), rng.standard_normal((b, hd)).astype(f32)
    dg, dcp = np.full_like(gates, np.nan), np.full((b, hd), np.nan, dtype=f32)
    S.check(L.dmd_lstm_pointwise_bwd(S.ptr(gates), S.ptr(c0), S.ptr(cc), S.ptr(dh), S.ptr(dc), S.ptr(dg), S.ptr(dcp), b, hd, None), "lstm_bwd")
    tc = np.tanh(c_ref)
    dcv = dc + dh * sig(o) * (1 - tc * tc)
    gg = np.tanh(g.astype(np.float64))
    want = np.concatenate([dcv * gg * sig(i) * (1 - sig(i)), dcv * c0 * sig(f) * (1 - sig(f)), dcv * sig(i) * (1 - gg * gg),
                           dh * tc * sig(o) * (1 - sig(o))], axis=1)
    assert np.abs(dg - want).max() < 2e-6 and np.abs(dcp - dcv * sig(f)).max() < 2e-6

    logits = (rng.standard_normal((50, 7)) * 2).astype(f32)
    expo = rng.exponential(size=(50, 7)).astype(f32)
    act = np.full(50, -1, dtype=np.int64)
    S.check(L.dmd_categorical_sample(S.ptr(logits), S.ptr(expo), S.ptr(act), 50, 7, None), "categorical")
    l64 = logits.astype(np.float64)
    probs = np.exp(l64 - l64.max(axis=1, keepdims=True))
    probs /= probs.sum(axis=1, keepdims=True)
    assert np.array_equal(act, (probs / expo).argmax(axis=1))  # torch.multinomial: argmax(probs / E)


# ---- dmd_lowres_chain: a chain of ResBlocks (+ attention, concatenated skips) at the 8x8 level in one launch -----------------------
def _pack16(w_oihw):
    """split-fp16 pack of an OIHW weight through dmd_pack_jobs"""
    co, ci, k, _ = w_oihw.shape
    dst = G(np.zeros((ci // 16, k * k, 2, 2, co, 8), dtype=np.float16))
    j = nv.PackJob()
    j.src, j.dst, j.Cout, j.Cin, j.k, j.kind, j.CoutPad, j.CinPad = S.ptr(w_oihw), S.ptr(dst), co, ci, k, nv.PACK_F16X2, co, ci
    S.check(S.lib().dmd_pack_jobs((nv.PackJob * 1)(j), 1, dst.size, None), "dmd_pack_jobs")
    return dst


def _ada_gn_silu(x, scale, shift):
    return _apply_norm(x, x.shape[1], x.shape[2], scale, shift, True, silu=True)


@pytest.mark.parametrize("schedule", SCHEDULES)
@pytest.mark.parametrize("c", [64, 32], ids=["denoiser-64ch", "rew-end-32ch"])
def test_lowres_chain(c, schedule, monkeypatch, request):
    monkeypatch.setenv("SIMT_SCHEDULE", str(schedule))
    rng = np.random.default_rng(41 + c)
    L = S.lib()
    n = 2
    x = rng.standard_normal((n, 8, 8, c)).astype(np.float32)
    table = (rng.standard_normal((n, 40 * c)) * 0.3).astype(np.float32)
    col = [0]

    def take(k):
        col[0] += k
        return col[0] - k

    def conv_w(co, ci, k):
        return (rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k)).astype(np.float32)

    # block 0: plain, output kept in slot 1; block 1: attention; block 2 (64-channel chain only): cat(x, slot 1) + projection + attention
    # (the 32-channel chain has no skip slots)
    specs = [dict(cat=False, attn=False, save=1 if c == 64 else -1), dict(cat=False, attn=True, save=-1)]
    if c == 64:
        specs.append(dict(cat=True, attn=True, save=-1))
    p = nv.LowresChainParams()
    p.N, p.nblocks, p.input_save_slot = n, len(specs), (0 if c == 64 else -1)
    out = np.full_like(x, np.nan)
    p.x, p.out, p.table, p.table_stride = S.ptr(x), S.ptr(out), S.ptr(table), table.shape[1]
    keep, blocks = [], []
    for i, sp in enumerate(specs):
        cin = 2 * c if sp["cat"] else c
        b = dict(sp, w1=conv_w(c, cin, 3), w2=conv_w(c, c, 3), b1=rng.standard_normal(c).astype(np.float32),
                 b2=rng.standard_normal(c).astype(np.float32))
        cb = p.blocks[i]
        cb.skip_slot, cb.save_slot = (1 if sp["cat"] else -1), sp["save"]
        o1 = take(2 * cin)
        cb.film1_mul[0], cb.film1_add[0] = o1, o1 + cin
        cb.film1_mul[1], cb.film1_add[1] = o1 + c, o1 + cin + c
        b["f1"] = (table[:, o1:o1 + cin], table[:, o1 + cin:o1 + 2 * cin])
        o2 = take(2 * c)
        cb.film2_mul, cb.film2_add = o2, o2 + c
        b["f2"] = (table[:, o2:o2 + c], table[:, o2 + c:o2 + 2 * c])
        packs = [_pack16(b["w1"]), _pack16(b["w2"])]
        cb.w1, cb.w2, cb.b1, cb.b2 = S.ptr(packs[0]), S.ptr(packs[1]), S.ptr(b["b1"]), S.ptr(b["b2"])
        if sp["cat"]:
            b["wp"], b["bp"] = conv_w(c, cin, 1), rng.standard_normal(c).astype(np.float32)
            packs.append(_pack16(b["wp"]))
            cb.wproj, cb.bproj = S.ptr(packs[-1]), S.ptr(b["bp"])
        if sp["attn"]:
            cb.has_attn = 1
            b["wqkv"], b["bqkv"] = conv_w(3 * c, c, 1), (rng.standard_normal(3 * c) * 0.2).astype(np.float32)
            b["wo"], b["bo"] = conv_w(c, c, 1), rng.standard_normal(c).astype(np.float32)
            b["gam"], b["bet"] = (1 + 0.2 * rng.standard_normal(c)).astype(np.float32), (0.2 * rng.standard_normal(c)).astype(np.float32)
            qkv = [_pack16(np.ascontiguousarray(b["wqkv"][k * c:(k + 1) * c])) for k in range(3)]
            packs += qkv + [_pack16(b["wo"])]
            cb.wq, cb.wk, cb.wv, cb.wo = (S.ptr(t) for t in packs[-4:])
            cb.gn_gamma, cb.gn_beta, cb.bqkv, cb.bo = S.ptr(b["gam"]), S.ptr(b["bet"]), S.ptr(b["bqkv"]), S.ptr(b["bo"])
        keep.append(packs)
        blocks.append(b)
    S.check((L.dmd_lowres_chain if c == 64 else L.dmd_lowres_chain32)(p, None), "dmd_lowres_chain")
    _same_bits_as_schedule_0(request, schedule, out)

    # fp64 restatement of ResBlock.forward (blocks.py:141-147) / SelfAttention2d.forward (blocks.py:62-72)
    cur, slots = x.astype(np.float64), {0: x.astype(np.float64)}
    for b in blocks:
        inp = np.concatenate([cur, slots[1]], axis=-1) if b["cat"] else cur
        r = _ref_conv([inp], b["wp"], b["bp"], 1, 1, 0, 8, 8) if b["cat"] else cur
        h = _ref_conv([_ada_gn_silu(inp, *b["f1"])], b["w1"], b["b1"], 3, 1, 0, 8, 8)
        cur = _ref_conv([_ada_gn_silu(h, *b["f2"])], b["w2"], b["b2"], 3, 1, 0, 8, 8) + r
        if b["attn"]:
            gam, bet = np.repeat(b["gam"][None], n, 0), np.repeat(b["bet"][None], n, 0)
            xn = _apply_norm(cur, 8, 8, gam, bet, False, silu=False)
            qkv = _ref_conv([xn], b["wqkv"], b["bqkv"], 1, 1, 0, 8, 8).reshape(n, 64, 3 * c)
            y, _, _ = _ref_attention(qkv, c)
            cur = xn + _ref_conv([y.reshape(n, 8, 8, c)], b["wo"], b["bo"], 1, 1, 0, 8, 8)
        if b["save"] >= 0:
            slots[b["save"]] = cur
    err = np.abs(out - cur).max() / np.abs(cur).max()
    assert err <= 2e-5, err


# ---- conv_f16ws_kernel: the dominant, wave-specialised persistent kernel (its inline assembly spelled in C++ for this build) --------
WS_CASES = [
    dict(n=2, h=16, w=16, cin=[64], cout=64, k=3, prologue=[1], film=True, stats=True, residual="raw"),     # WsGeom<false, 2, 9>
    dict(n=3, h=16, w=32, cin=[64, 64], cout=64, k=3, prologue=[1, 1], film=True, stats=True),             # cat(x, skip), 6 tiles on 3 "CUs"
    dict(n=1, h=16, w=16, cin=[16], cout=64, k=3),                                                          # conv_in: one chunk
    dict(n=2, h=8, w=8, cin=[64], cout=64, k=3, prologue=[1], stats=True),                                  # WsGeom<true, 2, 9>: 8x8 blocks of 2 images in a tile
    dict(n=5, h=8, w=8, cin=[64, 64], cout=64, k=3, prologue=[1, 0], residual="raw"),                       # ... a tile with one block only
    dict(n=2, h=16, w=16, cin=[32], cout=32, k=3, prologue=[1], film=True, stats=True, residual="raw"),     # WsGeom<false, 1, 9>
    dict(n=3, h=8, w=8, cin=[32, 32], cout=32, k=3, prologue=[1, 1], stats=True),                           # WsGeom<true, 1, 9>
    dict(n=1, h=16, w=16, cin=[64], cout=64, k=1, prologue=[2], residual="raw"),                            # 1x1 with a prologue: WsGeom<false, 2, 1>
    dict(n=2, h=8, w=8, cin=[32], cout=32, k=1, prologue=[2]),                                              # WsGeom<true, 1, 1>
    dict(n=2, h=32, w=32, cin=[64], cout=64, k=3, upsample=1, stats=True),                                  # Upsample's conv
    dict(n=1, h=16, w=16, cin=[64], cout=3, k=3, prologue=[1], nchw=True),                                  # conv_out: few-channel NCHW head
    dict(n=1, h=32, w=32, cin=[64], cout=64, k=3, prologue=[1], valid=(18, 27), stats=True),                # valid extent
    dict(n=1, h=16, w=16, cin=[64], cout=64, k=3, prologue=[1], valid=(9, 9), stats=True, force_b8=False),
    dict(n=2, h=16, w=32, cin=[64], cout=64, k=3, prologue=[1], film=True, stats=True, proj=True),          # WsGeomProj: fused skip projection
]


@pytest.mark.parametrize("schedule", SCHEDULES)
@pytest.mark.parametrize("case", WS_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()).replace(" ", ""))
def test_conv_f16ws(case, schedule, monkeypatch, request):
    monkeypatch.delenv("DIAMOND_CONV_LATENCY_TILES", raising=False)
    monkeypatch.setenv("SIMT_SCHEDULE", str(schedule))
    rng = np.random.default_rng(23)
    L = S.lib()
    n, h, w, cins, cout, k = case["n"], case["h"], case["w"], case["cin"], case["cout"], case["k"]
    up, prol = case.get("upsample", 0), case.get("prologue", [0] * len(cins))
    hv, wv = case.get("valid", (h, w))
    hs, ws = (h // 2, w // 2) if up else (h, w)
    hvs, wvs = (hv // 2, wv // 2) if up else (hv, wv)
    cin = sum(cins)
    cout_pad = 32 if cout < 32 else cout
    p = nv.ConvParams()
    p.N, p.H, p.W, p.Cout, p.CoutPad, p.taps, p.stride, p.upsample, p.nsrc, p.precision = n, h, w, cout, cout_pad, k * k, 1, up, len(cins), 1
    if "valid" in case:
        p.valid_h, p.valid_w = hv, wv
    keep, xs_ref = [], []
    for i, c in enumerate(cins):
        x = G((rng.standard_normal((n, hs, ws, c)) * 1.5 + 0.3).astype(np.float32), tight="start" if (n + hs + c) % 2 else "end")
        p.src[i].x, p.src[i].C, p.src[i].prologue = S.ptr(x), c, prol[i]
        if prol[i]:
            st = _partial_stats(x, hvs, wvs, 3, rng)
            film = case.get("film") or prol[i] == 2
            mul = (rng.standard_normal((n, c)) * 0.3).astype(np.float32) if film else None
            add = (rng.standard_normal((n, c)) * 0.3).astype(np.float32) if film else None
            p.src[i].norm = _norm(st, 3, mul, add, bool(case.get("film")))
            xs_ref.append(_apply_norm(x, hvs, wvs, mul, add, bool(case.get("film")), silu=prol[i] == 1))
            keep += [st, mul, add]
        else:
            xs_ref.append(x.astype(np.float64))
        keep.append(x)
    wt = np.zeros((cout_pad, cin, k, k), dtype=np.float32)
    wt[:cout] = rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)
    bias = np.zeros(cout_pad, dtype=np.float32)
    bias[:cout] = rng.standard_normal(cout)
    packed = np.zeros((cin // 16) * k * k * cout_pad * 16, dtype=np.float32)
    S.check(L.dmd_pack_conv_weight(S.ptr(wt), S.ptr(packed), cout_pad, cin, k, cout_pad, cin, None), "pack")
    w16 = _pack16(wt)
    p.w, p.bias, p.w_f16 = S.ptr(packed), S.ptr(bias), S.ptr(w16)
    ref = _ref_conv(xs_ref, wt[:cout], bias[:cout], k, 1, up, hv, wv)
    if case.get("residual"):
        r = G(rng.standard_normal((n, h, w, cout)).astype(np.float32))
        p.residual = S.ptr(r)
        ref = ref + r
    if case.get("proj"):
        j0, j1 = (G(rng.standard_normal((n, h, w, 64)).astype(np.float32)) for _ in range(2))
        wpj = (rng.standard_normal((cout, 128, 1, 1)) / np.sqrt(128)).astype(np.float32)
        bpj = rng.standard_normal(cout).astype(np.float32)
        wpj16 = _pack16(wpj)
        p.proj_nsrc, p.proj_C[0], p.proj_C[1] = 2, 64, 64
        p.proj_x[0], p.proj_x[1], p.proj_w_f16, p.proj_bias = S.ptr(j0), S.ptr(j1), S.ptr(wpj16), S.ptr(bpj)
        ref = ref + _ref_conv([j0.astype(np.float64), j1.astype(np.float64)], wpj, bpj, 1, 1, 0, h, w)
        assert L.dmd_conv2d_proj_eligible(p) == 1
    nchw = bool(case.get("nchw"))
    out = G(np.full((n, cout, h, w) if nchw else (n, h, w, cout), np.nan, dtype=np.float32))
    p.out, p.out_nchw = S.ptr(out), int(nchw)
    tiles = L.dmd_conv_stat_tiles(h, w)
    stats = G(np.full((n, cout // 32, tiles, 2), np.nan)) if case.get("stats") else None
    p.out_stats = S.ptr(stats)

    assert L.dmd_conv2d_f16x2_eligible(p) == 1
    buf = (nv.C.c_char * 96)()
    S.check(L.dmd_conv2d_kernel_name(p, buf, 96), "kernel_name")
    assert buf.value.decode().startswith("conv_f16ws_kernel<"), buf.value
    S.check(L.dmd_conv2d(p, None), "dmd_conv2d")
    got = out.transpose(0, 2, 3, 1) if nchw else out
    err = np.abs(got[:, :hv, :wv] - ref[:, :hv, :wv]).max()
    assert err <= 2e-5 * max(1.0, np.abs(ref).max()), (buf.value, err)
    _same_bits_as_schedule_0(request, schedule, out[:, :hv, :wv] if not nchw else out[:, :, :hv, :wv], stats)
    if stats is not None:
        want = _group_sums(np.ascontiguousarray(got.astype(np.float32)), hv, wv)
        np.testing.assert_allclose(stats.sum(axis=2), want, rtol=2e-6, atol=1e-4)  # (fp32 sums of a lane's 16 values inside)


def test_zz_schedule_comparisons_took_place():
    """(runs last in this file) the bitwise comparisons across wave schedules above were made, not skipped by a key mismatch"""
    if len(_BITS) < 50:
        pytest.skip("a subset of the file was selected")
    assert _COMPARED[0] >= 2 * len(_BITS) - 4, (_COMPARED[0], len(_BITS))
