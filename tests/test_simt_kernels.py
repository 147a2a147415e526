"""The kernels' own source, run lane by lane on the CPU (tests/simt: a SIMT interpreter, TEST INFRASTRUCTURE), against numpy
restatements of what each entry point of include/diamond_hip.h promises.  This checks index arithmetic, LDS layouts,
synchronisation structure and MFMA operand layouts without a GPU; the `-m gpu` tests remain the parity tests proper (the
interpreter's MFMA accumulation order is not the hardware's, and dmd_conv_f16ws.hip is not part of the host build)."""
import os

import numpy as np
import pytest

from tests.simt import loader as S


def test_product_refuses_the_host_build(monkeypatch):
    """diamond_amd has no CPU path: pointing it at the interpreter build must fail loudly."""
    import importlib
    S.lib()
    import diamond_amd.native as nv
    monkeypatch.setattr(nv, "LIB_PATH", S.LIB_PATH)
    monkeypatch.setattr(nv, "_lib", None)
    with pytest.raises(nv.NativeLibraryMissing, match="no CPU path"):
        nv.lib()


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


# ---- dmd_conv2d (conv_mfma_kernel instances; dmd_conv_f16ws.hip is not in the host build) ------------------------------------------
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
        x = (rng.standard_normal((n, hs, ws, c)) * 1.5 + 0.3).astype(np.float32)
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
        r = rng.standard_normal((n, h, w, cout)).astype(np.float32)
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
    out = np.full((n, cout, h, w) if nchw else (n, h, w, cout), np.nan, dtype=np.float32)
    p.out, p.out_nchw = S.ptr(out), int(nchw)
    tiles = L.dmd_conv_stat_tiles(h, w)
    stats = np.full((n, cout // 32, tiles, 2), np.nan) if case.get("stats") else None
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
]


# what the launcher reads per call: DIAMOND_WGRAD_MODE (the staged 32-pixel / prefetching kernels, split precision only) and
# DIAMOND_WGRAD_MAX_WG (fewer workgroups, each walking several tiles with its accumulators in registers)
WGRAD_PLANS = [dict(), dict(mode=2), dict(mode=3), dict(mode=2, max_wg=1), dict(mode=3, max_wg=3), dict(max_wg=2)]


@pytest.mark.parametrize("plan", WGRAD_PLANS, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()) or "default")
@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_conv2d_wgrad(case, plan, monkeypatch):
    if "mode" in plan and not case.get("precision"):
        pytest.skip("the staged modes are split-precision kernels")
    if "mode" in plan:
        monkeypatch.setenv("DIAMOND_WGRAD_MODE", str(plan["mode"]))
    if "max_wg" in plan:
        monkeypatch.setenv("DIAMOND_WGRAD_MAX_WG", str(plan["max_wg"]))
    rng = np.random.default_rng(11)
    L = S.lib()
    n, h, w, cin, cout, k = case["n"], case["h"], case["w"], case["cin"], case["cout"], case["k"]
    cin_real, prol = case.get("cin_real", cin), case.get("prologue", 0)
    x = (rng.standard_normal((n, h, w, cin)) * 1.2 - 0.2).astype(np.float32)
    x[..., cin_real:] = 0
    dy = rng.standard_normal((n, h, w, cout)).astype(np.float32)
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
    ws = np.full(L.dmd_wgrad_workspace_floats(p), np.nan, dtype=np.float32)
    dw = np.full((cout, cin_real, k, k), np.nan, dtype=np.float32)
    db = None if case.get("no_bias") else np.full(cout, np.nan, dtype=np.float32)
    p.workspace, p.dw, p.dbias = S.ptr(ws), S.ptr(dw), S.ptr(db)
    S.check(L.dmd_conv2d_wgrad(p, None), "dmd_conv2d_wgrad")

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
    ws = np.full(L.dmd_attention_bwd_workspace_floats(n, t, c), np.nan, dtype=np.float32)
    dqkv = np.full_like(qkv, np.nan)
    S.check(L.dmd_attention_bwd(S.ptr(qkv), S.ptr(y), S.ptr(dy), S.ptr(dqkv), S.ptr(ws), n, t, c, 8, None), "dmd_attention_bwd")
    g = dy.astype(np.float64).reshape(n, t, c // 8, 8).transpose(0, 2, 1, 3)
    dp = g @ v.transpose(0, 1, 3, 2)
    ds = p * (dp - (dp * p).sum(axis=-1, keepdims=True))
    dq, dk, dv = ds @ k / np.sqrt(8.0), ds.transpose(0, 1, 3, 2) @ q / np.sqrt(8.0), p.transpose(0, 1, 3, 2) @ g
    want = np.concatenate([a.transpose(0, 2, 1, 3).reshape(n, t, c) for a in (dq, dk, dv)], axis=-1)
    assert np.abs(dqkv - want).max() <= 1e-5 * np.abs(want).max()
