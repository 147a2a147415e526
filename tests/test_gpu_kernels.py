"""Per-kernel parity of libdiamond_hip (through the C ABI) on a real MI355X.

Truth = float64 torch-CPU evaluation of the same op on the same seeded inputs (test
infrastructure, not product).  Tolerances are written per test; integer / quantised results
are compared bit-exactly.
"""
import math

import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def gn_ref(x, groups, eps=1e-5):
    n, c, h, w = x.shape
    xg = x.reshape(n, groups, -1)
    m = xg.mean(-1, keepdim=True)
    v = (xg - m).square().mean(-1, keepdim=True)
    return ((xg - m) / torch.sqrt(v + eps)).reshape(n, c, h, w)


def make_act(x_nchw_cpu):
    """NHWC device activation with single-tile GN stats."""
    from diamond_amd import engine as E
    t = to_nhwc(x_nchw_cpu.float()).to(DEV)
    return E.gn_stats(t) if t.shape[3] % 32 == 0 else E.Act(t)


CONV_CASES = [
    # name, N, H, W (input spatial), [Cin...], Cout, taps, stride, upsample, prologue, residual, nchw
    ("c64_16x16", 2, 16, 16, [64], 64, 9, 1, False, 1, True, False),
    ("cat128_32x32", 2, 32, 32, [64, 64], 64, 9, 1, False, 1, True, False),
    ("cat128_8x8_cfgB", 3, 8, 8, [64, 64], 64, 9, 1, False, 1, True, False),
    ("down_32to16", 2, 32, 32, [64], 64, 9, 2, False, 0, False, False),
    ("down_16to8_cfgB", 3, 16, 16, [64], 64, 9, 2, False, 0, False, False),
    ("up_8to16", 2, 8, 8, [64], 64, 9, 1, True, 0, False, False),
    ("up_16to32", 1, 16, 16, [64], 64, 9, 1, True, 0, False, False),
    ("proj1x1_cat", 2, 16, 16, [64, 64], 64, 1, 1, False, 0, False, False),
    ("proj1x1_8x8", 3, 8, 8, [64, 64], 64, 1, 1, False, 0, False, False),
    ("qkv1x1_192", 2, 8, 8, [64], 192, 1, 1, False, 2, False, False),
    ("conv_in16", 2, 16, 32, [16], 64, 9, 1, False, 0, False, False),
    ("conv_out3_nchw", 2, 16, 16, [64], 3, 9, 1, False, 1, False, True),
    ("c32_wn2", 2, 16, 16, [32], 32, 9, 1, False, 1, True, False),
    ("c32_8x8_wn2", 3, 8, 8, [32], 32, 9, 1, False, 1, True, False),
    ("c32to64", 2, 16, 16, [32], 64, 9, 1, False, 1, False, False),
    ("c32_down", 2, 16, 16, [32], 32, 9, 2, False, 0, False, False),
    ("rect_24x40", 1, 24, 40, [64], 64, 9, 1, False, 1, True, False),
    ("c64_64x64", 2, 64, 64, [64], 64, 9, 1, False, 1, True, False),
    ("cat128_16x16_n5", 5, 16, 16, [64, 64], 64, 9, 1, False, 1, True, False),
    ("up_32to64", 1, 32, 32, [64], 64, 9, 1, True, 0, False, False),
    ("c64_8x8_n6_noprologue", 6, 8, 8, [64], 64, 9, 1, False, 0, False, False),
    ("c16to32_64x64", 2, 64, 64, [16], 32, 9, 1, False, 0, False, False),
    ("c32_16x16_n3_odd", 3, 16, 16, [32], 32, 9, 1, False, 1, True, False),
    ("c32_32x32", 1, 32, 32, [32], 32, 9, 1, False, 1, True, False),
    ("c32_8x8_n11", 11, 8, 8, [32], 32, 9, 1, False, 1, True, False),
    ("c64to32_16x16", 2, 16, 16, [64], 32, 9, 1, False, 0, False, False),
    ("proj1x1_cat_64x64", 2, 64, 64, [64, 64], 64, 1, 1, False, 0, False, False),
    ("skip1x1_32to64_16x16", 3, 16, 16, [32], 64, 1, 1, False, 0, False, False),
    ("proj1x1_prologue_res", 2, 16, 16, [64], 64, 1, 1, False, 1, True, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize("impl", ["mfma", "naive", "f16x2"])
def test_conv2d(case, impl):
    from diamond_amd import engine as E, native as nv

    name, n, h, w, cins, cout, taps, stride, up, prologue, use_res, nchw = case
    g = torch.Generator().manual_seed(hash(name) % 1000)
    k = 3 if taps == 9 else 1
    cin = sum(cins)
    xs = [torch.randn(n, c, h, w, generator=g, dtype=torch.float64) * 1.5 + 0.3 for c in cins]
    wgt = torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64) / math.sqrt(cin * k * k)
    bias = torch.randn(cout, generator=g, dtype=torch.float64) * 0.1
    scale = torch.randn(n, cin, generator=g, dtype=torch.float64) * 0.3
    shift = torch.randn(n, cin, generator=g, dtype=torch.float64) * 0.3
    ho, wo = (h * 2, w * 2) if up else (h // stride, w // stride)
    res = torch.randn(n, cout, ho, wo, generator=g, dtype=torch.float64) if use_res else None

    # ---- fp64 truth
    parts, c0 = [], 0
    for x, c in zip(xs, cins):
        y = x
        if prologue:
            y = gn_ref(x, max(1, c // 32)) * (1 + scale[:, c0:c0 + c, None, None]) + shift[:, c0:c0 + c, None, None]
            if prologue == 1:
                y = y * torch.sigmoid(y)
        parts.append(y)
        c0 += c
    xin = torch.cat(parts, 1)
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, wgt, bias, stride=stride, padding=1 if k == 3 else 0)
    if res is not None:
        ref = ref + res

    # ---- device
    mul = scale.float().to(DEV).contiguous()
    add = shift.float().to(DEV).contiguous()
    srcs, c0 = [], 0
    for x, c in zip(xs, cins):
        a = make_act(x)
        spec = None
        if prologue:
            spec = E.NormSpec(mul=mul[:, c0:], add=add[:, c0:], mul_stride=cin, add_stride=cin, plus_one=True)
        srcs.append((a, prologue, spec))
        c0 += c
    wp = nv.pack_conv_weight(wgt.float().to(DEV))
    bp = nv.pad_vector(bias.float().to(DEV), nv.cout_pad(cout))
    r_act = E.Act(to_nhwc(res.float()).to(DEV)) if res is not None else None
    want_stats = (cout % 32 == 0) and not nchw
    w16 = None
    if impl == "f16x2":
        # shapes conv_f16ws_kernel covers take its packed split weights; the others (stride 2, other channel counts, NCHW
        # heads) run conv_mfma_kernel's SPLIT instance, selected by the precision flag alone
        if (stride == 1 and not nchw and not (taps == 1 and up) and
                ((cout == 64 and cin <= 128) or (cout == 32 and cin <= 64))):
            w16 = nv.pack_conv_weight_f16x2(wgt.float().to(DEV))
    out = E.conv2d(srcs, wp, bp, cout, taps=taps, stride=stride, upsample=up, residual=r_act, want_stats=want_stats,
                   out_nchw=nchw, naive=(impl == "naive"), w_f16=w16, fast_math=(impl == "f16x2"))
    torch.cuda.synchronize()
    got = out.t if nchw else out.t.permute(0, 3, 1, 2)
    err = rel_err(got, ref)
    print(f"{name}/{impl}: rel err {err:.3e}")
    assert err < 2e-5, f"{name}/{impl}: rel err {err:.3e}"
    if want_stats:
        st = out.stats.cpu().sum(dim=2)  # (N, G, 2)
        refg = ref.reshape(n, cout // 32, -1)
        assert rel_err(st[..., 0], refg.sum(-1)) < 1e-4 or float((st[..., 0] - refg.sum(-1)).abs().max()) < 1e-2
        assert rel_err(st[..., 1], refg.square().sum(-1)) < 2e-5


@pytest.mark.parametrize("n,h,w", [(2, 16, 16), (3, 64, 64), (5, 8, 8)])
def test_conv_head_nchw_f16x2(n, h, w):
    """conv_out (64 -> 3, GroupNorm + SiLU prologue, NCHW output) on the 32-cout split-fp16 instance."""
    from diamond_amd import engine as E, native as nv

    g = torch.Generator().manual_seed(n * 31 + h)
    x = torch.randn(n, 64, h, w, generator=g, dtype=torch.float64) * 1.5 + 0.3
    wgt = torch.randn(3, 64, 3, 3, generator=g, dtype=torch.float64) / 24
    bias = torch.randn(3, generator=g, dtype=torch.float64) * 0.1
    gamma = torch.randn(64, generator=g, dtype=torch.float64) * 0.2 + 1
    beta = torch.randn(64, generator=g, dtype=torch.float64) * 0.2
    a = gn_ref(x, 2) * gamma[None, :, None, None] + beta[None, :, None, None]
    ref = F.conv2d(a * torch.sigmoid(a), wgt, bias, padding=1)
    wp32 = torch.zeros(32, 64, 3, 3)
    wp32[:3] = wgt.float()
    spec = E.NormSpec(mul=gamma.float().to(DEV), add=beta.float().to(DEV))
    out = E.conv2d([(make_act(x), nv.PROLOGUE_NORM_SILU, spec)], nv.pack_conv_weight(wgt.float().to(DEV), 32),
                   nv.pad_vector(bias.float().to(DEV), 32), 3, want_stats=False, out_nchw=True, cout_padded=32,
                   w_f16=nv.pack_conv_weight_f16x2(wp32.to(DEV)))
    torch.cuda.synchronize()
    assert tuple(out.t.shape) == (n, 3, h, w)
    err = rel_err(out.t, ref)
    assert err < 2e-5, err


def test_conv_residual_norm():
    """out = conv1x1(y) + GN_affine(x): the attention block's `x_normed + out_proj(y)` (blocks.py:72)."""
    from diamond_amd import engine as E, native as nv

    g = torch.Generator().manual_seed(5)
    n, c, h, w = 2, 64, 8, 8
    x = torch.randn(n, c, h, w, generator=g, dtype=torch.float64) * 2 + 1
    y = torch.randn(n, c, h, w, generator=g, dtype=torch.float64)
    wgt = torch.randn(c, c, 1, 1, generator=g, dtype=torch.float64) / 8
    bias = torch.randn(c, generator=g, dtype=torch.float64) * 0.1
    gamma = torch.randn(c, generator=g, dtype=torch.float64) * 0.2 + 1
    beta = torch.randn(c, generator=g, dtype=torch.float64) * 0.2
    ref = F.conv2d(y, wgt, bias) + gn_ref(x, 2) * gamma[None, :, None, None] + beta[None, :, None, None]
    xa = make_act(x)
    spec = E.NormSpec(mul=gamma.float().to(DEV), add=beta.float().to(DEV))
    out = E.conv2d([(E.Act(to_nhwc(y.float()).to(DEV)), 0, None)], nv.pack_conv_weight(wgt.float().to(DEV)),
                   bias.float().to(DEV), c, taps=1, residual=xa, residual_norm=spec)
    assert rel_err(out.t.permute(0, 3, 1, 2), ref) < 2e-5


@pytest.mark.parametrize("m,n,k", [(256, 7168, 256), (3, 4, 512), (16, 2048, 1024), (70, 5, 512), (256, 1, 512)])
def test_linear(m, n, k):
    from diamond_amd import engine as E

    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g, dtype=torch.float64)
    w = torch.randn(n, k, generator=g, dtype=torch.float64) / math.sqrt(k)
    b = torch.randn(n, generator=g, dtype=torch.float64)
    ref = a @ w.t() + b
    out = E.linear(a.float().to(DEV), w.float().to(DEV), b.float().to(DEV))
    assert rel_err(out, ref) < 1e-5
    out2 = E.linear(a.float().to(DEV), w.float().to(DEV), None, out=out.clone(), accumulate=True)
    assert rel_err(out2, 2 * ref - b) < 1e-5
    out3 = E.linear(a.float().to(DEV), w.float().to(DEV), b.float().to(DEV), silu=True)
    assert rel_err(out3, ref * torch.sigmoid(ref)) < 1e-5


@pytest.mark.parametrize("t,c", [(64, 64), (256, 64), (1024, 64), (64, 32), (4096, 64), (1024, 32)])
def test_attention(t, c):
    from diamond_amd import engine as E

    g = torch.Generator().manual_seed(t + c)
    n, heads = 2, c // 8
    side = int(math.sqrt(t))
    qkv = torch.randn(n, 3 * c, side, side, generator=g, dtype=torch.float64) * 1.5
    q, k, v = qkv.reshape(n, 3, heads, 8, t).permute(0, 1, 2, 4, 3).unbind(1)
    att = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(8), dim=-1)
    ref = (att @ v).transpose(2, 3).reshape(n, c, side, side)
    out = E.attention(E.Act(to_nhwc(qkv.float()).to(DEV)), c)
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 1e-5


def test_edm_pointwise_bit_exact():
    """Preconditioning scalars, uint8 quantisation and the Euler update reproduce the
    reference's fp32 op order exactly (checked against the CPU oracle's torch ops)."""
    from diamond_amd import native as nv
    from oracle import diamond_oracle as O

    g = torch.Generator().manual_seed(3)
    n, h, w = 3, 16, 24
    x = torch.randn(n, 3, h, w, generator=g) * 2
    f = torch.randn(n, 3, h, w, generator=g)
    obs = torch.rand(n, 12, h, w, generator=g) * 2 - 1
    spec = O.DenoiserSpec()
    xd, fd, obsd = x.to(DEV), f.to(DEV), obs.to(DEV)  # keep device copies alive across the launches
    for sigma in (torch.tensor(5.0 - 1.4e-6), torch.tensor(0.28308), torch.tensor(0.002), torch.tensor([0.7, 1.9, 0.05])):
        c_in, c_out, c_skip, c_noise = O.conditioners(spec, sigma)
        # the host evaluates the conditioners (Denoiser.compute_conditioners); the kernels only apply them
        cond = torch.stack([c.reshape(-1) for c in (c_in, c_out, c_skip, c_noise)], dim=1).contiguous().to(DEV)
        stride = 0 if cond.shape[0] == 1 else 4
        # pack
        packed = torch.empty(n, h, w, 16, device=DEV)
        nv.check(nv.lib().dmd_edm_pack_input(nv.fptr(xd), nv.fptr(obsd), nv.fptr(cond), stride, 0.5,
                                             nv.fptr(packed), n, 3, 12, h, w, 16, 1, 0, nv.stream()), "pack")
        ref = torch.cat((obs / 0.5, x * c_in, torch.zeros(n, 1, h, w)), 1).permute(0, 2, 3, 1)
        assert torch.equal(packed.cpu(), ref), "edm_pack_input is not bit-exact"
        # ring-indexed context: physical slot (head + t) % T holds logical frame t
        for head in (1, 3):
            ring = torch.empty(n, 4, 3, h, w)
            for t in range(4):
                ring[:, (head + t) % 4] = obs.reshape(n, 4, 3, h, w)[:, t]
            ringd = ring.to(DEV)
            packed2 = torch.empty(n, h, w, 16, device=DEV)
            nv.check(nv.lib().dmd_edm_pack_input(nv.fptr(xd), nv.fptr(ringd), nv.fptr(cond), stride, 0.5,
                                                 nv.fptr(packed2), n, 3, 12, h, w, 16, 4, head, nv.stream()), "pack ring")
            assert torch.equal(packed2.cpu(), ref), f"ring-indexed edm_pack_input (head {head}) is not bit-exact"
        # denoised
        den = torch.empty(n, 3, h, w, device=DEV)
        nv.check(nv.lib().dmd_edm_denoised(nv.fptr(xd), nv.fptr(fd), nv.fptr(cond), stride, nv.fptr(den), n,
                                           3 * h * w, nv.stream()), "denoised")
        ref_d = O.quantize_frame(c_skip * x + c_out * f)
        assert torch.equal(den.cpu(), ref_d), "edm_denoised is not bit-exact"
    # euler
    sigmas = O.build_sigmas(O.SamplerSpec())
    den = O.quantize_frame(torch.randn(n, 3, h, w, generator=g))
    out = torch.empty(n, 3, h, w, device=DEV)
    s, nx = sigmas[0], sigmas[1]
    dend = den.to(DEV)
    nv.check(nv.lib().dmd_euler_step(nv.fptr(xd), nv.fptr(dend), float(s), float(nx - s), nv.fptr(out),
                                     x.numel(), nv.stream()), "euler")
    ref_x = x + (x - den) / s * (nx - s)
    assert torch.equal(out.cpu(), ref_x), "euler_step is not bit-exact"
    # heun combine (diffusion_sampler.py:52-56), same op order
    den2 = O.quantize_frame(torch.randn(n, 3, h, w, generator=g))
    d = (x - den) / s
    d_2 = (ref_x - den2) / nx
    ref_h = x + (d + d_2) / 2 * (nx - s)
    out_h = torch.empty(n, 3, h, w, device=DEV)
    x2d, den2d = ref_x.to(DEV), den2.to(DEV)
    nv.check(nv.lib().dmd_heun_step(nv.fptr(xd), nv.fptr(dend), nv.fptr(x2d), nv.fptr(den2d), float(s), float(nx),
                                    float(nx - s), nv.fptr(out_h), x.numel(), nv.stream()), "heun")
    assert torch.equal(out_h.cpu(), ref_h), "heun_step is not bit-exact"


def test_cond_embed():
    from diamond_amd import native as nv
    from oracle import diamond_oracle as O

    g = torch.Generator().manual_seed(4)
    n = 5
    fw = torch.randn(1, 128, generator=g)
    emb = torch.randn(4, 64, generator=g)
    act = torch.randint(0, 4, (n, 4), generator=g)
    sigma = torch.rand(n, generator=g) * 5 + 0.002
    c_noise = O.conditioners(O.DenoiserSpec(), sigma)[3]
    ref = O.fourier_features(fw, c_noise) + F.embedding(act, emb).flatten(1)
    out = torch.empty(n, 256, device=DEV)
    cond = torch.zeros(n, 4)
    cond[:, 3] = c_noise
    cd, fwd, actd, embd = cond.to(DEV), fw.to(DEV), act.to(DEV), emb.to(DEV)
    nv.check(nv.lib().dmd_cond_embed(nv.fptr(cd), 4, nv.fptr(fwd), nv.ptr(actd),
                                     nv.fptr(embd), nv.fptr(out), n, 128, 4, 64, 0, 4, nv.stream()), "cond")
    assert float((out.cpu() - ref).abs().max()) < 2e-6
    # ring-indexed actions: logical step t at column (head + t) % T
    head = 2
    ring = torch.empty_like(act)
    for t in range(4):
        ring[:, (head + t) % 4] = act[:, t]
    out2 = torch.empty(n, 256, device=DEV)
    ringd = ring.to(DEV)
    nv.check(nv.lib().dmd_cond_embed(nv.fptr(cd), 4, nv.fptr(fwd), nv.ptr(ringd), nv.fptr(embd), nv.fptr(out2), n, 128, 4,
                                     64, head, 4, nv.stream()), "cond ring")
    assert torch.equal(out2, out)


def test_categorical_sample_bit_exact():
    from diamond_amd import native as nv
    from oracle import diamond_oracle as O

    g = torch.Generator().manual_seed(8)
    for a in (4, 3, 2, 18):
        logits = torch.randn(512, a, generator=g) * 2
        e = torch.empty(512, a).exponential_(1, generator=g)
        out = torch.empty(512, dtype=torch.long, device=DEV)
        ld, ed = logits.to(DEV), e.to(DEV)
        nv.check(nv.lib().dmd_categorical_sample(nv.fptr(ld), nv.fptr(ed), nv.ptr(out), 512, a, nv.stream()), "cat")
        assert torch.equal(out.cpu(), O.categorical_sample(logits, e))


def test_maxpool_lstm_pointwise():
    from diamond_amd import native as nv
    from oracle import diamond_oracle as O

    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 32, 16, 16, generator=g)
    out = torch.empty(2, 8, 8, 32, device=DEV)
    arg = torch.empty(2, 8, 8, 32, dtype=torch.uint8, device=DEV)
    stats = torch.empty(2, 1, 1, 2, dtype=torch.float64, device=DEV)
    xd = to_nhwc(x).to(DEV)
    nv.check(nv.lib().dmd_maxpool2(nv.fptr(xd), nv.fptr(out), nv.ptr(arg), nv.ptr(stats), 2, 16, 16, 32, nv.stream()), "pool")
    ref = F.max_pool2d(x, 2)
    assert torch.equal(out.cpu().permute(0, 3, 1, 2), ref)
    assert rel_err(stats.cpu()[:, 0, 0, 1], ref.double().square().sum((1, 2, 3))) < 1e-6
    # lstm cell
    gates = torch.randn(7, 2048, generator=g) * 2
    c0 = torch.randn(7, 512, generator=g)
    h = torch.empty(7, 512, device=DEV)
    c = torch.empty(7, 512, device=DEV)
    gd, cd = gates.to(DEV), c0.to(DEV)
    nv.check(nv.lib().dmd_lstm_pointwise(nv.fptr(gd), nv.fptr(cd), nv.fptr(h), nv.fptr(c), 7, 512, nv.stream()), "lstm")
    i, f, gg, o = gates.double().chunk(4, 1)
    c_ref = torch.sigmoid(f) * c0.double() + torch.sigmoid(i) * torch.tanh(gg)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    assert rel_err(c, c_ref) < 1e-6 and rel_err(h, h_ref) < 1e-6


# ---------------------------------------------------------------------------------------------
# actor-critic encoder backward kernels: truth = float64 torch-CPU autograd of the same op
# ---------------------------------------------------------------------------------------------
WGRAD_CASES = [
    # name, N, H, W, Cin (padded), cin_real, Cout, taps, prologue
    ("in16_3to32", 3, 16, 16, 16, 3, 32, 9, 0),
    ("c32_32", 5, 16, 24, 32, 32, 32, 9, 1),
    ("c32_64", 2, 16, 16, 32, 32, 64, 9, 1),
    ("c64_64_8x8_odd", 3, 8, 8, 64, 64, 64, 9, 1),
    ("skip1x1_32_64", 3, 16, 16, 32, 32, 64, 1, 0),
    ("c32_32_many_tiles", 40, 64, 64, 32, 32, 32, 9, 1),
    ("c64_64_many_tiles", 6, 32, 32, 64, 64, 64, 9, 1),  # the denoiser's shape: eight sub-tiles per image, table refreshes per image
]


# launch plans: at most DIAMOND_WGRAD_MAX_WG workgroups (default 256; 512 for the 32-output-channel 3x3 shapes) walk contiguous tile ranges; 7: many tiles per workgroup
# (accumulators carried across tiles, the next tile's loads in flight under the MFMAs), 1024: the two-pass partial reduction
@pytest.mark.parametrize("max_wg", [None, 7, 1024], ids=["plan256", "plan7", "plan1024"])
@pytest.mark.parametrize("split", [False, True], ids=["exact", "f16x2"])
@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv_wgrad(case, split, max_wg, dmd_env):
    from diamond_amd import ac_native as A, engine as E

    dmd_env(DIAMOND_WGRAD_MAX_WG=max_wg)
    name, n, h, w, cin, cin_real, cout, taps, prologue = case
    g = torch.Generator().manual_seed(len(name) * 7 + n)
    k = 3 if taps == 9 else 1
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64) * 1.3 + 0.2
    x[:, cin_real:] = 0
    gamma = torch.randn(cin, generator=g, dtype=torch.float64) * 0.2 + 1
    beta = torch.randn(cin, generator=g, dtype=torch.float64) * 0.2
    dy = torch.randn(n, cout, h, w, generator=g, dtype=torch.float64)
    wgt = torch.zeros(cout, cin_real, k, k, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    a = x
    if prologue:
        a = gn_ref(x, max(1, cin // 32)) * gamma[None, :, None, None] + beta[None, :, None, None]
        a = a * torch.sigmoid(a)
    out = F.conv2d(a[:, :cin_real], wgt, bias, padding=1 if k == 3 else 0)
    out.backward(dy)
    xa = make_act(x)
    spec = E.NormSpec(mul=gamma.float().to(DEV), add=beta.float().to(DEV)) if prologue else None
    dyd = to_nhwc(dy.float()).to(DEV)
    dw, db = A._wgrad(xa, prologue, spec, dyd, taps, cin_real, split=split)
    torch.cuda.synchronize()
    assert rel_err(dw, wgt.grad) < 2e-5, f"{name}: dW rel err {rel_err(dw, wgt.grad):.3e}"
    assert rel_err(db, bias.grad) < 2e-5, f"{name}: db rel err {rel_err(db, bias.grad):.3e}"


@pytest.mark.parametrize("max_wg", [None, 5], ids=["plan256", "plan5"])
@pytest.mark.parametrize("shape", [(7, 32, 32, 64, 64, 9), (9, 16, 24, 32, 32, 9), (5, 16, 16, 32, 64, 1)], ids=str)
def test_conv_wgrad_two_roles_same_bits(shape, max_wg, dmd_env):
    """wgrad_ps_kernel (producer / consumer waves; the split-fp16 default) against wgrad_kernel<G, true> (DIAMOND_WGRAD_PS=0) on a
    source without prologue: the same pixels at the same k of every MFMA in the same order -- the same bits on the hardware too."""
    from diamond_amd import ac_native as A

    n, h, w, cin, cout, taps = shape
    g = torch.Generator().manual_seed(n * 31 + cin)
    x = (torch.randn(n, h, w, cin, generator=g) * 1.3 + 0.2).to(DEV)
    dy = torch.randn(n, h, w, cout, generator=g).to(DEV)
    got = []
    for ps in (1, 0):
        dmd_env(DIAMOND_WGRAD_PS=ps, DIAMOND_WGRAD_MAX_WG=max_wg)
        dw, db = A._wgrad(E_act(x), 0, None, dy, taps, cin, split=True)
        torch.cuda.synchronize()
        got.append((dw.cpu(), db.cpu()))
    assert torch.isfinite(got[0][0]).all() and torch.equal(got[0][0], got[1][0])
    assert rel_err(got[0][1], got[1][1].double()) < 1e-5  # (the bias gradient is summed over pixel pairs: another fp32 order)


@pytest.mark.parametrize("shape", [(32, 64, 64, 64, 64, 9, 1), (96, 64, 64, 32, 32, 9, 1), (96, 64, 64, 16, 32, 9, 0), (64, 16, 16, 32, 64, 9, 1),
                                   (96, 64, 64, 32, 32, 9, 2)], ids=str)
def test_conv_wgrad_run_to_run(shape):
    """the producer / consumer weight gradient at launch sizes of the training step (many sub-tiles per workgroup, two workgroups
    per CU for the 32-channel shape, a new image's table every few sub-tiles), 12 runs: one answer -- a missing barrier between
    the roles, or a buffer reused a step early, would show as differing bits now and then.  (Round 6 met a third cause: with the SiLU
    of the staging compiled as straight-line code, the two-workgroups-per-CU shapes -- 32 or 16 input channels to 32 outputs, 3 x 3 --
    gave three answers in three runs, profiles/r06n_wgrad_race.txt; the second assert pins the many-workgroup plan to the
    one-workgroup-per-CU plan of the same launch.)"""
    from diamond_amd import ac_native as A, engine as E

    n, h, w, cin, cout, taps, prologue = shape
    g = torch.Generator().manual_seed(n + cin)
    x = (torch.randn(n, h, w, cin, generator=g) * 1.3 + 0.2).to(DEV)
    dy = torch.randn(n, h, w, cout, generator=g).to(DEV)
    xa = E.gn_stats(x) if prologue else E.Act(x)  # (16 channels: conv_in's raw source)
    spec = E.NormSpec(mul=(torch.randn(cin, generator=g) * 0.2 + 1).to(DEV), add=(torch.randn(cin, generator=g) * 0.2).to(DEV)) if prologue else None
    first = None
    for _ in range(12):
        dw, db = A._wgrad(xa, prologue, spec, dy, taps, cin, split=True)
        torch.cuda.synchronize()
        if first is None:
            first = (dw.clone(), db.clone())
            assert torch.isfinite(dw).all() and float(dw.abs().max()) > 0
        else:
            assert torch.equal(dw, first[0]) and torch.equal(db, first[1])
    if cout <= 32 and taps == 9:  # (dmd_wgrad_plan's shape_cap: 512 workgroups; against 256 = one per CU: another summation order only)
        from tests.conftest import reload_dmd_env
        os.environ["DIAMOND_WGRAD_MAX_WG"] = "256"
        reload_dmd_env()
        try:
            dw1, db1 = A._wgrad(xa, prologue, spec, dy, taps, cin, split=True)
            torch.cuda.synchronize()
        finally:
            del os.environ["DIAMOND_WGRAD_MAX_WG"]
            reload_dmd_env()
        assert rel_err(first[0], dw1.double()) < 2e-6 and rel_err(first[1], db1.double()) < 2e-6, (rel_err(first[0], dw1.double()), rel_err(first[1], db1.double()))


def E_act(t):
    from diamond_amd import engine as E

    return E.Act(t)


@pytest.mark.parametrize("n,c,h,w,skip", [(3, 32, 16, 16, True), (2, 64, 8, 8, False), (2, 32, 64, 64, True), (2, 64, 24, 40, True)])
def test_gn_silu_bwd(n, c, h, w, skip):
    from diamond_amd import ac_native as A, engine as E

    g = torch.Generator().manual_seed(n + c + h)
    x = (torch.randn(n, c, h, w, generator=g, dtype=torch.float64) * 1.7 + 0.4).requires_grad_(True)
    gamma = (torch.randn(c, generator=g, dtype=torch.float64) * 0.2 + 1).requires_grad_(True)
    beta = (torch.randn(c, generator=g, dtype=torch.float64) * 0.2).requires_grad_(True)
    da = torch.randn(n, c, h, w, generator=g, dtype=torch.float64)
    dskip = torch.randn(n, c, h, w, generator=g, dtype=torch.float64) if skip else None
    u = F.group_norm(x, c // 32, gamma, beta, eps=1e-5)
    a = u * torch.sigmoid(u)
    tot = (a * da).sum() + ((x * dskip).sum() if skip else 0)
    tot.backward()
    xa = make_act(x.detach())
    spec = E.NormSpec(mul=gamma.detach().float().to(DEV), add=beta.detach().float().to(DEV))
    dad = to_nhwc(da.float()).to(DEV)
    dsd = to_nhwc(dskip.float()).to(DEV) if skip else None
    dx, dmul, dadd = A._gn_silu_bwd(xa, spec, dad, dsd)
    torch.cuda.synchronize()
    assert rel_err(dx.permute(0, 3, 1, 2), x.grad) < 2e-5
    assert rel_err(dmul.sum(0), gamma.grad) < 2e-5
    assert rel_err(dadd.sum(0), beta.grad) < 2e-5


def test_maxpool_bwd():
    from diamond_amd import ac_native as A

    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 32, 16, 24, generator=g).requires_grad_(True)
    dp = torch.randn(3, 32, 8, 12, generator=g)
    F.max_pool2d(x, 2).backward(dp)
    pooled, arg = A._maxpool(to_nhwc(x.detach()).to(DEV))
    dx = A._maxpool_bwd(to_nhwc(dp).to(DEV), arg)
    assert torch.equal(dx.cpu().permute(0, 3, 1, 2), x.grad)


def test_actor_critic_encoder_grads_vs_oracle():
    """Whole encoder, forward + every parameter gradient, vs the CPU oracle under torch autograd (fp32);
    tolerance 1e-4 relative (north_star)."""
    import diamond_amd as D
    from diamond_amd.testing import fill_module_, synthetic_frames
    from oracle import diamond_oracle as O

    agent = D.Agent(D.default_agent_config())
    fill_module_(agent, 5)
    ac = agent.actor_critic
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in ac.state_dict().items()}
    g = torch.Generator().manual_seed(21)
    obs = synthetic_frames(g, 5, 3, 64, 64)
    wfeat = torch.randn(5, 1024, generator=g)
    ref = O.ac_encoder(sd, O.ActorCriticSpec(), obs).flatten(1)
    (ref * wfeat).sum().backward()
    ac = ac.to(DEV)
    feat = ac.encode(obs.to(DEV))
    (feat * wfeat.to(DEV)).sum().backward()
    assert rel_err(feat.detach(), ref.detach()) < 1e-4
    for k, p in ac.named_parameters():
        if k.startswith("encoder."):
            assert rel_err(p.grad, sd[k].grad) < 1e-4, f"{k}: {rel_err(p.grad, sd[k].grad):.3e}"
