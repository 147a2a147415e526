"""VALID EXTENT (include/diamond_hip.h, ABI v6): a tensor whose size is off the kernels' tile grid lives in a larger buffer;
conv-input positions outside the valid extent read as zero, GroupNorm counts / emitted statistics cover the valid extent only,
attention keys outside it stay out of the softmax.  Truth = float64 torch-CPU on the CROPPED tensors; the buffers' margins
are filled with large garbage so that any unmasked read shows."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_kernels import gn_ref, rel_err, to_nhwc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _embed(x_nchw, hp, wp, fill=37.0):
    """NHWC buffer (N, hp, wp, C) holding x in [:h, :w], garbage elsewhere."""
    n, c, h, w = x_nchw.shape
    buf = torch.full((n, hp, wp, c), fill)
    buf[:, :h, :w] = to_nhwc(x_nchw)
    return buf.to(DEV)


def _stats_total(act):
    """(N, G, 2) fp64 totals of an activation's partial sums."""
    return act.stats.sum(2).cpu()


def _ref_totals(y_nchw):
    n, c, h, w = y_nchw.shape
    g = y_nchw.double().reshape(n, c // 32, -1)
    return torch.stack([g.sum(-1), g.square().sum(-1)], -1)


CASES = [
    # name, cin list, cout, taps, stride, upsample, prologue, residual, valid (h, w) of the OUTPUT, buffer (H, W) of the output
    ("ws_c64_36x36_in_48", [64], 64, 9, 1, False, 1, True, (36, 36), (48, 48)),
    ("ws_cat128_18x20_in_32", [64, 64], 64, 9, 1, False, 1, False, (18, 20), (32, 32)),
    ("ws_up_9x10_to_18x20", [64], 64, 9, 1, True, 0, False, (18, 20), (32, 32)),
    ("ws_c32_72x72_in_80_b8", [32], 32, 9, 1, False, 1, True, (72, 72), (80, 88)),
    ("ws_convin_c16_68x76_in_128", [16], 64, 9, 1, False, 0, False, (68, 76), (128, 128)),
    ("mfma_down_72_to_36", [64], 64, 9, 2, False, 0, False, (36, 36), (64, 64)),
    ("mfma_qkv_1x1_9x10_in_16", [64], 192, 1, 1, False, 2, False, (9, 10), (16, 16)),
]


@pytest.mark.parametrize("name,cins,cout,taps,stride,up,prologue,res,valid,buf", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("precision", ["f16x2", "f32"])
def test_conv_valid_extent(name, cins, cout, taps, stride, up, prologue, res, valid, buf, precision):
    from diamond_amd import engine as E, native as nv

    g = torch.Generator().manual_seed(sum(map(ord, name)))
    n = 3
    vh, vw = valid
    sh, sw = (vh // 2, vw // 2) if up else (vh * stride, vw * stride)      # source valid extent
    bh, bw = (buf[0] // 2, buf[1] // 2) if up else (buf[0] * stride, buf[1] * stride)  # source buffer
    cin = sum(cins)
    xs = [torch.randn(n, c, sh, sw, generator=g) * 1.5 + 0.2 for c in cins]
    mul = torch.randn(n, cin, generator=g) * 0.3
    add = torch.randn(n, cin, generator=g) * 0.3
    k = 3 if taps == 9 else 1
    wgt = torch.randn(cout, cin, k, k, generator=g) / (cin * taps) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    r = torch.randn(n, cout, vh, vw, generator=g) if res else None
    # ---- truth on the cropped tensors
    parts, c0 = [], 0
    for x in xs:
        c = x.shape[1]
        if prologue:
            xn = gn_ref(x.double(), c // 32)
            xn = xn * (1 + mul[:, c0:c0 + c].double().view(n, c, 1, 1)) + add[:, c0:c0 + c].double().view(n, c, 1, 1)
            xn = F.silu(xn) if prologue == 1 else xn
        else:
            xn = x.double()
        parts.append(xn)
        c0 += c
    xin = torch.cat(parts, 1)
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, wgt.double(), bias.double(), stride=stride, padding=k // 2)
    if res:
        ref = ref + r.double()
    assert tuple(ref.shape[2:]) == (vh, vw)
    # ---- the launch
    srcs, c0 = [], 0
    for x in xs:
        c = x.shape[1]
        t = _embed(x, bh, bw)
        a = E.gn_stats(t, (sh, sw)) if (prologue and c % 32 == 0) else E.Act(t, valid=(sh, sw))
        spec = E.NormSpec(mul=mul[:, c0:].contiguous().to(DEV), add=add[:, c0:].contiguous().to(DEV), mul_stride=cin - c0, add_stride=cin - c0,
                          plus_one=True) if prologue else None
        srcs.append((a, {0: nv.PROLOGUE_NONE, 1: nv.PROLOGUE_NORM_SILU, 2: nv.PROLOGUE_NORM}[prologue], spec))
        c0 += c
    r_act = E.Act(_embed(r, buf[0], buf[1]), valid=valid) if res else None
    w16 = nv.pack_conv_weight_f16x2(wgt.to(DEV)) if (precision == "f16x2" and cout in (32, 64) and stride == 1) else None
    out = E.conv2d(srcs, nv.pack_conv_weight(wgt.to(DEV)), nv.pad_vector(bias.to(DEV), nv.cout_pad(cout)), cout, taps=taps, stride=stride,
                   upsample=up, residual=r_act, want_stats=cout % 32 == 0, w_f16=w16, fast_math=precision == "f16x2")
    assert out.valid == valid and tuple(out.shape[1:3]) == tuple(buf)
    got = out.t[:, :vh, :vw].permute(0, 3, 1, 2)
    e = rel_err(got, ref)
    assert e < 2e-5, f"{name} [{precision}]: {e:.3e}"
    if out.stats is not None:
        tot, want = _stats_total(out), _ref_totals(ref)
        d0 = float((tot[..., 0] - want[..., 0]).abs().max() / want[..., 1].sqrt().max())
        d1 = float(((tot[..., 1] - want[..., 1]).abs() / want[..., 1]).max())
        assert d0 < 1e-4 and d1 < 1e-5, f"{name} [{precision}]: statistics over the valid extent: sum {d0:.3e}, sum of squares {d1:.3e}"


def test_attention_and_gn_stats_valid_extent():
    from diamond_amd import engine as E

    g = torch.Generator().manual_seed(5)
    n, c, hp, wp, vh, vw = 2, 64, 16, 16, 9, 10
    qkv = torch.randn(n, 3 * c, vh, vw, generator=g)
    y = E.attention(E.Act(_embed(qkv, hp, wp, fill=9.0), valid=(vh, vw)), c)
    q, k, v = (t.reshape(n, c // 8, 8, vh * vw).transpose(2, 3).double() for t in qkv.chunk(3, 1))
    ref = torch.softmax(q @ k.transpose(2, 3) / 8 ** 0.5, -1) @ v          # (n, heads, T, 8)
    ref = ref.transpose(2, 3).reshape(n, c, vh, vw)
    assert rel_err(y[:, :vh, :vw].permute(0, 3, 1, 2), ref) < 2e-5
    x = torch.randn(n, c, vh, vw, generator=g)
    a = E.gn_stats(_embed(x, hp, wp), (vh, vw))
    assert float((_stats_total(a) - _ref_totals(x)).abs().max()) < 1e-6


# ---- the backward kernels of the actor-critic encoder (ABI v8): dmd_conv2d_wgrad, dmd_gn_silu_bwd, max-pool --------------------
@pytest.mark.parametrize("split", [False, True], ids=["exact", "f16x2"])
@pytest.mark.parametrize("cin,cout,taps,valid,buf", [(32, 32, 9, (36, 36), (40, 40)), (32, 64, 9, (18, 18), (24, 24)),
                                                     (32, 64, 1, (18, 18), (24, 24)), (64, 64, 9, (9, 9), (16, 16)), (16, 32, 9, (72, 72), (80, 80))])
def test_wgrad_valid_extent(cin, cout, taps, valid, buf, split):
    """dW / db of a convolution whose input and output exist on the valid extent only: margins of x AND of dy are garbage"""
    from diamond_amd import ac_native as A, engine as E

    g = torch.Generator().manual_seed(cin + cout + taps + valid[0])
    n, (vh, vw), k = 3, valid, 3 if taps == 9 else 1
    prologue = 1 if cin % 32 == 0 else 0
    x = torch.randn(n, cin, vh, vw, generator=g, dtype=torch.float64) * 1.3 + 0.2
    gamma = torch.randn(cin, generator=g, dtype=torch.float64) * 0.2 + 1
    beta = torch.randn(cin, generator=g, dtype=torch.float64) * 0.2
    dy = torch.randn(n, cout, vh, vw, generator=g, dtype=torch.float64)
    wgt = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    a = x
    if prologue:
        a = gn_ref(x, cin // 32) * gamma[None, :, None, None] + beta[None, :, None, None]
        a = a * torch.sigmoid(a)
    F.conv2d(a, wgt, bias, padding=1 if k == 3 else 0).backward(dy)
    xa = E.gn_stats(_embed(x.float(), *buf), valid) if prologue else E.Act(_embed(x.float(), *buf), valid=valid)
    spec = E.NormSpec(mul=gamma.float().to(DEV), add=beta.float().to(DEV)) if prologue else None
    dw, db = A._wgrad(xa, prologue, spec, _embed(dy.float(), *buf, fill=-53.0), taps, cin, split=split)
    assert rel_err(dw, wgt.grad) < 2e-5 and rel_err(db, bias.grad) < 2e-5, (rel_err(dw, wgt.grad), rel_err(db, bias.grad))


@pytest.mark.parametrize("c,valid,buf,skip", [(32, (36, 36), (40, 40), True), (64, (9, 9), (16, 16), False), (32, (18, 20), (24, 24), True)])
def test_gn_silu_bwd_valid_extent(c, valid, buf, skip):
    """sums, count and dx over the valid extent; dx is ZERO outside it (what the max-pool backward / next wgrad rely on)"""
    from diamond_amd import ac_native as A, engine as E

    g = torch.Generator().manual_seed(c + valid[0])
    n, (vh, vw) = 2, valid
    x = (torch.randn(n, c, vh, vw, generator=g, dtype=torch.float64) * 1.7 + 0.4).requires_grad_(True)
    gamma = (torch.randn(c, generator=g, dtype=torch.float64) * 0.2 + 1).requires_grad_(True)
    beta = (torch.randn(c, generator=g, dtype=torch.float64) * 0.2).requires_grad_(True)
    da = torch.randn(n, c, vh, vw, generator=g, dtype=torch.float64)
    dskip = torch.randn(n, c, vh, vw, generator=g, dtype=torch.float64) if skip else None
    u = F.group_norm(x, c // 32, gamma, beta, eps=1e-5)
    tot = ((u * torch.sigmoid(u)) * da).sum() + ((x * dskip).sum() if skip else 0)
    tot.backward()
    xa = E.gn_stats(_embed(x.detach().float(), *buf), valid)
    spec = E.NormSpec(mul=gamma.detach().float().to(DEV), add=beta.detach().float().to(DEV))
    dx, dmul, dadd = A._gn_silu_bwd(xa, spec, _embed(da.float(), *buf, fill=91.0), _embed(dskip.float(), *buf, fill=-17.0) if skip else None)
    assert rel_err(dx[:, :vh, :vw].permute(0, 3, 1, 2), x.grad) < 2e-5
    assert rel_err(dmul.sum(0), gamma.grad) < 2e-5 and rel_err(dadd.sum(0), beta.grad) < 2e-5
    outside = dx.clone()
    outside[:, :vh, :vw] = 0
    assert float(outside.abs().max()) == 0.0, "dx outside the valid extent must be exactly zero"


def test_maxpool_valid_extent_floors_like_the_reference():
    """MaxPool2d(2) of a 9x9 valid extent in a 16x16 buffer = F.max_pool2d's 4x4 (floor), statistics over those 16 pixels"""
    from diamond_amd import ac_native as A

    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 9, 9, generator=g)
    act, arg = A._maxpool(_embed(x, 16, 16, fill=1e4), (9, 9))
    ref = F.max_pool2d(x, 2)
    assert act.valid == (4, 4) and tuple(act.t.shape) == (2, 8, 8, 32)
    assert torch.equal(act.t[:, :4, :4].cpu().permute(0, 3, 1, 2), ref)
    assert torch.allclose(_stats_total(act), _ref_totals(ref), rtol=1e-12, atol=1e-9)
