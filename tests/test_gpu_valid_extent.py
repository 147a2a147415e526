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
