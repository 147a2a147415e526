"""Edges of the split-fp16 (DMD_PRECISION_F16X2) arithmetic on a real MI355X.

Contract under test (dmd_conv_f16ws.hip header, include/diamond_hip.h):
  * operands are represented with error <= max(2^-22 |x|, 2^-25); the fp16 range (|x| < 65520) is usable to its end;
  * nothing is clamped: a finite operand beyond the range makes every output it touches NaN (loud), all other
    outputs are unaffected; NaN / Inf inputs give non-finite outputs exactly where F.conv2d's are non-finite;
  * operands below 2^-25 are flushed: an ABSOLUTE error floor, which is why small-scale tensors (gradients) are
    pre-scaled by a power of two by the caller (ac_native._EncoderFn.backward).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_kernels import rel_err, to_nhwc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _conv_f16x2(x_nchw, wgt, bias=None, taps=9):
    from diamond_amd import engine as E, native as nv

    cout = wgt.shape[0]
    a = E.Act(to_nhwc(x_nchw.float()).to(DEV))
    wd = wgt.float().to(DEV)
    out = E.conv2d([(a, nv.PROLOGUE_NONE, None)], nv.pack_conv_weight(wd), None if bias is None else bias.float().to(DEV), cout,
                   taps=taps, want_stats=False, w_f16=nv.pack_conv_weight_f16x2(wd))
    torch.cuda.synchronize()
    return out.t.permute(0, 3, 1, 2).cpu()


def _conv1x1_stream(x_nchw, wgt, split):
    from diamond_amd import engine as E, native as nv

    a = E.Act(to_nhwc(x_nchw.float()).to(DEV))
    wd = wgt.float().to(DEV)
    out = E.conv2d([(a, nv.PROLOGUE_NONE, None)], nv.pack_conv_weight(wd), None, wgt.shape[0], taps=1, want_stats=False,
                   fast_math=split)
    torch.cuda.synchronize()
    return out.t.permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("cout,cin", [(64, 64), (32, 32), (64, 128)])
def test_f16x2_full_fp16_range(cout, cin):
    """|x| up to 6.5e4 (the end of the fp16 range): still fp32-class accuracy."""
    g = torch.Generator().manual_seed(cout + cin)
    x = (torch.randn(2, cin, 16, 16, generator=g, dtype=torch.float64) * 2e4).clamp(-65000, 65000)
    x[0, 0, 3, 3], x[1, 5, 8, 9] = 65504.0, -65504.0
    w = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) / math.sqrt(cin * 9)
    ref = F.conv2d(x, w, padding=1)
    got = _conv_f16x2(x, w)
    assert torch.isfinite(got).all()
    assert rel_err(got, ref) < 2e-5


def test_f16x2_out_of_range_operand_is_loud_not_saturated():
    """A finite fp32 value beyond the fp16 range: exactly the outputs whose receptive field contains it are NaN,
    every other output is unaffected (and correct)."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, 16, 16, generator=g, dtype=torch.float64)
    w = torch.randn(64, 64, 3, 3, generator=g, dtype=torch.float64) / 24
    x[1, 7, 5, 9] = 1.0e5
    ref = F.conv2d(x, w, padding=1)
    got = _conv_f16x2(x, w)
    touched = torch.zeros(2, 64, 16, 16, dtype=torch.bool)
    touched[1, :, 4:7, 8:11] = True
    assert bool((~torch.isfinite(got))[touched].all()), "an out-of-range operand produced a finite (silently wrong) output"
    assert bool(torch.isfinite(got)[~touched].all())
    assert float((got[~touched].double() - ref[~touched]).abs().max() / ref[~touched].abs().max()) < 2e-5
    # same contract on the streaming 1x1 kernel's split path
    w1 = torch.randn(64, 64, 1, 1, generator=g, dtype=torch.float64) / 8
    got1 = _conv1x1_stream(x, w1, split=True)
    t1 = torch.zeros(2, 64, 16, 16, dtype=torch.bool)
    t1[1, :, 5, 9] = True
    assert bool((~torch.isfinite(got1))[t1].all()) and bool(torch.isfinite(got1)[~t1].all())


def test_f16x2_nan_inf_propagate_like_conv2d():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 64, 16, 16, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    x[0, 3, 0, 0] = float("nan")
    x[1, 60, 15, 7] = float("inf")
    x[2, 31, 8, 8] = float("-inf")
    ref = F.conv2d(x, w, padding=1)
    got = _conv_f16x2(x.double(), w.double())
    assert torch.equal(torch.isfinite(got), torch.isfinite(ref)), "non-finite outputs are not where F.conv2d's are"
    fin = torch.isfinite(ref)
    assert float((got[fin] - ref[fin]).abs().max() / ref[fin].abs().max()) < 2e-5


def test_f16x2_mixed_dynamic_range_inside_k_slices():
    """Six decades of magnitude inside every 16-channel K slice: error stays at 2e-5 of the output scale, per image."""
    g = torch.Generator().manual_seed(3)
    mag = 10.0 ** (torch.rand(2, 64, 16, 16, generator=g, dtype=torch.float64) * 6 - 3)  # 1e-3 .. 1e3
    x = mag * torch.sign(torch.randn(2, 64, 16, 16, generator=g, dtype=torch.float64))
    x[1] *= 1e-3  # second image: 1e-6 .. 1 (its own, much smaller, output scale)
    w = torch.randn(64, 64, 3, 3, generator=g, dtype=torch.float64) / 24
    ref = F.conv2d(x, w, padding=1)
    got = _conv_f16x2(x, w)
    for n in range(2):
        e = rel_err(got[n], ref[n])
        print(f"image {n}: output scale {float(ref[n].abs().max()):.3e}, rel err {e:.3e}")
        assert e < 2e-5, (n, e)


def test_f16x2_tiny_operands_absolute_floor():
    """|x| ~ 1e-7 is below what two fp16 pieces resolve relatively: the error obeys the documented ABSOLUTE bound
    2^-25 * sum|w| per output (plus fp32 rounding), it is not fp32-relative."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 64, 16, 16, generator=g, dtype=torch.float64) * 1e-7
    w = torch.randn(64, 64, 3, 3, generator=g, dtype=torch.float64) / 24
    ref = F.conv2d(x, w, padding=1)
    got = _conv_f16x2(x, w)
    bound = 2.0 ** -25 * F.conv2d(torch.ones_like(x), w.abs(), padding=1) + 1e-12
    assert bool(((got.double() - ref).abs() <= bound).all())
    # the same tensor pre-scaled by 2^23 (what the gradient path does) is fp32-accurate again
    got_s = _conv_f16x2(x * 2.0 ** 23, w) * 2.0 ** -23
    assert rel_err(got_s, ref) < 2e-5


def test_actor_critic_encoder_grads_tiny_upstream_gradient():
    """with loss = mean over B*T the encoder's upstream gradient is ~1e-6 and the split-fp16 dgrad convs
    would run on their absolute floor; the backward pre-scales by a power of two.  Upstream gradients of 1e-9:
    every parameter gradient still within 1e-4 of the CPU oracle's autograd."""
    import diamond_amd as D
    from diamond_amd.testing import fill_module_, synthetic_frames
    from oracle import diamond_oracle as O

    agent = D.Agent(D.default_agent_config())
    fill_module_(agent, 5)
    ac = agent.actor_critic
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in ac.state_dict().items()}
    g = torch.Generator().manual_seed(22)
    obs = synthetic_frames(g, 5, 3, 64, 64)
    wfeat = torch.randn(5, 1024, generator=g) * 1e-9
    ref = O.ac_encoder(sd, O.ActorCriticSpec(), obs).flatten(1)
    (ref * wfeat).sum().backward()
    ac = ac.to(DEV)
    feat = ac.encode(obs.to(DEV))
    (feat * wfeat.to(DEV)).sum().backward()
    for k, p in ac.named_parameters():
        if k.startswith("encoder."):
            e = rel_err(p.grad, sd[k].grad)
            assert e < 1e-4, f"{k}: {e:.3e}"
    # and an all-zero upstream gradient gives exact zeros (no inf/NaN from the scale computation)
    ac.zero_grad()
    feat = ac.encode(obs.to(DEV))
    (feat * 0.0).sum().backward()
    for k, p in ac.named_parameters():
        if k.startswith("encoder."):
            assert float(p.grad.abs().max()) == 0.0, k
