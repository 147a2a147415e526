"""The fused skip projection (dmd_conv_f16ws.hip: PROJECTION; ResBlock.forward, /root/reference/src/models/blocks.py:133,147):
`proj(cat(x, skip)) + conv2(silu(norm2(h)))` as ONE launch, against
  (1) the same two operations as separate launches (1x1 stream kernel -> residual of conv2): same split-fp16 products,
      only the fp32 summation order differs -> 2e-6 of the output scale, GroupNorm statistics 1e-6 relative;
  (2) the float64 torch-CPU evaluation on sampled images (2e-5 relative);
  (3) an N=2 launch of sampled images, BITWISE (the result of an image may not depend on where its tiles fall).
Shapes cover tiles_per_wg = 1, 2, 3 (odd), 16 on the three levels that use the kernel; the ResBlock-level test checks that
blocks.ResBlock.run takes the fused path and matches the unfused one."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_kernels import gn_ref, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = [  # name, N, H
    ("tpw1_64x64_N8", 8, 64),
    ("tpw3_64x64_N41_odd", 41, 64),
    ("tpw16_64x64_N256", 256, 64),
    ("tpw4_32x32_N256", 256, 32),
    ("tpw2_16x16_N300", 300, 16),
    ("tpw1_16x16_N3", 3, 16),
]


def _inputs(n, hw, seed):
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(n, hw, hw, 64, generator=g)
    xa = torch.randn(n, hw, hw, 64, generator=g)
    xb = torch.randn(n, hw, hw, 64, generator=g) * 1.7 + 0.3
    mul = torch.randn(n, 64, generator=g) * 0.2
    add = torch.randn(n, 64, generator=g) * 0.2
    w2 = torch.randn(64, 64, 3, 3, generator=g) / (64 * 9) ** 0.5
    b2 = torch.randn(64, generator=g) * 0.1
    wp = torch.randn(64, 128, 1, 1, generator=g) / 128 ** 0.5
    bp = torch.randn(64, generator=g) * 0.1
    return [t.to(DEV) for t in (h, xa, xb, mul, add, w2, b2, wp, bp)]


def _launch(E, nv, h, xa, xb, mul, add, w2, b2, wp, bp, fused):
    a = E.gn_stats(h)
    spec = E.NormSpec(mul=mul, add=add, mul_stride=64, add_stride=64, plus_one=True)
    w2p, w2h = nv.pack_conv_weight(w2), nv.pack_conv_weight_f16x2(w2)
    wph = nv.pack_conv_weight_f16x2(wp)
    if fused:
        return E.conv2d([(a, nv.PROLOGUE_NORM_SILU, spec)], w2p, b2, 64, w_f16=w2h, proj=([E.Act(xa), E.Act(xb)], wph, bp))
    r = E.conv2d([(E.Act(xa), nv.PROLOGUE_NONE, None), (E.Act(xb), nv.PROLOGUE_NONE, None)], nv.pack_conv_weight(wp), bp, 64,
                 taps=1, want_stats=False, w_f16=wph)
    return E.conv2d([(a, nv.PROLOGUE_NORM_SILU, spec)], w2p, b2, 64, w_f16=w2h, residual=r)


@pytest.mark.parametrize("name,n,hw", CASES, ids=[c[0] for c in CASES])
def test_fused_projection_vs_separate_launches_fp64_and_small_batch(name, n, hw):
    from diamond_amd import engine as E, native as nv

    args = _inputs(n, hw, 1234 + n)
    h, xa, xb, mul, add, w2, b2, wp, bp = args
    fused = _launch(E, nv, *args, fused=True)
    assert "WsGeomProj" in E.kernel_key(_params_of(E, nv, *args)), "the fused launch must run the projection geometry"
    sep = _launch(E, nv, *args, fused=False)
    scale = float(sep.t.abs().max())
    d = float((fused.t - sep.t).abs().max()) / scale
    assert d < 2e-6, f"{name}: fused vs separate launches {d:.3e}"
    # partial sums of a tile (2048 or 4096 values): compared on the scale of the tile's sum of squares (a sum can cancel to ~0)
    sq = sep.stats[..., 1:2].abs().clamp_min(1.0)
    ds = float(((fused.stats - sep.stats).abs() / torch.cat([sq.sqrt() * 64, sq], dim=-1)).max())
    assert ds < 1e-6, f"{name}: GroupNorm partial sums {ds:.3e}"
    # sampled images: fp64 reference and the bitwise small-batch launch
    pick = sorted({0, n // 2, n - 1})[:2] if n > 1 else [0]
    idx = torch.tensor(pick, device=DEV)
    small = _launch(E, nv, h[idx].contiguous(), xa[idx].contiguous(), xb[idx].contiguous(), mul[idx].contiguous(),
                    add[idx].contiguous(), w2, b2, wp, bp, fused=True)
    assert torch.equal(small.t, fused.t[idx]), f"{name}: an image's result depends on its position in the tile walk"
    for k, i in enumerate(pick):
        hn = gn_ref(h[i:i + 1].cpu().double().permute(0, 3, 1, 2), 2)
        hn = F.silu(hn * (1 + mul[i].cpu().double().view(1, 64, 1, 1)) + add[i].cpu().double().view(1, 64, 1, 1))
        ref = F.conv2d(hn, w2.cpu().double(), b2.cpu().double(), padding=1)
        xcat = torch.cat([xa[i:i + 1], xb[i:i + 1]], dim=3).cpu().double().permute(0, 3, 1, 2)
        ref = ref + F.conv2d(xcat, wp.cpu().double(), bp.cpu().double())
        e = rel_err(fused.t[i:i + 1].cpu().permute(0, 3, 1, 2), ref)
        assert e < 2e-5, f"{name}: image {i} vs fp64 {e:.3e}"


def _params_of(E, nv, h, xa, xb, mul, add, w2, b2, wp, bp):
    """The dmd_conv_params of the fused launch (for dmd_conv2d_kernel_name)."""
    import ctypes as C

    p = nv.ConvParams()
    n, hh, ww, _ = h.shape
    p.N, p.H, p.W, p.Cout, p.CoutPad, p.taps, p.stride, p.nsrc = n, hh, ww, 64, 64, 9, 1, 1
    a = E.gn_stats(h)
    p.src[0].x, p.src[0].C, p.src[0].prologue = nv.ptr(h), 64, nv.PROLOGUE_NORM_SILU
    p.src[0].norm = E.NormSpec(mul=mul, add=add, mul_stride=64, add_stride=64, plus_one=True).to_native(a)
    keep = (nv.pack_conv_weight(w2), nv.pack_conv_weight_f16x2(w2), nv.pack_conv_weight_f16x2(wp), torch.empty_like(h))
    p.w, p.w_f16, p.proj_w_f16, p.out = nv.ptr(keep[0]), nv.ptr(keep[1]), nv.ptr(keep[2]), nv.ptr(keep[3])
    p.precision = nv.PRECISION_F16X2
    p.proj_nsrc = 2
    p.proj_x[0], p.proj_x[1] = nv.ptr(xa), nv.ptr(xb)
    p.proj_C[0], p.proj_C[1] = 64, 64
    assert nv.lib().dmd_conv2d_proj_eligible(C.byref(p)) == 1
    p._keep = keep
    return p


def test_proj_eligibility_and_loud_failure():
    import ctypes as C

    from diamond_amd import engine as E, native as nv

    args = _inputs(2, 16, 7)
    p = _params_of(E, nv, *args)
    lib = nv.lib()
    p.proj_C[1] = 32
    assert lib.dmd_conv2d_proj_eligible(C.byref(p)) == 0
    assert lib.dmd_conv2d(C.byref(p), nv.stream()) != 0 and b"projection" in lib.dmd_last_error()
    p.proj_C[1] = 64
    p.src[0].C = 32  # 2 chunk steps per tile: no fused projection
    assert lib.dmd_conv2d_proj_eligible(C.byref(p)) == 0
    p.src[0].C = 64
    p.H = p.W = 8
    assert lib.dmd_conv2d_proj_eligible(C.byref(p)) == 0
    p.H = p.W = 16
    assert lib.dmd_conv2d_naive(C.byref(p), nv.stream()) != 0, "the naive path must refuse a fused projection"
    torch.cuda.synchronize()


def test_resblock_takes_the_fused_path_and_matches_the_unfused_one(monkeypatch):
    import diamond_amd as D
    from diamond_amd import blocks as BL, engine as E, native as nv
    from diamond_amd.testing import fill_module_

    blk = BL.ResBlock(128, 64, 256, attn=False)
    fill_module_(blk, 3)
    with torch.no_grad():
        blk.conv2.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(5))  # (zero-initialised by the constructor)
    blk = blk.to(DEV)
    g = torch.Generator().manual_seed(11)
    n = 5
    x = torch.randn(n, 32, 32, 64, generator=g).to(DEV)
    skip = torch.randn(n, 32, 32, 64, generator=g).to(DEV)
    cond = torch.randn(n, 256, generator=g).to(DEV)
    film = BL.FilmTable(blk)

    def run():
        cache = E.PackCache()
        ctx = BL.RunCtx(cache, film, film.compute(cond), precision="f16x2")
        names = []
        prof = nv.LaunchProfiler()
        monkeypatch.setattr(nv, "PROFILER", prof)
        out = blk.run(ctx, [E.gn_stats(x), E.gn_stats(skip)])
        torch.cuda.synchronize()
        monkeypatch.setattr(nv, "PROFILER", None)
        names = list(prof.summary().keys())
        return out, names

    monkeypatch.setattr(E, "FUSE_PROJ", True)
    fused, names_f = run()
    monkeypatch.setattr(E, "FUSE_PROJ", False)
    sep, names_s = run()
    assert any("WsGeomProj" in k for k in names_f) and not any("conv1x1_stream" in k for k in names_f), names_f
    assert any("conv1x1_stream" in k for k in names_s) and not any("WsGeomProj" in k for k in names_s), names_s
    d = float((fused.t - sep.t).abs().max() / sep.t.abs().max())
    assert d < 2e-6, f"ResBlock fused vs unfused {d:.3e}"
