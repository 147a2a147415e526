"""Would Winograd F(2x2, 3x3) fit the parity budget?  (CPU experiment, no GPU; SURVEY.md 8(d) names it as the way to execute fewer
FLOPs than algorithmic.)

The 3x3 stride-1 convolutions of the denoiser (93 % of the window's FLOPs) run on conv_f16ws_kernel as split-fp32 products on the
f16 matrix cores: x = xh + xl, w = wh + wl, three MFMAs per product (wh xh + wh xl + wl xh), fp32 accumulation.  Winograd F(2,3)
would execute 2.25x fewer multiplications, in a transformed domain: V = B^T d B of 4x4 input tiles, U = G g G^T of the weights,
M = sum_c U * V, Y = A^T M A.  Its operands V are sums of four activations and its outputs are differences of nine M's, so the
2^-22 relative error of the split operands is amplified.  The budget it must fit: quantised frames on the reference's uint8 level
except for <= 1e-4 of the pixels (tests/test_gpu_models.py, tests/test_oracle_golden.py: check_quantised), and actions / rewards /
ends bit-exact over the golden windows.

This script runs the ORACLE's denoiser (oracle/diamond_oracle.py: test infrastructure, used here as the lab bench it is) with
its conv replaced by emulations of the three arithmetic schemes, on the bench's synthetic weights, and reports for each: the
error of the network output against an fp64 run, and the fraction of quantised pixels that land on another uint8 level than (a)
the fp64 run's, (b) the fp32 reference arithmetic's.

    python tools/winograd_numerics.py [batch]  -> one JSON object (profiles/r04_winograd_numerics.json, "denoiser")
    python tools/winograd_numerics.py --window -> the golden two-window rollout (tests/golden/window.pt: the reference's own
                                                  actions / rewards / ends) under each arithmetic: are the integers still exact?
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import diamond_oracle as O  # noqa: E402
from tests.conftest import make_oracle_agent  # noqa: E402
from diamond_amd.testing import synthetic_actions, synthetic_frames  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def split(x):
    h = x.half().float()
    return h, (x - h).half().float()


def conv_split(x, w, b, terms=3):
    """the shipped arithmetic: f16 x f16 products are exact in fp32, accumulation in fp32"""
    xh, xl = split(x)
    wh, wl = split(w)
    y = F.conv2d(xh, wh, None, padding=1) + F.conv2d(xl, wh, None, padding=1) + F.conv2d(xh, wl, None, padding=1)
    if terms == 4:
        y = y + F.conv2d(xl, wl, None, padding=1)
    return y if b is None else y + b.view(1, -1, 1, 1)


def conv_winograd(x, w, b, mode):
    """mode 'f32': transforms and products in fp32 (what a fp32 Winograd does); 'split': V and U split into f16 pairs, three
    products; 'split4': four products (ul vl kept)"""
    n, c, h, wd = x.shape
    o = w.shape[0]
    d = F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)  # n c th tw 4 4
    v = torch.einsum("ij,ncxyjk,lk->ncxyil", BT, d, BT)
    u = torch.einsum("ij,ocjk,lk->ocil", G, w.double(), G).float()
    if mode == "f32":
        m = torch.einsum("ocil,ncxyil->noxyil", u, v)
    else:
        vh, vl = split(v)
        uh, ul = split(u)
        m = torch.einsum("ocil,ncxyil->noxyil", uh, vh) + torch.einsum("ocil,ncxyil->noxyil", uh, vl) \
            + torch.einsum("ocil,ncxyil->noxyil", ul, vh)
        if mode == "split4":
            m = m + torch.einsum("ocil,ncxyil->noxyil", ul, vl)
    y = torch.einsum("ij,noxyjk,lk->noxyil", AT, m, AT)  # n o th tw 2 2
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(n, o, h, wd)
    return y if b is None else y + b.view(1, -1, 1, 1)


def make_conv(scheme):
    def conv(sd, p, x, stride=1, padding=1):
        w, b = sd[p + ".weight"], sd.get(p + ".bias")
        big = w.shape[-1] == 3 and stride == 1 and w.shape[1] >= 32 and x.dtype == torch.float32
        if scheme == "direct" or not big:
            if scheme != "direct" and w.shape[-1] == 3 and x.dtype == torch.float32:  # stride-2 / conv_in / conv_out: shipped arithmetic
                xh, xl = split(x)
                wh, wl = split(w)
                k = dict(stride=stride, padding=padding)
                y = F.conv2d(xh, wh, None, **k) + F.conv2d(xl, wh, None, **k) + F.conv2d(xh, wl, None, **k)
                return y if b is None else y + b.view(1, -1, 1, 1)
            return F.conv2d(x, w, b, stride=stride, padding=padding)
        if scheme == "split":
            return conv_split(x, w, b)
        if x.shape[-1] % 2 or x.shape[-2] % 2 or x.shape[-1] < WINO_MIN:
            return conv_split(x, w, b)
        return conv_winograd(x, w, b, scheme[len("wino_"):])
    return conv


WINO_MIN = 32  # Winograd on the 64x64 and 32x32 levels (96 % of the 3x3 FLOPs), the shipped arithmetic below


def window():
    from tests.conftest import load_golden
    from diamond_amd.testing import initial_condition_batches

    gold = load_golden("window.pt")
    orig = O.conv
    out = {}
    for scheme in ("direct", "split", "wino_split"):
        O.conv = orig if scheme == "direct" else make_conv(scheme)
        a = make_oracle_agent()
        b, t = gold["b"], gold["backup_every"]
        draws = O.DrawSource(torch.Generator().manual_seed(gold["rng_seed"]))
        torch.manual_seed(gold["rng_seed"])
        draws.g = torch.default_generator
        env = O.ImaginationEnv(a, initial_condition_batches(gold["pool_seed"], b, 4), b, gold["horizon"], draws,
                               num_batches_to_preload=gold["preload"])
        state = (env.reset(), torch.zeros(b, 512), torch.zeros(b, 512))
        rows = []
        with torch.no_grad():
            for w in gold["windows"]:
                (obs, act, rew, end, trunc, logits, val, vb), state = O.rollout(a, env, state, t, draws)
                q = obs.add(1).div(2).mul(255).round().to(torch.int32)
                rows.append({"act": bool(torch.equal(act, w["act"])), "rew": bool(torch.equal(rew, w["rew"])),
                             "end": bool(torch.equal(end, w["end"])), "trunc": bool(torch.equal(trunc, w["trunc"])),
                             "pixels_off_reference_level": float((q != w["obs_u8"].int()).double().mean()),
                             "frames": int(obs.shape[0] * obs.shape[1])})
        out[scheme] = rows
        print(scheme, rows, file=sys.stderr)
    O.conv = orig
    print(json.dumps({"golden": "tests/golden/window.pt (free-running: a flipped pixel feeds the next frames)", "schemes": out}, indent=1))


def main():
    if "--window" in sys.argv:
        return window()
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.set_num_threads(os.cpu_count())
    a = make_oracle_agent()
    g = torch.Generator().manual_seed(123)
    obs = synthetic_frames(g, b, 12, 64, 64)
    act = synthetic_actions(g, 4, b, 4)
    noise = torch.randn(b, 3, 64, 64, generator=g)
    sig = O.build_sigmas(a.sspec)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in a.denoiser.items()}
    orig = O.conv
    out = {"batch": b, "levels_on_winograd": f">= {WINO_MIN}x{WINO_MIN}", "sigmas": [float(s) for s in sig[:-1]], "schemes": {}}
    runs = {}
    for scheme in ("fp64", "direct", "split", "wino_f32", "wino_split", "wino_split4"):
        res = []
        for sigma in sig[:-1]:
            x = noise * sigma + obs[:, -3:] * 0.5
            if scheme == "fp64":
                O.conv = orig
                d, f = O.denoise(sd64, a.dspec, x.double(), sigma.double(), obs.double(), act, return_model_output=True)
            else:
                O.conv = make_conv(scheme)
                d, f = O.denoise(a.denoiser, a.dspec, x, sigma, obs, act, return_model_output=True)
            res.append((d.double(), f.double()))
        runs[scheme] = res
        print(scheme, "done", file=sys.stderr)
    O.conv = orig

    def u8(d):
        return d.add(1).div(2).mul(255).round().to(torch.int32)

    for scheme, res in runs.items():
        if scheme == "fp64":
            continue
        rows = []
        for i, (d, f) in enumerate(res):
            d64, f64 = runs["fp64"][i]
            dd, fd = runs["direct"][i]
            rows.append({
                "model_output_err_vs_fp64": float((f - f64).abs().max() / f64.abs().max()),
                "model_output_rms_err_vs_fp64": float((f - f64).pow(2).mean().sqrt() / f64.pow(2).mean().sqrt()),
                "pixels_off_fp64_level": float((u8(d) != u8(d64)).double().mean()),
                "pixels_off_fp32_reference_level": float((u8(d) != u8(dd)).double().mean()),
            })
        out["schemes"][scheme] = {k: [r[k] for r in rows] for k in rows[0]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
