"""gpurun_out/pmc/{FETCH,WRITE}_SIZE.json -> profiles/<tag>_pmc_traffic.json and profiles/pmc_traffic.json
(HBM bytes per launch per kernel, keyed by the rocprofv3 kernel name normalised exactly like
dmd_conv2d_kernel_name() spells it, so bench.py finds its dominant kernel by name).

    python tools/pmc_to_profile.py r02a        (after `bash tools/pmc_collect.sh` on the GPU box)

Corrections (MI355X_MICROARCH.md, HBM section; re-checked by the calibration copy of `bench.py --pmc-calibrate`):
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> x2;
WRITE_SIZE matches a known byte count 1:1.  The measured calibration factors are stored under "_calibration"."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def normalise(name: str) -> str:
    """'void conv_f16ws_kernel<WsGeom<false, 2, 9> >(dmd_conv_params, int, int)' -> 'conv_f16ws_kernel<WsGeom<false, 2, 9>>'"""
    n = name.strip()
    if n.startswith("void "):
        n = n[5:]
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    n = n[:cut].strip()
    while "> >" in n:
        n = n.replace("> >", ">>")
    return n


def main() -> None:
    tag = sys.argv[1] if len(sys.argv) > 1 else "latest"
    cfg = sys.argv[2] if len(sys.argv) > 2 else "1"  # BASELINE configs[] index the window was taken on (tools/pmc_collect.sh CFG=)
    suffix = "" if cfg == "1" else f"_cfg{cfg}"
    src = os.path.join(ROOT, "gpurun_out", "pmc" + suffix)
    f = json.load(open(os.path.join(src, "FETCH_SIZE.json")))
    w = json.load(open(os.path.join(src, "WRITE_SIZE.json")))
    out = {}
    for name, fv in f.items():
        if name not in w or not normalise(name).startswith(("conv", "wgrad", "attention", "linear_mfma", "edm_", "gn_", "lowres", "maxpool2")):
            continue
        wv = w[name]
        out[normalise(name)] = {
            "launches": fv["launches"], "fetch_bytes_per_launch": fv["mean"] * 2 * 1024, "write_bytes_per_launch": wv["mean"] * 1024,
            "hbm_bytes_per_launch": fv["mean"] * 2 * 1024 + wv["mean"] * 1024,
            "workload": f"one bench window of configs[{cfg}] + its warm-up window (bench.py --config {cfg} --steps 1 --warmup 1), separate --pmc passes",
            "profile_set": tag}
    cal = [k for k in f if "heun_step_kernel" in k and k in w]  # bench.py --pmc-calibrate: 2 x (4 x 256 MiB read, 256 MiB written)
    if cal:
        cp = cal[0]
        out["_calibration"] = {"kernel": cp, "known_bytes_read": 2 * 2 ** 30, "known_bytes_written": 2 ** 29,
                               "FETCH_SIZE_total_KiB": f[cp]["total"], "WRITE_SIZE_total_KiB": w[cp]["total"],
                               "fetch_factor": 2 * 2 ** 30 / (f[cp]["total"] * 1024),
                               "write_factor": 2 ** 29 / (w[cp]["total"] * 1024)}
    dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic{suffix}.json")
    json.dump(out, open(dst, "w"), indent=1)
    shutil.copyfile(dst, os.path.join(ROOT, "profiles", f"pmc_traffic{suffix}.json"))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        shutil.copyfile(os.path.join(src, f"{c}.json"), os.path.join(ROOT, "profiles", f"{tag}_pmc_{c}{suffix}.json"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
