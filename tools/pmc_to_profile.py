"""gpurun_out/pmc/{FETCH,WRITE}_SIZE.json -> profiles/pmc_traffic.json (bytes per launch per conv kernel).

Corrections (MI355X_MICROARCH.md, HBM section; re-checked here on a 1 GiB copy): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads ->
x2; WRITE_SIZE matched the known byte count 1:1."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = json.load(open(os.path.join(ROOT, "gpurun_out/pmc/FETCH_SIZE.json")))
w = json.load(open(os.path.join(ROOT, "gpurun_out/pmc/WRITE_SIZE.json")))
out = {}
for name, fv in f.items():
    m = re.search(r"conv_mfma_kernel<ConvGeom<(\d+), (true|false), (\d+), (\d+)>", name)
    m2 = re.search(r"conv_f16w?s_kernel<(?:F16|Ws)Geom<(true|false)(?:, (\d))?(?:, (\d))?>", name)
    if not m and not m2:
        continue
    if m2:
        key = f"conv_f16s<{'B8' if m2.group(1) == 'true' else 'A16'},c{32 * int(m2.group(2) or 2)}{',1x1' if m2.group(3) == '1' else ''}>"
    else:
        key = f"conv_mfma<WN{m.group(1)},{'B' if m.group(2) == 'true' else 'A'},taps{m.group(3)},s{m.group(4)}>"
    wv = w[name]
    out[key] = {"launches": fv["launches"], "fetch_bytes_per_launch": fv["mean"] * 2 * 1024,
                "write_bytes_per_launch": wv["mean"] * 1024,
                "hbm_bytes_per_launch": fv["mean"] * 2 * 1024 + wv["mean"] * 1024,
                "workload": "2 x Denoiser.denoise at B=256, 64x64 (tools/pmc_target.py), separate --pmc passes"}
cp = "__amd_rocclr_copyBuffer"
out["_calibration"] = {"kernel": cp, "known_bytes_read": 2 * 2 ** 30, "known_bytes_written": 2 * 2 ** 30,
                       "FETCH_SIZE_total_KiB": f[cp]["total"], "WRITE_SIZE_total_KiB": w[cp]["total"],
                       "fetch_factor": 2 * 2 ** 30 / (f[cp]["total"] * 1024), "write_factor": 2 * 2 ** 30 / (w[cp]["total"] * 1024)}
json.dump(out, open(os.path.join(ROOT, "profiles/pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
