"""Samples the GPU's clock / power sensors while something else runs (round 4: which clock does the chip run under which kernel?).

    python tools/smi_sampler.py out.csv [hz] &      # samples until it gets SIGTERM / SIGINT, then prints a JSON summary
    ... workload ...
    kill $!

Sources, whichever the box has: the amdgpu hwmon files (freq1_input = sclk in Hz, power1_average / power1_input in microwatts),
`pp_dpm_sclk` (the DPM level marked with `*`), and -- once per second, they take ~0.3 s per call -- `rocm-smi --showclocks
--showpower --json` / `amd-smi metric --clock --power --json`.  Every sample is one CSV row `t, source, field, value`."""
import glob
import json
import os
import signal
import subprocess
import sys
import time

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/smi.csv"
hz = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
stop = False


def _stop(*_):
    global stop
    stop = True


signal.signal(signal.SIGTERM, _stop)
signal.signal(signal.SIGINT, _stop)

files = {}
for pat, name in (("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", "sclk_hz"),
                  ("/sys/class/drm/card*/device/hwmon/hwmon*/freq2_input", "mclk_hz"),
                  ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "power_uw"),
                  ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", "power_in_uw"),
                  ("/sys/class/drm/card*/device/pp_dpm_sclk", "dpm_sclk")):
    for p in sorted(glob.glob(pat)):
        files[p] = name
rows = []
t0 = time.time()
last_cli = 0.0


def cli_sample(t):
    for cmd, src in ((["rocm-smi", "--showclocks", "--showpower", "--json"], "rocm-smi"),
                     (["amd-smi", "metric", "--clock", "--power", "--json"], "amd-smi")):
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=5)
            txt = r.stdout.decode(errors="replace").strip()
            if not txt:
                continue
            try:
                txt = json.dumps(json.loads(txt))
            except ValueError:
                txt = txt[:4000].replace("\n", " ")
            rows.append((t, src, "raw", txt))
        except Exception as e:  # noqa: BLE001
            rows.append((t, src, "error", repr(e)[:200]))


while not stop:
    t = time.time() - t0
    for p, name in files.items():
        try:
            v = open(p).read().strip()
        except OSError:
            continue
        if name == "dpm_sclk":
            cur = [ln for ln in v.splitlines() if ln.endswith("*")]
            v = cur[0].split()[1].rstrip("Mhz*") if cur else ""
        rows.append((t, "sysfs", name, v))
    if t - last_cli >= 1.0:
        last_cli = t
        cli_sample(t)
    time.sleep(max(0.0, 1.0 / hz - ((time.time() - t0) - t)))

with open(out, "w") as fh:
    fh.write("t,source,field,value\n")
    for t, s, f, v in rows:
        fh.write(f"{t:.3f},{s},{f},{json.dumps(v) if (',' in str(v)) else v}\n")
summ = {}
for _, s, f, v in rows:
    if s == "sysfs":
        try:
            summ.setdefault(f, []).append(float(v))
        except ValueError:
            pass
res = {}
for f, vs in summ.items():
    vs.sort()
    scale = 1e-6 if f.endswith("_hz") else (1e-6 if f.endswith("_uw") else 1.0)
    unit = "MHz" if f.endswith("_hz") else ("W" if f.endswith("_uw") else "MHz")
    res[f] = {"n": len(vs), "median": vs[len(vs) // 2] * scale, "p10": vs[len(vs) // 10] * scale, "p90": vs[(9 * len(vs)) // 10] * scale,
              "max": vs[-1] * scale, "unit": unit}
res["sysfs_files"] = sorted(files)
res["cli_samples"] = sum(1 for r in rows if r[2] == "raw")
print(json.dumps(res))
