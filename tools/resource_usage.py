"""Registers / scratch / LDS of every kernel the library instantiates (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/resource_usage.py [--check] [file.hip ...]      (default: every source of diamond_amd/csrc)

One line per kernel.  --check: exit code 1 if a kernel uses scratch although it is not on the ALLOWED_SCRATCH list below
(tests/test_boundary.py runs this for the wave-specialised convolution and the attention kernels)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diamond_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "--cuda-device-only"]
# kernels that are known to spill, with the reason (everything else must have ScratchSize 0)
ALLOWED_SCRATCH = {
}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().splitlines()
        return out if len(out) == len(names) else names
    except OSError:
        return names


def extra_flags(src_path):
    """per-source hipcc flags on top of the common ones (diamond_amd/csrc/extra_flags.txt, shared with build.sh)"""
    import os
    d = os.path.dirname(os.path.abspath(src_path))
    try:
        for line in open(os.path.join(d, "extra_flags.txt")):
            if not line.startswith("#") and line.split()[:1] == [os.path.basename(src_path)]:
                return line.split()[1:]
    except OSError:
        pass
    return []


def usage(path):
    r = subprocess.run(["hipcc"] + FLAGS + extra_flags(os.path.join(CSRC, path) if not os.path.isabs(path) else path) + ["-x", "hip", "-c", path, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True, cwd=CSRC)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"\s([A-Za-z][A-Za-z \[\]/]*): (\d+) \[-Rpass-analysis", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:])
        raise SystemExit(f"hipcc failed on {path}")
    for row, nm in zip(rows, demangle([r_["name"] for r_ in rows])):
        row["name"] = nm
    return rows


def main():
    args = [a for a in sys.argv[1:] if a != "--check"]
    files = args or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(8, len(files))) as ex:  # one hipcc per source file, in parallel
        tables = list(ex.map(lambda f: usage(os.path.join(CSRC, f) if not os.path.isabs(f) else f), files))
    for f, rows in zip(files, tables):
        for row in rows:
            scratch = row.get("ScratchSize [bytes/lane]", -1)
            note = ""
            if scratch != 0:
                key = next((k for k in ALLOWED_SCRATCH if k in row["name"]), None)
                note = f"  <-- scratch ({ALLOWED_SCRATCH[key]})" if key else "  <-- SCRATCH"
                bad += key is None
            print(f"{os.path.basename(f):20s} {row['name'][:96]:96s} VGPR {row.get('VGPRs', -1):3d} AGPR {row.get('AGPRs', -1):3d} "
                  f"SGPR {row.get('TotalSGPRs', -1):3d} scratch {scratch:4d} B/lane  LDS {row.get('LDS Size [bytes/block]', -1):6d} B  "
                  f"waves/SIMD {row.get('Occupancy [waves/SIMD]', -1)}{note}")
    if "--check" in sys.argv and bad:
        raise SystemExit(f"{bad} kernel(s) use scratch memory")


if __name__ == "__main__":
    main()
