#!/usr/bin/env python3
"""Are the kernels of this working tree the SAME MACHINE CODE as those of another revision?

    python tools/isa_diff.py <git-rev> [--rename OLD_SUBSTR=NEW_SUBSTR ...]

Compiles diamond_amd/csrc/*.hip of <git-rev> and of the working tree to gfx950 assembly (device side only) and compares every
kernel's instruction stream (labels normalised, comments and directives dropped).  Used when sources are refactored around
kernels whose measurements / GPU test runs should stay valid: e.g. the round-3 refactors for tests/simt left all 102 kernels of
the last GPU-validated commit byte-identical.  --rename maps mangled-name fragments when a template parameter list changed."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip --cuda-device-only -S".split()


def kernels(asm_path):
    txt = open(asm_path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        body = "\n".join(l for l in m.group(2).splitlines() if not l.strip().startswith((";", ".")))
        out[m.group(1)] = re.sub(r"\.LBB\d+_\d+", "L", body)
    return out


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


def compile_tree(csrc, outdir):
    procs = []
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
        o = os.path.join(outdir, os.path.basename(f)[:-4] + ".s")
        procs.append(subprocess.Popen(["hipcc", *FLAGS, *extra_flags(f), f, "-o", o], stderr=subprocess.DEVNULL, cwd=csrc))
    for p in procs:
        p.wait()
    res = {}
    for s in glob.glob(os.path.join(outdir, "*.s")):
        res.update(kernels(s))
    return res


def main():
    rev = sys.argv[1]
    renames = [a[len("--rename"):].lstrip("= ").split("=") for a in sys.argv[2:] if a.startswith("--rename")]
    with tempfile.TemporaryDirectory() as tmp:
        old_src = os.path.join(tmp, "old", "diamond_amd", "csrc")
        os.makedirs(old_src)
        os.makedirs(os.path.join(tmp, "old", "include"))
        names = subprocess.check_output(["git", "ls-tree", "--name-only", rev, "diamond_amd/csrc/", "include/"], cwd=ROOT, text=True).split()
        for n in names:
            if n.endswith((".hip", ".h", ".cpp", "extra_flags.txt")):  # (the per-source flags are part of what a revision compiles to)
                with open(os.path.join(tmp, "old", n), "wb") as fh:
                    fh.write(subprocess.check_output(["git", "show", f"{rev}:{n}"], cwd=ROOT))
        os.makedirs(os.path.join(tmp, "o")), os.makedirs(os.path.join(tmp, "n"))
        old = compile_tree(old_src, os.path.join(tmp, "o"))
        new = compile_tree(os.path.join(ROOT, "diamond_amd", "csrc"), os.path.join(tmp, "n"))
    same, diff, gone = 0, [], []
    for k, v in old.items():
        k2 = k
        for a, b in renames:
            k2 = k2.replace(a, b)
        if k2 not in new:
            gone.append(k)
        elif new[k2] == v:
            same += 1
        else:
            diff.append(k)
    print(f"{len(old)} kernels in {rev}: {same} byte-identical in the working tree, {len(diff)} different, {len(gone)} not found; "
          f"{len(new) - same - len(diff)} kernels only in the working tree")
    for k in diff:
        print("  DIFFERENT", k)
    for k in gone:
        print("  NOT FOUND", k)
    return 1 if diff or gone else 0


if __name__ == "__main__":
    sys.exit(main())
