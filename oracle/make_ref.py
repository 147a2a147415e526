"""Recipe for `oracle/_ref/`: the REFERENCE ITSELF as a CPU baseline that can travel.  *** TEST INFRASTRUCTURE ONLY ***

    python oracle/make_ref.py            (build container only: needs /root/reference)

The reference is Python; the GPU box has no /root/reference.  This compiles the reference's own sources WHERE THEY LIE
(`/root/reference/src/**/*.py`) to CPython bytecode -- `py_compile` with an explicit output path, i.e. the same thing a
Makefile does for a C reference: outputs only, into the git-ignored `oracle/_ref/src/` (no source file is copied, nothing is
written under /root/reference).  The bytecode imports sourceless (`agent.pyc`, `models/blocks.pyc`, ...) on the GPU box, which
runs the same image and therefore the same interpreter (checked at import: the magic number is stored in `MANIFEST.json`).

Who may use the result: `oracle/reference_window.py`, and through it only tests/ and bench.py's `cpu_baseline` leg
(`kind: "reference"`).  The product never imports it.
"""
import importlib.util
import json
import os
import py_compile
import shutil
import sys

REF_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref", "src")


def build(verbose: bool = True) -> bool:
    """Returns False (and does nothing) where the reference is absent."""
    if not os.path.isdir(REF_SRC):
        return False
    # built beside the old copy and swapped in on success: a reference file that does not compile leaves the last good
    # oracle/_ref in place (and raises: the caller decides what that means -- __graft_entry__.build() only logs it)
    tmp = os.path.join(HERE, "_ref.tmp")
    shutil.rmtree(tmp, ignore_errors=True)
    out = os.path.join(tmp, "src")
    n = 0
    for root, dirs, files in os.walk(REF_SRC):
        dirs[:] = [d for d in dirs if d != "__pycache__"]
        for f in files:
            if not f.endswith(".py"):
                continue
            src = os.path.join(root, f)
            rel = os.path.relpath(src, REF_SRC)
            dst = os.path.join(out, rel + "c")  # legacy (sourceless) location: foo.pyc where foo.py would be
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            py_compile.compile(src, cfile=dst, dfile=os.path.join("reference/src", rel), doraise=True,
                               invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
            n += 1
    with open(os.path.join(tmp, "MANIFEST.json"), "w") as fh:
        json.dump({"what": "CPython bytecode of /root/reference/src (py_compile, no sources), see oracle/make_ref.py",
                   "files": n, "python": sys.version.split()[0], "magic": importlib.util.MAGIC_NUMBER.hex()}, fh, indent=1)
    shutil.rmtree(os.path.join(HERE, "_ref"), ignore_errors=True)
    os.replace(tmp, os.path.join(HERE, "_ref"))
    if verbose:
        print(f"oracle/_ref: {n} modules of {REF_SRC} compiled to bytecode")
    return True


if __name__ == "__main__":
    if not build():
        print(f"{REF_SRC} not present: nothing built", file=sys.stderr)
        sys.exit(1)
