"""Lint for the inline-asm activation loads of conv_f16ws_kernel (diamond_amd/csrc/dmd_conv_f16ws.hip).

hipcc does not know that an `asm volatile("global_load_dwordx4 ...")` leaves its destination registers pending until
the matching hand-counted `s_waitcnt vmcnt(N) ; await v[a:b]` statement: it may copy, spill or overwrite them in
between (guide §5.7 item 1), silently.  This script compiles the file with -save-temps (or reads a given .s) and checks,
per kernel, in program order:
  * no instruction reads or writes a destination register between an asm load and the await that names it,
  * no scratch (spill) instruction exists in a kernel that uses asm loads,
  * the compiler's own `s_waitcnt vmcnt(0)` count inside such kernels is reported (each one drains the prefetch).
The check is a forward may-analysis over the kernel's control-flow graph (labels, s_branch / s_cbranch, s_endpgm).
Exit code 1 on a violation.

    python tools/asm_lint.py [file.s]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "diamond_amd", "csrc", "dmd_conv_f16ws.hip")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
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


def compile_s():
    d = tempfile.mkdtemp(prefix="asm_lint_")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *extra_flags(SRC), "-x", "hip", "-c", SRC,
           "-save-temps", "-o", os.path.join(d, "x.o")]
    subprocess.check_call(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(d):
        if f.endswith("gfx950.s"):
            return os.path.join(d, f)
    raise SystemExit("no device .s produced")


def parse_kernels(path):
    """{kernel: [(line_no, kind, text)]} with kind in {label, asm_load, await, asm_drain, inst}."""
    kernels, cur, in_asm = {}, None, False
    for no, line in enumerate(open(path).read().splitlines(), 1):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None or not t:
            continue
        if t.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            cur.append((no, "label", m.group(1)))
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith(";") or t.startswith("."):
            continue
        code = t.split(";")[0].strip()
        if in_asm and code.startswith("global_load_dwordx4"):
            cur.append((no, "asm_load", code))
        elif in_asm and code.startswith("s_waitcnt") and "await" in t:
            cur.append((no, "await", t))
        elif in_asm and code.startswith("s_waitcnt vmcnt(0)"):
            cur.append((no, "asm_drain", code))
        else:
            cur.append((no, "inst", code))
    return kernels


def lint_kernel(path, name, insts):
    """Forward may-analysis of `pending` (registers with an asm load in flight) over the kernel's CFG."""
    # basic blocks
    starts = {0}
    label_at = {}
    for i, (no, kind, text) in enumerate(insts):
        if kind == "label":
            starts.add(i)
            label_at[text] = i
        elif kind == "inst" and re.match(r"s_(c?branch|endpgm)", text):
            starts.add(i + 1)
    starts = sorted(x for x in starts if x < len(insts))
    block_of = {}
    blocks = []
    for bi, b in enumerate(starts):
        e = starts[bi + 1] if bi + 1 < len(starts) else len(insts)
        blocks.append((b, e))
        block_of[b] = bi
    succ = []
    for b, e in blocks:
        no, kind, text = insts[e - 1]
        out = []
        if kind == "inst" and text.startswith("s_endpgm"):
            pass
        elif kind == "inst" and text.startswith("s_branch"):
            out.append(block_of[label_at[text.split()[1]]])
        else:
            # s_cbranch_execz: taken only with EXEC == 0, where no vector instruction (asm load included) has any effect
            if kind == "inst" and text.startswith("s_cbranch") and not text.startswith("s_cbranch_execz"):
                out.append(block_of[label_at[text.split()[1]]])
            if e < len(insts):
                out.append(block_of[e])
        succ.append(out)

    def transfer(bi, state, report):
        pending = dict(state)
        bad = 0
        b, e = blocks[bi]
        for no, kind, text in insts[b:e]:
            if kind == "label":
                continue
            if kind == "asm_load":
                dst = regs(text.split(",")[0])
                addr = regs(",".join(text.split(",")[1:]))
                if report:
                    for r in sorted(dst & set(pending)):
                        print(f"{path}:{no}: {name}: asm load overwrites v{r}, still pending from line {pending[r]}")
                        bad += 1
                    hit = addr & (set(pending) - dst)
                    if hit:
                        print(f"{path}:{no}: {name}: asm load address uses pending registers {sorted(hit)}")
                        bad += 1
                for r in dst:
                    pending[r] = no
            elif kind == "await":
                named = regs(text.split("await")[1])
                if report and not named:
                    print(f"{path}:{no}: {name}: await names no register")
                    bad += 1
                for r in named:
                    pending.pop(r, None)
            elif kind == "asm_drain":
                pending.clear()
            else:
                if re.match(r"s_waitcnt\b.*vmcnt\(0\)", text):
                    pending.clear()  # a compiler-inserted full drain also completes the asm loads (conservative, legal)
                    continue
                hit = regs(text) & set(pending)
                if hit and report:
                    print(f"{path}:{no}: {name}: `{text}` touches {sorted(hit)} pending from an asm load (line {min(pending[r] for r in hit)})")
                    bad += 1
        return pending, bad

    state_in = [dict() for _ in blocks]
    work = [0]
    seen = {0}
    while work:
        bi = work.pop()
        out, _ = transfer(bi, state_in[bi], False)
        for sj in succ[bi]:
            merged = dict(state_in[sj])
            changed = sj not in seen
            for r, no in out.items():
                if r not in merged:
                    merged[r] = no
                    changed = True
            if changed:
                state_in[sj] = merged
                seen.add(sj)
                work.append(sj)
    bad = 0
    for bi in range(len(blocks)):
        if bi in seen:
            bad += transfer(bi, state_in[bi], True)[1]
    return bad


def lint(path):
    bad = 0
    for name, insts in parse_kernels(path).items():
        loads = sum(1 for _, k, _ in insts if k == "asm_load")
        if not loads:
            continue
        awaits = sum(1 for _, k, _ in insts if k == "await")
        scratch = sum(1 for _, k, t in insts if k == "inst" and "scratch_" in t)
        drains = sum(1 for _, k, t in insts if k == "inst" and re.match(r"s_waitcnt\b.*vmcnt\(0\)", t))
        bad += lint_kernel(path, name, insts)
        if scratch:
            print(f"{name}: {scratch} scratch instructions in a kernel with inline-asm loads")
            bad += 1
        print(f"{name}: {loads} asm loads, {awaits} awaits, compiler vmcnt(0) waits: {drains}, scratch: {scratch}")
    return bad


if __name__ == "__main__":
    p = sys.argv[1] if len(sys.argv) > 1 else compile_s()
    n = lint(p)
    print("asm_lint:", "OK" if n == 0 else f"{n} violation(s)")
    sys.exit(1 if n else 0)
