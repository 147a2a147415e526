"""gpurun_out/pmc_sq_<tag>/pass{1,2}.json (+ durations) of tools/pmc_sq.sh -> profiles/<set>_sq_counters.json and profiles/sq_counters.json
(the copy bench.py reads for `roofline.kernels[*].mfma_busy`).

    python tools/pmc_sq_to_profile.py gpurun_out/pmc_sq_r06 r06 [--tagged-only]

Derived per kernel (per-launch averages summed over the chip):
  mfma_busy          SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs  over  SQ_BUSY_CYCLES / 32 shader engines' SQs: the share of the kernel's busy
                     cycles in which a SIMD's matrix pipe is executing -- both counters tick in the same (throttled) clock domain, so no clock
                     assumption enters (the r03 file derived it from a wall time x an assumed clock: 0.38-0.57 depending on the clock)
  mfma_insts_x32     SQ_INSTS_MFMA x 32 cycles / 1024 / (SQ_BUSY_CYCLES / 32): the same from the instruction count (a 32x32x16 f16 MFMA occupies
                     the pipe 32 cycles, profiles/r04_clock.json); 16x16 instances issue shorter MFMAs, there this is an upper estimate
  valu_per_mfma      SQ_INSTS_VALU (incl. MFMA) / SQ_INSTS_MFMA
  lds_conflict_share SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_inst_share    SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (wave-cycles spent waiting for an instruction to become issuable)
  valu_active, vmem_active, lds_active, scalar_active, mfma_valu_coexec
                     SQ_ACTIVE_INST_{VALU,VMEM,LDS,SCA} / SQ_VALU_MFMA_COEXEC_CYCLES normalised like mfma_busy (per SIMD, over the kernel's busy
                     cycles; ACTIVE_INST counters add up over the waves of a SIMD, so a value above 1 means several waves inside such
                     instructions at once): which issue resource the non-MFMA time goes to
  valu_issue_share_est  issue cycles of the non-MFMA vector instructions (4 per instruction, 8 per transcendental) per SIMD over the busy cycles
  neither_pipe_share_est  1 - mfma_busy - (valu_issue_share_est - mfma_valu_coexec): the share of a SIMD's cycles in which neither the matrix
                     pipe nor the vector ALU works (waits: LDS, barriers, vector memory)
  l2_hit_rate        TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)"""
import json, os, sys

src, tag = sys.argv[1], sys.argv[2]
only_tagged = len(sys.argv) > 3 and sys.argv[3] == "--tagged-only"  # (another workload than bench.py's: profiles/sq_counters.json stays)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import glob
passes = {}
for f in sorted(glob.glob(os.path.join(src, "pass[0-9].json"))):
    try:
        passes[int(os.path.basename(f)[4])] = json.load(open(f))
    except Exception:  # (a pass whose counters this rocprofv3 does not know leaves an empty file)
        pass
p1, p2 = passes.get(1, {}), passes.get(2, {})
dur = {}
for i in sorted(passes):
    f = os.path.join(src, f"pass{i}_durations.json")
    if os.path.exists(f):
        for k, v in json.load(open(f)).items():
            dur.setdefault(k, {})[f"avg_us_pass{i}"] = round(v["avg_us"], 2)
out = {"_about": __doc__.split("Derived per kernel")[0].strip() + "  Workload: tools/pmc_target.py (denoiser forwards, batch 256, 64x64), two rocprofv3 --pmc "
       "passes of 8 SQ counters each (no tracing domains beside --kernel-trace).",
       "_derived": "Derived per kernel" + __doc__.split("Derived per kernel")[1], "profile_set": tag, "kernels": {}}
for k in sorted(set(p1) | set(p2)):
    d = dict(p1.get(k, {}))
    for i in sorted(passes):
        if i > 1:
            d.update({kk: vv for kk, vv in passes[i].get(k, {}).items() if kk != "launches"})
    d.update(dur.get(k, {}))
    busy = d.get("SQ_BUSY_CYCLES")
    if busy:
        per_sq = busy / 32.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            d["mfma_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / per_sq, 4)
        if "SQ_INSTS_MFMA" in d:
            d["mfma_insts_x32"] = round(d["SQ_INSTS_MFMA"] * 32.0 / 1024.0 / per_sq, 4)
            if d["SQ_INSTS_MFMA"]:
                d["valu_per_mfma"] = round(d.get("SQ_INSTS_VALU", 0.0) / d["SQ_INSTS_MFMA"], 3)
    if busy:  # shares of the kernel's busy cycles, per SIMD (x / 1024 over SQ_BUSY_CYCLES / 32), like mfma_busy
        for name, key in (("valu_active", "SQ_ACTIVE_INST_VALU"), ("vmem_active", "SQ_ACTIVE_INST_VMEM"), ("lds_active", "SQ_ACTIVE_INST_LDS"),
                          ("scalar_active", "SQ_ACTIVE_INST_SCA"), ("mfma_valu_coexec", "SQ_VALU_MFMA_COEXEC_CYCLES")):
            if key in d:
                d[name] = round(d[key] / 1024.0 / (busy / 32.0), 4)
    if busy and "SQ_INSTS_VALU" in d and "SQ_INSTS_MFMA" in d and "SQ_INSTS_VALU_TRANS_F32" in d:
        # issue cycles of the non-MFMA vector instructions: 4 per wave64 instruction, 8 for the transcendentals (profiles/r04_trans_probe.txt)
        plain = d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"] - d["SQ_INSTS_VALU_TRANS_F32"]
        d["valu_issue_share_est"] = round((4.0 * plain + 8.0 * d["SQ_INSTS_VALU_TRANS_F32"]) / 1024.0 / (busy / 32.0), 4)
        if "mfma_busy" in d and "mfma_valu_coexec" in d:
            d["neither_pipe_share_est"] = round(max(0.0, 1.0 - d["mfma_busy"] - (d["valu_issue_share_est"] - d["mfma_valu_coexec"])), 4)
    if d.get("TCC_HIT_sum") is not None and (d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0)) > 0:
        d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
    if d.get("SQ_WAVE_CYCLES"):
        d["wait_inst_share"] = round(d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"], 4)
    short = k.replace("void ", "").split("(")[0].replace("> >", ">>")
    out["kernels"][short] = d
for name in (f"{tag}_sq_counters.json",) + (() if only_tagged else ("sq_counters.json",)):
    json.dump(out, open(os.path.join(root, "profiles", name), "w"), indent=1)
for k, d in out["kernels"].items():
    print(f"{k[:64]:64s} mfma_busy {d.get('mfma_busy')}  valu {d.get('valu_active')}  vmem {d.get('vmem_active')}  lds {d.get('lds_active')}  coexec {d.get('mfma_valu_coexec')}  valu_issue~ {d.get('valu_issue_share_est')}  neither~ {d.get('neither_pipe_share_est')}  l2hit {d.get('l2_hit_rate')}  wait_inst {d.get('wait_inst_share')}")
