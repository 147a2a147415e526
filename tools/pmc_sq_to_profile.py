"""gpurun_out/pmc_sq_<tag>/pass{1,2}.json (+ durations) of tools/pmc_sq.sh -> profiles/<set>_sq_counters.json and profiles/sq_counters.json
(the copy bench.py reads for `roofline.kernels[*].mfma_busy`).

    python tools/pmc_sq_to_profile.py gpurun_out/pmc_sq_r06 r06

Derived per kernel (per-launch averages summed over the chip):
  mfma_busy          SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs  over  SQ_BUSY_CYCLES / 32 shader engines' SQs: the share of the kernel's busy
                     cycles in which a SIMD's matrix pipe is executing -- both counters tick in the same (throttled) clock domain, so no clock
                     assumption enters (the r03 file derived it from a wall time x an assumed clock: 0.38-0.57 depending on the clock)
  mfma_insts_x32     SQ_INSTS_MFMA x 32 cycles / 1024 / (SQ_BUSY_CYCLES / 32): the same from the instruction count (a 32x32x16 f16 MFMA occupies
                     the pipe 32 cycles, profiles/r04_clock.json); 16x16 instances issue shorter MFMAs, there this is an upper estimate
  valu_per_mfma      SQ_INSTS_VALU (incl. MFMA) / SQ_INSTS_MFMA
  lds_conflict_share SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_inst_share    SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (wave-cycles spent waiting for an instruction to become issuable)"""
import json, os, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p1, p2 = (json.load(open(os.path.join(src, f"pass{i}.json"))) for i in (1, 2))
dur = {}
for i in (1, 2):
    f = os.path.join(src, f"pass{i}_durations.json")
    if os.path.exists(f):
        for k, v in json.load(open(f)).items():
            dur.setdefault(k, {})[f"avg_us_pass{i}"] = round(v["avg_us"], 2)
out = {"_about": __doc__.split("Derived per kernel")[0].strip() + "  Workload: tools/pmc_target.py (denoiser forwards, batch 256, 64x64), two rocprofv3 --pmc "
       "passes of 8 SQ counters each (no tracing domains beside --kernel-trace).",
       "_derived": "Derived per kernel" + __doc__.split("Derived per kernel")[1], "profile_set": tag, "kernels": {}}
for k in sorted(set(p1) | set(p2)):
    d = dict(p1.get(k, {}))
    d.update({kk: vv for kk, vv in p2.get(k, {}).items() if kk != "launches"})
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
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
    if d.get("SQ_WAVE_CYCLES"):
        d["wait_inst_share"] = round(d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"], 4)
    short = k.replace("void ", "").split("(")[0].replace("> >", ">>")
    out["kernels"][short] = d
for name in (f"{tag}_sq_counters.json", "sq_counters.json"):
    json.dump(out, open(os.path.join(root, "profiles", name), "w"), indent=1)
for k, d in out["kernels"].items():
    print(f"{k[:64]:64s} mfma_busy {d.get('mfma_busy')}  by insts {d.get('mfma_insts_x32')}  valu/mfma {d.get('valu_per_mfma')}  lds conflicts {d.get('lds_conflict_share')}  wait_inst {d.get('wait_inst_share')}")
