#!/bin/bash
# Round 6 (after the pipelined loop's removal): GPU suite, bench line, rocprofv3 kernel stats of the bench command, and a kernel trace of
# the B = 1 latency mode (f4: where does a frame's time go?)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-r06e}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 -x > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench_line.json; tail -3 $O/bench.err
(cd /tmp && rm -rf /tmp/prof_lat && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lat -o lat -- \
   python $R/bench.py --config latency --steps 100 > $O/latency.log 2>&1
 t=$(find /tmp/prof_lat -name "*kernel_trace.csv" | head -1); k=$(find /tmp/prof_lat -name "*kernel_stats.csv" | head -1)
 [ -n "$k" ] && cp $k $O/latency_kernel_stats.csv
 [ -n "$t" ] && python - "$t" > $O/latency_trace_summary.txt <<'PY'
import csv, sys, collections
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
end = rows[-1][1]
rows = [r for r in rows if r[0] >= end - 100e6]          # the last 100 ms: ~30 graph-replayed frames
busy = sum(e - s for s, e, _ in rows); span = rows[-1][1] - rows[0][0]
print(f"last 100 ms: {len(rows)} kernels, busy {busy/1e6:.1f} ms of {span/1e6:.1f} ms ({100*busy/span:.1f} %), mean kernel {busy/len(rows)/1e3:.1f} us, mean gap {(span-busy)/len(rows)/1e3:.1f} us")
d = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    d[n[:90]][0] += 1; d[n[:90]][1] += e - s
for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e6:7.2f} ms {c:5d}x {t/c/1e3:7.1f} us  {n}")
PY
 cat $O/latency_trace_summary.txt | head -30)
(rocm-smi --showproductname --showclocks --showpower 2>/dev/null | head -40) > $O/box.txt
