"""GPU idle-gap analysis of a rocprofv3 kernel trace CSV (development aid).
usage: python tools/trace_gaps.py kernel_trace.csv[.gz] [min_gap_us] [last_ms]   (last_ms: look at the last N ms of the trace only)
Prints the busy fraction, a gap histogram, the largest (previous kernel -> next kernel) gap classes and every individual gap
above min_gap_us (default 500) with its time before the end of the trace -- for `bench.py --steps K --warmup W` the timed
windows are the last K x ~330 ms."""
import csv, gzip, sys
from collections import defaultdict
path = sys.argv[1]
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 5e5
f = gzip.open(path, "rt") if path.endswith(".gz") else open(path)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(f)]
rows.sort()
if len(sys.argv) > 3:  # the timed window(s) at the end of the trace only (warm-up windows build caches and idle more)
    cut = rows[-1][1] - float(sys.argv[3]) * 1e6
    rows = [r for r in rows if r[0] >= cut]
end = rows[-1][1]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print(f"span {span/1e6:.1f} ms, kernel busy {busy/1e6:.1f} ms ({100*busy/span:.1f} %), {len(rows)} kernels")
gaps = defaultdict(lambda: [0, 0])
hist = defaultdict(lambda: [0, 0])
big = []
prev_end, prev_name = rows[0][1], rows[0][2]
for s, e, n in rows[1:]:
    g = s - prev_end
    if g > 0:
        key = (prev_name[:50], n[:50])
        gaps[key][0] += 1
        gaps[key][1] += g
        b = "<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else "<1ms" if g < 1e6 else ">=1ms"
        hist[b][0] += 1
        hist[b][1] += g
        if g >= min_gap:
            big.append(((end - s) / 1e6, g / 1e6, prev_name[:60], n[:60]))
    prev_end, prev_name = max(prev_end, e), n
print("gap histogram:", {k: (v[0], round(v[1] / 1e6, 2)) for k, v in hist.items()})
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{t/1e6:8.2f} ms  {c:6d}x  {a}  ->  {b}")
# the neighbourhood of every gap >= 3 ms: the 5 kernels in front of it and the 5 behind it (what was the host doing?)
for i in range(1, len(rows)):
    g = rows[i][0] - max(r[1] for r in rows[max(0, i - 4):i])
    if g >= 3e6:
        print(f"gap of {g/1e6:.2f} ms at -{(end - rows[i][0])/1e6:.1f} ms:")
        for s_, e_, n_ in rows[max(0, i - 5):i]:
            print(f"      before  {n_[:110]}  {(e_-s_)/1e3:.1f} us")
        for s_, e_, n_ in rows[i:i + 5]:
            print(f"      after   {n_[:110]}  {(e_-s_)/1e3:.1f} us")
print(f"individual gaps >= {min_gap/1e3:.0f} us (ms before the end of the trace, gap ms, previous -> next):")
for t, g, a, b in big[-80:]:
    print(f"  -{t:8.1f}  {g:7.2f}  {a}  ->  {b}")

# the stretches between two LARGE kernels (>= 50 us): what the device does while the host is the critical path
stretch, cur = [], None
for s_, e_, n_ in rows:
    if e_ - s_ >= 50_000:
        if cur is not None and cur[2] >= 25:
            stretch.append(cur)
        cur = [e_, e_, 0, 0, []]  # start (end of a large kernel), last end, small kernels, their busy ns, their names
    elif cur is not None:
        cur[1], cur[2], cur[3] = max(cur[1], e_), cur[2] + 1, cur[3] + (e_ - s_)
        cur[4].append((n_, e_ - s_, s_ - cur[0]))
long_ = sorted(stretch, key=lambda c: -(c[1] - c[0]))[:40]
tot = sum(c[1] - c[0] for c in stretch)
print(f"stretches of >= 25 small kernels between two large ones: {len(stretch)}, {tot/1e6:.1f} ms in total; the longest (ms, kernels, their busy ms):")
print("  " + "  ".join(f"{(c[1]-c[0])/1e6:.2f}/{c[2]}/{c[3]/1e6:.2f}" for c in long_))

import os
want = os.environ.get("TRACE_GAPS_STRETCH_WITH", "categorical_sample")  # the per-step FORWARD stretches contain the action sampling
fwd = [c for c in stretch if any(want in n_ for n_, _, _ in c[4])]
if fwd:
    tot_f = sum(c[1] - c[0] for c in fwd)
    print(f"stretches containing '{want}': {len(fwd)}, {tot_f/1e6:.1f} ms in total, {sum(c[3] for c in fwd)/1e6:.1f} ms of it busy, "
          f"{sum(c[2] for c in fwd)/len(fwd):.0f} kernels each")
    stretch = fwd
if stretch:  # the kernels of the MEDIAN stretch, in order: name, duration us, start offset us
    import re
    med = sorted(stretch, key=lambda c: c[1] - c[0])[len(stretch) // 2]
    print(f"the median stretch ({(med[1]-med[0])/1e6:.2f} ms, {med[2]} kernels), in order (start us : name, us):")
    short = lambda n: re.sub(r"at::native::|\(anonymous namespace\)::|void |std::array<char\*, \d+ul> ?|<unnamed>::", "", n)[:90]
    for n_, d_, off in med[4]:
        print(f"  {off/1e3:8.1f} : {short(n_)}  {d_/1e3:.1f}")

# TRACE_GAPS_DUMP=<k>: every kernel of the stretch whose kernel count is the most frequent one >= k (the typical per-step stretch)
dump = os.environ.get("TRACE_GAPS_DUMP")
if dump and stretch:
    from collections import Counter
    cnt = Counter(c[2] for c in stretch if c[2] >= int(dump))
    if cnt:
        mode = cnt.most_common(1)[0][0]
        pick = sorted([c for c in stretch if c[2] == mode], key=lambda c: c[1] - c[0])
        med = pick[len(pick) // 2]
        print(f"the typical stretch of {mode} kernels ({len(pick)} of them; this one {(med[1]-med[0])/1e6:.2f} ms wall, {med[3]/1e6:.2f} ms busy), in order:")
        for n_, d_, off in med[4]:
            print(f"  {off/1e3:8.1f} : {short(n_)}  {d_/1e3:.1f}")
