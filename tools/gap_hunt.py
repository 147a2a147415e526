"""Join a rocprofv3 kernel trace with the HIP API trace (development aid): for every GPU idle gap of >= MIN_US inside the last
window, which API call launched the kernel behind the gap, how long before the gap began was it issued (host ahead: the gap is a
device-side wait; host behind: the host is the critical path), and which API calls / memory copies lie between the launches of
the two kernels around the gap.
usage: python tools/gap_hunt.py kernel_trace.csv hip_api_trace.csv [memory_copy_trace.csv]   (env: MIN_US=60, LAST_MS=330)"""
import csv, os, sys
from collections import Counter

min_gap = float(os.environ.get("MIN_US", "60")) * 1e3
last = float(os.environ.get("LAST_MS", "330")) * 1e6
K = [r for r in csv.DictReader(open(sys.argv[1]))]
A = [r for r in csv.DictReader(open(sys.argv[2]))] if len(sys.argv) > 2 and sys.argv[2] else []
M = [r for r in csv.DictReader(open(sys.argv[3]))] if len(sys.argv) > 3 and sys.argv[3] else []
print("kernel columns:", list(K[0].keys()))
if A:
    print("api columns:", list(A[0].keys()))
if M:
    print("copy columns:", list(M[0].keys()))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Correlation_Id"), r.get("Stream_Id", r.get("Queue_Id"))) for r in K))
end = ks[-1][1]
ks = [k for k in ks if k[0] >= end - last]
api = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Correlation_Id"), r.get("Thread_Id")) for r in A))
by_corr = {a[3]: a for a in api}
copies = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", ""), r.get("Correlation_Id")) for r in M))
print(f"{len(ks)} kernels in the last {last/1e6:.0f} ms; streams: {Counter(k[4] for k in ks).most_common(6)}")
short = lambda n: n.replace("void ", "").replace("at::native::", "")[:70]
classes = Counter()
shown = 0
prev = ks[0]
frontier = ks[0][1]
for k in ks[1:]:
    g = k[0] - frontier
    if g >= min_gap:
        la, lb = by_corr.get(prev[3]), by_corr.get(k[3])
        lead = (frontier - lb[1]) / 1e3 if lb else None  # > 0: the launch call had returned that long BEFORE the device went idle
        kind = "device-side" if lead is not None and lead > 0 else "host-side"
        classes[(short(prev[2])[:40], short(k[2])[:40], kind)] += g
        if shown < 40:
            shown += 1
            print(f"\ngap {g/1e3:7.1f} us at -{(end-k[0])/1e6:7.2f} ms  {short(prev[2])} -> {short(k[2])}   [{kind}; launch returned {lead} us before the gap began]")
            if la and lb:
                between = [a for a in api if la[0] <= a[0] <= lb[1] and a[4] == lb[4]]
                for a in between[:60]:
                    print(f"      {(a[0]-frontier)/1e3:9.1f} .. {(a[1]-frontier)/1e3:9.1f} us  {a[2]}")
                for c in copies:
                    if frontier - 2e5 <= c[0] <= k[0] + 1e4:
                        print(f"      copy {c[2]} {(c[0]-frontier)/1e3:9.1f} .. {(c[1]-frontier)/1e3:9.1f} us")
    frontier = max(frontier, k[1])
    prev = k if k[1] >= frontier else prev
print("\ngap classes (ms):")
for key, t in classes.most_common(25):
    print(f"  {t/1e6:7.2f}  {key}")
