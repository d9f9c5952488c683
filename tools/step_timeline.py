"""Developer tool: timeline of ONE training (or forward) step from a rocprofv3 kernel trace: every kernel between two
consecutive k_adam launches (or the given marker kernel), start offset / duration / gap to the previous kernel's end,
plus the busy sums per stream.    python tools/step_timeline.py <rocprof dir> [marker] [which]"""
import csv, glob, sys, collections
f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_adam"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
i0, i1 = idx[which], idx[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
last_end = {}
busy = collections.defaultdict(float)
print(f"step = {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us, {i1 - i0} kernels")
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy[q] += (e - s) / 1e3
    name = r["Kernel_Name"].replace("void ", "").replace("gnf::", "")[:48]
    print(f"q{q:>3} @{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {name}")
print({k: round(v, 1) for k, v in busy.items()})
