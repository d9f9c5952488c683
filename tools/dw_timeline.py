"""Developer tool: per-half-step durations of the backward kernel and the dW GEMM from a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
i0, i1 = idx[20], idx[21]
t0 = int(rows[i0]["Start_Timestamp"])
out = []
for r in rows[i0:i1]:
    n = r["Kernel_Name"]
    if "half_bwd" in n or "gemm_dw" in n:
        out.append(("B" if "half_bwd" in n else "W", (int(r["Start_Timestamp"]) - t0) / 1e3,
                    (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print(" ".join(f"{k}@{s:.0f}:{d:.0f}" for k, s, d in out))
print("step", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "us")
