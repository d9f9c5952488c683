"""Print the top kernels of a rocprofv3 --kernel-trace --stats --output-format csv directory."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
        print(f"{r['Name'][:72]:72s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {r['Percentage']}")
