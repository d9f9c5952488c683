#!/bin/bash
# PMC passes for one bench workload under one GNF_OPTIONS setting (run on the GPU box):
#   tools/pmc_shape.sh <tag> <workload> [GNF_OPTIONS] [kernel-name substrings, comma separated]
# -> gpurun_out/pmc_<tag>/{kernel_stats.txt, pmc_means.txt}
# The counter means are printed for the launch's dominant kernel and for every kernel whose name contains one of the
# substrings of the 4th argument (e.g. "k_aggregate,k_half_bwd_dw"), each with its rocprofv3 average duration.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; wl=$2; export GNF_OPTIONS=$3; also=$4
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload $wl --no-cpu-baseline --no-secondary --latency-steps 0 --prewarm-ms 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o kt -- $B --steps 20 --warmup 5 > $out/bench_under_rocprof.log 2>&1
python -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_stamp('$wl'), ' '.join(sorted(set(bench.WORKLOAD_SOURCES['$wl']))))" > $out/source_stamp.txt
python $R/tools/kstats.py $out/trace 6 > $out/kernel_stats.txt
i=0
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT" \
         "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES" \
         "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d $out/p$i -o pmc -- $B --steps 3 --warmup 1 --kernel-timing-steps 1 > $out/p$i.log 2>&1
done
python - <<PY > $out/pmc_means.txt
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True)[0])))
dom = max(rows, key=lambda r: float(r["TotalDurationNs"]))["Name"]
names = [dom] + [r["Name"] for r in rows if r["Name"] != dom and any(s and s in r["Name"] for s in "$also".split(","))]
avg = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in rows}
calls = {r["Name"]: int(r["Calls"]) for r in rows}
st = open("$out/source_stamp.txt").read().split()
print("# workload $wl  GNF_OPTIONS=$3  kernel sources stamp (bench.kernel_source_stamp):", st[0], " sources:", ",".join(st[1:]))
print("# PMC passes (separate runs, --pmc only with --kernel-trace): mean per dispatch; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 prints them")
print("# (HBM-side bytes per launch = FETCH_SIZE x 1024 x 2 [gfx950 correction, MI355X_MICROARCH.md] + WRITE_SIZE x 1024)")
files = sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True))
data = [list(csv.DictReader(open(f))) for f in files]
for k, name in enumerate(names):
    print(f"# {'dominant kernel' if k == 0 else 'kernel'}: {name[:110]}  (kernel-trace pass: {calls[name]} calls, average {avg[name]:.2f} us)")
    got = {}
    for rws in data:
        agg = collections.defaultdict(list)
        for r in rws:
            if r["Kernel_Name"] == name:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in agg.items():
            got[c] = sum(v) / len(v)
            print(f"{sum(v)/len(v):18.1f}  n={len(v):4d}  {c}")
    if "FETCH_SIZE" in got and "WRITE_SIZE" in got:
        b = got["FETCH_SIZE"] * 1024 * 2 + got["WRITE_SIZE"] * 1024
        print(f"#   -> HBM-side traffic {b / 1e6:.2f} MB per launch = {b / avg[name] / 1e3:.1f} GB/s over the kernel-trace average")
PY
$B --steps 50 --warmup 10 > $out/bench.json 2> $out/bench.err
rm -rf $out/trace $out/p[0-9] $out/p[0-9].log   # (the raw csv files: tens of MB per workload; gpurun merges 64 MiB back at most)
cat $out/kernel_stats.txt $out/pmc_means.txt; tail -1 $out/bench.json
