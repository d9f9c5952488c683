#!/bin/bash
# rocprofv3 PMC passes (separate runs, --pmc only with --kernel-trace) for the generic GEMM on the wide_fc forward
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/bench.py --workload wide_fc --steps 2 --warmup 1 --kernel-timing-steps 1 --no-cpu-baseline --no-secondary"
i=0
for p in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm/p$i -o pmc -- $CMD > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/root/repo/gpurun_out/pmc_gemm/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'k_gemm' in k:
            acc[k + " grid " + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(acc.items()):
    print(k)
    for name in sorted(c):
        v = c[name]
        print(f"    {name:28s} {sum(v)/len(v):16.1f}   n={len(v)}")
    if 'TCC_HIT_sum' in c:
        h, m = sum(c['TCC_HIT_sum'])/len(c['TCC_HIT_sum']), sum(c['TCC_MISS_sum'])/len(c['TCC_MISS_sum'])
        print(f"    -> L2 hit rate {h/(h+m):.2f}")
PY
