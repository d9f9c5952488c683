#!/bin/bash
# rocprofv3 PMC passes (separate runs, --pmc only with --kernel-trace) for the training-step kernels; summary -> stdout
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/bench.py --workload config2_train --steps 3 --warmup 1 --kernel-timing-steps 1"
i=0
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d $R/gpurun_out/pmc_train/p$i -o pmc -- $CMD > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/root/repo/gpurun_out/pmc_train/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if any(t in k for t in ('k_half_bwd_fused', 'k_gemm_dw_grouped', 'k_half_fused', 'k_reduce_grouped', 'k_aggregate_bwd')):
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
print("# rocprofv3 --pmc <group> --kernel-trace -- python bench.py --workload config2_train --steps 3 --warmup 1   (r1e; mean per dispatch)")
print("# FETCH_SIZE / WRITE_SIZE are KiB at the L2's memory side (FETCH_SIZE x2 on gfx950 for wide coalesced reads, see MI355X_MICROARCH.md)")
for k, c in sorted(acc.items()):
    print(k)
    for name in sorted(c):
        v = c[name]
        print(f"    {name:28s} {sum(v)/len(v):16.1f}   n={len(v)}")
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'SQ_WAVE_CYCLES' in c:
        m, w = sum(c['SQ_VALU_MFMA_BUSY_CYCLES'])/len(c['SQ_VALU_MFMA_BUSY_CYCLES']), sum(c['SQ_WAVE_CYCLES'])/len(c['SQ_WAVE_CYCLES'])
        print(f"    -> matrix pipe busy / wave cycles = {m/w:.2f}")
    if 'FETCH_SIZE' in c:
        f_ = sum(c['FETCH_SIZE'])/len(c['FETCH_SIZE']); w_ = sum(c.get('WRITE_SIZE', [0]))/max(1, len(c.get('WRITE_SIZE', [0])))
        print(f"    -> memory-side traffic per dispatch ~ {2*f_/1024:.1f} MiB read (x2 corrected) + {w_/1024:.1f} MiB written")
PY
