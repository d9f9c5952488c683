#!/bin/bash
# Round-end evidence run (on the GPU box): the PMC / kernel-stat passes of the forward workloads, kernel stats of the
# training workloads, one un-profiled bench line per workload.   tools/final_profile.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r3z}
cd $R
for w in config2 config4 config5 config2_attn; do
  timeout 600 bash tools/pmc_shape.sh ${tag}_$w $w > gpurun_out/pmc_${tag}_$w.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
for w in config2_train default_flags_train; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_${tag}_$w -o kt -- python $R/bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --latency-steps 0 > $R/gpurun_out/kt_${tag}_$w.log 2>&1
  python $R/tools/kstats.py $R/gpurun_out/kt_${tag}_$w 24 > $R/gpurun_out/kt_${tag}_$w.txt
done
cd $R
for w in config2_fc config2_attn default_flags config4 config5 wide_fc config2_train default_flags_train; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --latency-steps 0 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_$w.json
done
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_config2_default_run.json
ls gpurun_out | grep $tag
