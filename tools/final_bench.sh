#!/bin/bash
# One un-profiled bench line per workload -> gpurun_out/<tag>_bench_<workload>.json (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r3z}
cd $R
for w in config2_fc config2_attn default_flags config4 config5 wide_fc config2_train default_flags_train; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --latency-steps 0 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_$w.json
done
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_config2_default_run.json
timeout 900 python bench.py --all-workloads --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_all_workloads.json
