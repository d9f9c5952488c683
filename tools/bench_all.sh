#!/bin/bash
# Every workload of bench.py once (developer tool): one JSON line each into gpurun_out/<tag>_workloads.jsonl
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r2}
out=$R/gpurun_out/${tag}_workloads.jsonl
: > $out
for w in config2_fc config4 config5 config2_attn wide_fc config2_train default_flags_train; do
  python $R/bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --kernel-timing-steps 5 2>/dev/null | grep '^{' >> $out
done
python - <<PY
import json
for l in open("$out"):
    d = json.loads(l)
    print(f"{d['metric'][:60]:60s} {d['ms_per_step']:9.4f} ms  {d['value']/1e6:8.2f} M/s  frac {d['roofline']['frac']}  rt {d.get('round_trip_max_abs_err')}")
PY
