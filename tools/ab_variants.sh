#!/bin/bash
# Run bench.py REPS times per variant library, interleaved, and print half-step us / ms_per_step
# (developer A/B tool).  Usage: tools/ab_variants.sh [reps]
cd "$(dirname "$0")/.."
reps=${1:-1}
for r in $(seq $reps); do
for so in graph-normalizing-flows_amd/variants/libgnf_*.so; do
  n=$(basename $so .so)
  GNF_LIB_PATH=$PWD/$so python bench.py --steps 40 --warmup 10 --no-cpu-baseline --kernel-timing-steps 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n', 'half_step_us', d['roofline']['kernel_us'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])
"
done
done
