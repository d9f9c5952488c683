#!/bin/bash
# Run bench.py once per variant library and print kernel_us / ms_per_step (developer A/B tool).
cd "$(dirname "$0")/.."
for so in graph-normalizing-flows_amd/variants/libgnf_*.so; do
  n=$(basename $so .so)
  GNF_LIB_PATH=$PWD/$so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --kernel-timing-steps 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n', 'kernel_us', d['roofline']['kernel_us'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'lp', d['log_prob_xs_per_node'])
"
done
