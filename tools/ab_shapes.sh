#!/bin/bash
# A/B the (MT,NETS) workgroup shapes of the fused kernel through GNF_FORCE_SHAPE (developer tool).
cd "$(dirname "$0")/.."
for sh in 12 11 21 22; do
  GNF_FORCE_SHAPE=$sh python bench.py --steps 30 --warmup 5 --no-cpu-baseline --kernel-timing-steps 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('shape $sh', 'half_step_us', d['roofline']['kernel_us'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'lp', d['log_prob_xs_per_node'])
"
done
