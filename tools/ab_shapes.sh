#!/bin/bash
# A/B the launch shapes of the fused kernel over batch sizes (developer tool, run on the GPU box):
#   tools/ab_shapes.sh <workload> "<graphs per GPU list>" "<force_shape list; 0 = automatic>"
# e.g. tools/ab_shapes.sh config4 "16 32 64 128 256" "0 12 22 20 40"
cd "$(dirname "$0")/.."
wl=${1:-config4}
for g in ${2:-16 32 64 128 256}; do
  for sh in ${3:-0 22 20 40}; do
    GNF_OPTIONS="force_shape=$sh" python bench.py --workload $wl --graphs-per-gpu $g --steps 10 --warmup 3 --prewarm-ms 50 --no-cpu-baseline \
        --no-secondary --latency-steps 0 --kernel-timing-steps 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$wl graphs $g nodes', d['config']['nodes_total'], 'shape $sh', 'half_step_us', d['roofline']['kernel_us'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])
"
  done
done
