#!/bin/bash
# A/B of the large-batch kernel's closing round (big_plan) over batch sizes (developer tool, run on the GPU box):
#   tools/ab_tail.sh <workload> "<graphs per GPU list>" "<force_shape:fused_variant list>"
# force_shape 0 = the automatic rule, 22 = the 32-row both-nets shape, 40 = the large-batch kernel;
# fused_variant 64 = whole 4-tile workgroups only (no closing round of small ones)
cd "$(dirname "$0")/.."
wl=${1:-config4}
for g in ${2:-128 160 192 224 256}; do
  for sv in ${3:-0:0 22:0 40:0 40:64}; do
    sh=${sv%%:*}; v=${sv##*:}
    GNF_OPTIONS="force_shape=$sh,fused_variant=$v" python bench.py --workload $wl --graphs-per-gpu $g --steps 10 --warmup 3 --prewarm-ms 50 --no-cpu-baseline \
        --no-secondary --latency-steps 0 --kernel-timing-steps 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$wl graphs $g nodes', d['config']['nodes_total'], 'shape $sh variant $v', 'half_step_us', d['roofline']['kernel_us'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])
"
  done
done
