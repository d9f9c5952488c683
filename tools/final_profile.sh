#!/bin/bash
# Round-end evidence (on the GPU box), two phases because bench.py quotes counter traffic only from a
# profiles/pmc_traffic.json that was taken on the build's own kernel sources:
#   tools/final_profile.sh pmc <tag> [workloads]   PMC / kernel-stat passes (tools/pmc_shape.sh) of the forward workloads (kernel A
#                                       reported next to the dominant kernel), the training workloads and the wide layers
#     -> here: for w in ...; do python tools/summarize_profile.py --pmc-shape gpurun_out/pmc_<tag>_$w $w; done; commit
#   tools/final_profile.sh bench <tag>  one un-profiled bench line per workload + the driver-shaped default run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
phase=${1:-pmc}; tag=${2:-r4z}
cd $R
if [ "$phase" = pmc ]; then
  only=${3:-}   # optional: a comma-separated list of workloads (re-measuring the ones whose kernels changed)
  want() { [ -z "$only" ] || case ",$only," in *",$1,"*) true;; *) false;; esac; }
  for w in config2 config4 config5 config2_attn default_flags; do
    want $w && timeout 600 bash tools/pmc_shape.sh ${tag}_$w $w "" k_aggregate > gpurun_out/pmc_${tag}_$w.log 2>&1
  done
  for w in config2_train default_flags_train; do
    want $w && timeout 600 bash tools/pmc_shape.sh ${tag}_$w $w "" k_half_fused,k_half_bwd,k_reduce,k_attn,k_adam > gpurun_out/pmc_${tag}_$w.log 2>&1
  done
  want wide_fc && timeout 600 bash tools/pmc_shape.sh ${tag}_wide_fc wide_fc "" k_linear_short,k_gemm,k_splitk,k_aggregate,k_coupling > gpurun_out/pmc_${tag}_wide_fc.log 2>&1
  # the data driver's literal defaults (train_grevnet_with_data.py:40-46, 100-117): forward and one trainer step; the wide MLPs' trainer step alone
  want data_default_flags && timeout 600 bash tools/pmc_shape.sh ${tag}_data_default_flags data_default_flags "" k_linear_short,k_attn,k_gemm,k_bn,k_coupling,k_splitk > gpurun_out/pmc_${tag}_data_default_flags.log 2>&1
  for w in data_default_flags_train wide_fc_train; do
    want $w && timeout 900 bash tools/pmc_shape.sh ${tag}_$w $w "" k_linear_short,k_gemm,k_attn,k_adam,k_pack,k_bn,k_coupling,k_aggregate > gpurun_out/pmc_${tag}_$w.log 2>&1
  done
else
  for w in config2_fc config2_attn default_flags config4 config5 wide_fc config2_train default_flags_train data_default_flags data_default_flags_train wide_fc_train; do
    timeout 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --latency-steps 0 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_$w.json
  done
  ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) 2> gpurun_out/${tag}_bench_config2_default_run.time | tail -1 > gpurun_out/${tag}_bench_config2_default_run.json
fi
ls gpurun_out | grep $tag
