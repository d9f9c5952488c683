#!/bin/bash
# A/B sweep of the weight-gradient GEMM launch shapes on a training workload (run on the GPU box).
# usage: dw_sweep.sh [workload] variant...   variants: grouped grouped_serial auto auto_serial unitsN unitsN_serial ldsB
cd /tmp && export TMPDIR=/tmp
R=/root/repo
W=config2_train
case $1 in *_train) W=$1; shift ;; esac
run() {  # name, env...
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dw_$name -- python $R/bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/dw_$name.json 2>/dev/null
  echo "== $name: $(python -c "import json;d=json.loads(open('$R/gpurun_out/dw_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms/step"
  python $R/tools/kstats.py $R/gpurun_out/dw_$name 8 | grep -E "gemm_dw|half_bwd|reduce_grouped"
}
[ $# -eq 0 ] && set -- grouped auto
for v in "$@"; do
  case $v in
    grouped) run $v GNF_DW_GROUPED=1 ;;
    grouped_serial) run $v GNF_DW_GROUPED=1 GNF_TRAIN_NO_OVERLAP=1 ;;
    auto) run $v X=1 ;;
    latefork) run $v GNF_DW_LATE_FORK=1 ;;
    lib_*) run $v GNF_LIB_PATH=$R/graph-normalizing-flows_amd/variants/libgnf_${v#lib_}.so ;;
    auto_serial) run $v GNF_TRAIN_NO_OVERLAP=1 ;;
    units*_serial) c=${v#units}; c=${c%_serial}; run $v GNF_TRAIN_NO_OVERLAP=1 GNF_DW_WIDE_UNITS=$c ;;
    units*) c=${v#units}; run $v GNF_DW_WIDE_UNITS=$c ;;
    lds*) c=${v#lds}; run $v GNF_DW_WIDE_LDS=$c ;;
  esac
done
