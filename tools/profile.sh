#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats + separate PMC passes (run on the GPU box).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r1}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_stamp())" > $out/source_stamp.txt
EXTRA="${@:2}"   # e.g. --workload config2_attn
CMD="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --latency-steps 0 $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o kt -- $CMD > $out/bench_under_rocprof.log 2>&1
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  d=$out/pmc_$(echo $p | cut -d" " -f1)
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d $d -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --latency-steps 0 --kernel-timing-steps 1 $EXTRA > $d.log 2>&1
done
python $R/bench.py --steps 200 --warmup 20 $EXTRA > $out/bench.json 2> $out/bench.err
ls -R $out | head -40
