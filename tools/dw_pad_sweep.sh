# developer sweep: dynamic-LDS padding of the dW GEMM (keeps its workgroups off the CUs the fused backward kernel fills)
for c in 0 24000 52000 70000; do
  export GNF_DW_LDS_PAD=$c
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_pad_$c -- python /root/repo/bench.py --workload config2_train --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pad', $c, d['value'], d['ms_per_step'])"
  cd /root/repo; python tools/kstats.py gpurun_out/prof_pad_$c 6 | grep -E "dw_grouped|reduce_grouped|bwd_fused|aggregate_bwd"
  python bench.py --workload config2_train --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   no-profiler', d['ms_per_step'])"
done
