for c in 64 128 256 512; do
  export GNF_DW_CHUNK=$c
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_dw_$c -- python /root/repo/bench.py --workload config2_train --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk', $c, d['value'], d['ms_per_step'])"
  cd /root/repo; python tools/kstats.py gpurun_out/prof_dw_$c 6 | grep -E "dw_grouped|reduce_grouped|bwd_fused"
done
