set -u
out=gpurun_out/r05/a2; mkdir -p $out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { # tag, env...
  tag=$1; shift
  env "$@" MACAW_FORCE_COLLECTIVES=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_$tag.json 2> $out/bench_$tag.err
  (cd /tmp && env "$@" MACAW_FORCE_COLLECTIVES=1 timeout 400 rocprofv3 --kernel-trace -d /tmp/ov_$tag -o t --output-format csv -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$out/trace_$tag.log 2>&1)
  f=$(find /tmp/ov_$tag -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python scripts/trace_last_step.py "$f" > $out/last_step_$tag.txt 2>&1
  rm -rf /tmp/ov_$tag
}
run normal_cus16 MACAW_COMM_NORMAL_PRIORITY=1 MACAW_COMM_CUS=16
run high_cus0 MACAW_COMM_CUS=0
run high_cus8 MACAW_COMM_CUS=8
run high_cus16 MACAW_COMM_CUS=16
run high_cus32 MACAW_COMM_CUS=32
run high_cus16_q8 MACAW_COMM_CUS=16 GPU_MAX_HW_QUEUES=8
for f in $out/last_step_*.txt; do echo $f; head -3 $f; done
