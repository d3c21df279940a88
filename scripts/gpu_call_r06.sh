#!/bin/bash
# one gpurun call of round 6: everything is written under gpurun_out/r06/<tag>/
# usage: scripts/gpu_call_r06.sh <tag> <step> [<step> ...]
set -u
tag=$1; shift
R=$PWD
out=$R/gpurun_out/r06/$tag
mkdir -p $out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
GB=$R/scripts/probe/_probe_gemm_bench
for step in "$@"; do
  t0=$(date +%s)
  case $step in
    vendor)  # which hipBLASLt kernel runs per step shape (kernel trace of the harness with cfg 100 only)
      (cd /tmp && GB_ITERS=5 GB_ROUNDS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/vk -o v --output-format csv -- \
         $GB $R/scripts/gemm_shapes_vendor_only.txt > $out/vendor_only.csv 2> $out/vendor_only.err)
      t=$(find /tmp/vk -name '*kernel_trace.csv' | head -1)
      [ -n "$t" ] && python3 $R/scripts/trace_kernels_by_grid.py "$t" Cijk > $out/vendor_kernels_by_grid.txt
      rm -rf /tmp/vk ;;
    benchq)
      timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err ;;
    bench)
      timeout 600 python bench.py --steps 20 --warmup 3 > $out/bench_cfg3_full.json 2> $out/bench_cfg3_full.err ;;
    tests)
      timeout 2400 python -m pytest tests -m gpu -q -rf --timeout 300 --durations=12 -p no:cacheprovider > $out/tests.log 2>&1
      echo "pytest rc=$?" >> $out/tests.log ;;
    *) if [ -f "$R/scripts/probe/r06_$step.sh" ]; then bash $R/scripts/probe/r06_$step.sh $out > $out/$step.log 2>&1; else echo "unknown step $step"; fi ;;
  esac
  echo "[$step] $(( $(date +%s) - t0 )) s"
done
