#!/bin/bash
# one gpurun call of round 5: everything is written under gpurun_out/r05/<tag>/
# usage: scripts/gpu_call_r05.sh <tag> <step> [<step> ...]
set -u
tag=$1; shift
out=gpurun_out/r05/$tag
mkdir -p $out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
GB=scripts/probe/_probe_gemm_bench
for step in "$@"; do
  t0=$(date +%s)
  case $step in
    overlap)   # the collective path on ONE rank with the GEMMs planned for 256 - comm_cus CUs.
               # per setting: the bench line (comm object) un-profiled, then a kernel trace reduced to its last step
      for cus in ${OVERLAP_CUS:-0 8 16 32}; do
        MACAW_FORCE_COLLECTIVES=1 MACAW_COMM_CUS=$cus timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline \
          > $out/bench_cfg3_1rank_rccl_cus$cus.json 2> $out/bench_cfg3_1rank_rccl_cus$cus.err
        (cd /tmp && MACAW_FORCE_COLLECTIVES=1 MACAW_COMM_CUS=$cus timeout 400 rocprofv3 --kernel-trace -d /tmp/ov_$cus -o t --output-format csv \
           -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$out/overlap_trace_cus$cus.log 2>&1)
        f=$(find /tmp/ov_$cus -name '*kernel_trace.csv' | head -1)
        [ -n "$f" ] && python scripts/trace_last_step.py "$f" > $out/cfg3_1rank_rccl_cus${cus}_last_step.txt 2>&1
        rm -rf /tmp/ov_$cus
      done ;;
    benchq)
      timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err ;;
    bench)
      timeout 600 python bench.py --steps 20 --warmup 3 > $out/bench_cfg3_full.json 2> $out/bench_cfg3_full.err ;;
    bench2)
      timeout 600 python bench.py --config 2 --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_cfg2.err ;;
    bench4)
      timeout 900 python bench.py --config 4 --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err ;;
    thf)
      timeout 900 python -m pytest tests/test_hf_trainer_gpu.py -q -rf --timeout 600 -p no:cacheprovider > $out/t_hf.log 2>&1
      echo "pytest rc=$?" >> $out/t_hf.log ;;
    tattn)
      timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -k "attn or attention or flash or forward_backward" -q -rf --timeout 300 -p no:cacheprovider > $out/t_attn.log 2>&1
      echo "pytest rc=$?" >> $out/t_attn.log ;;
    tgemm)
      timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -k "gemm or fp8" -q -rf --timeout 240 -p no:cacheprovider > $out/t_gemm.log 2>&1
      echo "pytest rc=$?" >> $out/t_gemm.log ;;
    tests)
      timeout 1500 python -m pytest tests -m gpu -q -rf --timeout 300 --durations=12 -p no:cacheprovider > $out/tests.log 2>&1
      echo "pytest rc=$?" >> $out/tests.log ;;
    v9)      # the hand-placed 4-wave K loop: parity vs v7 in the harness (cfg 15 checks against cfg 11), K-slope, step shapes
      GB_CHECK=1 GB_ITERS=3 GB_ROUNDS=1 timeout 300 $GB scripts/gemm_shapes_v9_check.txt > $out/v9_check.csv 2> $out/v9_check.err
      for i in 1 2; do GB_ITERS=10 GB_ROUNDS=3 timeout 200 $GB scripts/gemm_shapes_v9_kslope.txt > $out/v9_kslope_$i.csv 2>> $out/v9.err; done
      GB_COLD=1 GB_ITERS=10 GB_ROUNDS=3 timeout 400 $GB scripts/gemm_shapes_v9_step.txt > $out/v9_step_cold.csv 2>> $out/v9.err ;;
    v9pmc)   # cycles, not seconds: v7 / v8 / v9 / vendor on one cube
      printf '8192 8192 8192 0 11 14 15 100\n' > /tmp/pmc_shape.txt
      (cd /tmp && GB_ITERS=3 GB_ROUNDS=1 timeout 120 rocprofv3 \
         --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
         -d /tmp/pmc9 -o p --output-format csv -- $OLDPWD/$GB /tmp/pmc_shape.txt > $OLDPWD/$out/v9_pmc.log 2>&1)
      f=$(find /tmp/pmc9 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $out/v9_pmc.csv
      (cd /tmp && GB_ITERS=3 GB_ROUNDS=1 timeout 120 rocprofv3 --kernel-trace -d /tmp/kt9 -o k --output-format csv \
         -- $OLDPWD/$GB /tmp/pmc_shape.txt > $OLDPWD/$out/v9_kt.log 2>&1)
      f=$(find /tmp/kt9 -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && cp $f $out/v9_kt.csv ;;
    attn)
      timeout 200 python scripts/bench_attn.py > $out/attn.txt 2>&1 ;;
    *) echo "unknown step $step" ;;
  esac
  echo "$step: $(( $(date +%s) - t0 )) s" >> $out/timing.txt
done
tail -5 $out/t_*.log $out/tests*.log 2>/dev/null
cat $out/timing.txt
