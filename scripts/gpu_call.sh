#!/bin/bash
# one gpurun call of round 3: everything is written under gpurun_out/r03/<tag>/
# usage: scripts/gpu_call.sh <tag> <step> [<step> ...]   steps: probe tests bench flaky
set -u
tag=$1; shift
out=gpurun_out/r03/$tag
mkdir -p $out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for step in "$@"; do
  t0=$(date +%s)
  case $step in
    probe)
      timeout 300 python scripts/probe/gloo_cuda_race.py 150 0 > $out/gloo_probe_idle.txt 2>&1
      timeout 400 python scripts/probe/gloo_cuda_race.py 150 1 > $out/gloo_probe_burner.txt 2>&1 ;;
    tests)
      timeout 900 python -m pytest tests -m gpu -q -rf --timeout 240 --durations=12 -p no:cacheprovider > $out/tests.log 2>&1
      echo "pytest rc=$?" >> $out/tests.log ;;
    tests_s)
      timeout 900 python -m pytest tests -m gpu -q -rf -s --timeout 240 --durations=12 -p no:cacheprovider > $out/tests.log 2>&1
      echo "pytest rc=$?" >> $out/tests.log ;;
    tests_fast)   # everything but the full-size file
      timeout 600 python -m pytest tests -m gpu -q -rf -s --timeout 240 --durations=12 -p no:cacheprovider \
        --ignore=tests/test_fullsize_gpu.py > $out/tests_fast.log 2>&1
      echo "pytest rc=$?" >> $out/tests_fast.log ;;
    cfg5)
      timeout 400 python bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_cfg5_fp8.json 2> $out/bench_cfg5_fp8.err
      timeout 400 python bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline --no-fp8 > $out/bench_cfg5_bf16.json 2> $out/bench_cfg5_bf16.err
      timeout 400 python bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline --fp8-mlp > $out/bench_cfg5_fp8mlp.json 2> $out/bench_cfg5_fp8mlp.err ;;
    splitk)
      timeout 300 python scripts/probe/splitk_stress.py 3000 2 0 > $out/splitk_idle.txt 2>&1
      timeout 400 python scripts/probe/splitk_stress.py 3000 3 1 > $out/splitk_burner.txt 2>&1 ;;
    tailab)
      for i in 1 2; do
        MK_GEMM_NO_TAIL8=1 scripts/probe/_probe_gemm_bench scripts/gemm_shapes_tail.txt > $out/tail_quarters_$i.csv 2> $out/tail_q.err
        scripts/probe/_probe_gemm_bench scripts/gemm_shapes_tail.txt > $out/tail_eighths_$i.csv 2> $out/tail_e.err
      done ;;
    cfg5one)
      timeout 400 python bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_cfg5_fp8.json 2> $out/bench_cfg5_fp8.err ;;
    cfg4)
      timeout 400 python bench.py --config 4 --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err ;;
    bench)
      timeout 600 python bench.py --steps 8 --warmup 3 > $out/bench_cfg3.json 2> $out/bench_cfg3.err ;;
    exp1)   # round-3 late experiments: 17-32-row decode kernels, six-stage eighth tail, RMSNorm-backward prefetch
      timeout 500 python -m pytest tests/test_kernels_gpu.py -k "gemm or rmsnorm" -x -q -rf --timeout 240 -p no:cacheprovider > $out/t_kern.log 2>&1
      echo "pytest rc=$?" >> $out/t_kern.log
      MK_GEMM_TAIL8_NS6=1 timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -k "v7_256 or fp8" -x -q -rf --timeout 240 -p no:cacheprovider > $out/t_ns6.log 2>&1
      echo "pytest rc=$?" >> $out/t_ns6.log
      for i in 1 2; do
        scripts/probe/_probe_gemm_bench scripts/gemm_shapes_tail.txt > $out/tail_ns5_$i.csv 2> $out/tail.err
        MK_GEMM_TAIL8_NS6=1 scripts/probe/_probe_gemm_bench scripts/gemm_shapes_tail.txt > $out/tail_ns6_$i.csv 2>> $out/tail.err
      done
      GB_COLD=1 scripts/probe/_probe_gemm_bench scripts/gemm_shapes_decode32.txt > $out/decode32_cold.csv 2> $out/decode32.err
      scripts/probe/_probe_gemm_bench scripts/gemm_shapes_decode32.txt > $out/decode32_warm.csv 2>> $out/decode32.err
      timeout 120 python scripts/bench_norm.py > $out/norm_prefetch.txt 2>&1
      MK_RMSNORM_BWD_NO_PREFETCH=1 timeout 120 python scripts/bench_norm.py > $out/norm_r2.txt 2>&1
      BG_SKINNY32=19,20,21,22 timeout 400 python scripts/bench_generate.py 32 24 > $out/generate_skinny32.txt 2>&1
      timeout 300 python bench.py --config 2 --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_cfg2.err
      timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err
      MK_GEMM_TAIL8_NS6=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_cfg3_ns6.json 2> $out/bench_cfg3_ns6.err
      timeout 400 python bench.py --config 5 --steps 3 --warmup 2 --no-cpu-baseline > $out/bench_cfg5.json 2> $out/bench_cfg5.err ;;
    exp2)   # decode: kernel per N at 17-32 rows, non-temporal weight loads; RMSNorm-backward block sweep
      timeout 300 python -m pytest tests/test_kernels_gpu.py -k "skinny or rmsnorm or decode" -x -q -rf --timeout 240 -p no:cacheprovider > $out/t_kern.log 2>&1
      echo "pytest rc=$?" >> $out/t_kern.log
      MK_DECODE_W_NT=1 timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -k "skinny or decode or generate" -x -q -rf --timeout 240 -p no:cacheprovider > $out/t_nt.log 2>&1
      echo "pytest rc=$?" >> $out/t_nt.log
      for nt in 0 1; do
        MK_DECODE_W_NT=$nt GB_COLD=1 scripts/probe/_probe_gemm_bench scripts/gemm_shapes_decode32.txt > $out/decode32_cold_nt$nt.csv 2> $out/decode32.err
        MK_DECODE_W_NT=$nt GB_COLD=1 scripts/probe/_probe_gemm_bench scripts/gemm_shapes_decode.txt > $out/decode_cold_nt$nt.csv 2>> $out/decode32.err
      done
      timeout 120 python scripts/bench_norm.py > $out/norm_sweep.txt 2>&1
      BG_W_NT=0,1 timeout 400 python scripts/bench_generate.py 1 8 16 32 > $out/generate_nt.txt 2>&1
      export TMPDIR=/tmp
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/dec32 -o d --output-format csv -- python $OLDPWD/scripts/decode_once.py 32 > /dev/null 2>&1)
      f=$(find /tmp/dec32 -name '*kernel_trace.csv' | head -1)
      [ -n "$f" ] && python scripts/decode_trace.py "$f" > $out/decode_step_B32.txt 2>&1
      timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err ;;
    final)  # end-of-round evidence from the final binary: the full GPU suite, then profile_round.sh r03f
      timeout 900 python -m pytest tests -m gpu -q -rf -s --timeout 240 --durations=8 -p no:cacheprovider > gpurun_out/r03f_gpu_tests.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/r03f_gpu_tests.log
      bash scripts/profile_round.sh r03f bench prof traffic3 decode > gpurun_out/r03f_profile.log 2>&1 ;;
    flaky)
      timeout 900 python scripts/probe/flaky_dp.py 8 > $out/flaky.txt 2>&1 ;;
    *) echo "unknown step $step" ;;
  esac
  echo "$step: $(( $(date +%s) - t0 )) s" >> $out/timing.txt
done
tail -5 $out/tests*.log 2>/dev/null
cat $out/timing.txt
