#!/bin/bash
# one gpurun call of round 4: everything is written under gpurun_out/r04/<tag>/
# usage: scripts/gpu_call_r04.sh <tag> <step> [<step> ...]
set -u
tag=$1; shift
out=gpurun_out/r04/$tag
mkdir -p $out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
GB=scripts/probe/_probe_gemm_bench
for step in "$@"; do
  t0=$(date +%s)
  case $step in
    v8)      # v7 / v8 / vendor yardstick on cubes, step shapes and edges
      GB_ITERS=10 GB_ROUNDS=3 timeout 600 $GB scripts/gemm_shapes_v8.txt > $out/gemm_v8.csv 2> $out/gemm_v8.err
      GB_COLD=1 GB_ITERS=10 GB_ROUNDS=2 timeout 600 $GB scripts/gemm_shapes_v8.txt > $out/gemm_v8_cold.csv 2>> $out/gemm_v8.err ;;
    trwait)  # v7 reduction-major layouts with / without the compiler's vmcnt(0) before transpose reads
      for i in 1 2; do
        LD_LIBRARY_PATH=scripts/probe/_probe_trwait GB_ITERS=10 GB_ROUNDS=2 timeout 300 $GB scripts/gemm_shapes_trwait.txt > $out/trwait_compilerwait_$i.csv 2>> $out/trwait.err
        GB_ITERS=10 GB_ROUNDS=2 timeout 300 $GB scripts/gemm_shapes_trwait.txt > $out/trwait_nowait_$i.csv 2>> $out/trwait.err
      done ;;
    v8var)   # v8 schedule variants and timing-only ablations (scripts/probe/build_v8_variants.sh)
      for i in 1 2; do for v in ${V8VARS:-1 0 3 4 5}; do
        LD_LIBRARY_PATH=scripts/probe/_probe_v8var MK_GEMM_V8_VAR=$v GB_ITERS=10 GB_ROUNDS=2 timeout 120 $GB scripts/gemm_shapes_v8var.txt > $out/v8var_${v}_$i.csv 2>> $out/v8var.err
      done; done
      printf '4096 4096 4096 0 11 100\n8192 8192 8192 0 11 100\n4608 12288 4096 0 11 100\n16384 8192 4096 0 11 100\n' > /tmp/ref_shapes.txt
      GB_ITERS=10 GB_ROUNDS=2 timeout 120 $GB /tmp/ref_shapes.txt > $out/v8var_ref.csv 2>> $out/v8var.err ;;
    v8pmc)   # cycles, not seconds (DVFS): GRBM / SQ counters + kernel durations of v7, v8 variants and the vendor kernel
      printf '8192 8192 8192 0 11 14 100\n' > /tmp/pmc_shape.txt
      for v in ${V8VARS:-1 3}; do
        (cd /tmp && LD_LIBRARY_PATH=$OLDPWD/scripts/probe/_probe_v8var MK_GEMM_V8_VAR=$v GB_ITERS=3 GB_ROUNDS=1 timeout 120 rocprofv3 \
           --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
           -d /tmp/pmcv_$v -o p --output-format csv -- $OLDPWD/$GB /tmp/pmc_shape.txt > $OLDPWD/$out/pmc_var$v.log 2>&1)
        f=$(find /tmp/pmcv_$v -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $out/pmc_var$v.csv
        (cd /tmp && LD_LIBRARY_PATH=$OLDPWD/scripts/probe/_probe_v8var MK_GEMM_V8_VAR=$v GB_ITERS=3 GB_ROUNDS=1 timeout 120 rocprofv3 \
           --kernel-trace -d /tmp/ktv_$v -o k --output-format csv -- $OLDPWD/$GB /tmp/pmc_shape.txt > $OLDPWD/$out/kt_var$v.log 2>&1)
        f=$(find /tmp/ktv_$v -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && cp $f $out/kt_var$v.csv
      done ;;
    kslope)
      for i in 1 2; do GB_ITERS=10 GB_ROUNDS=3 timeout 200 $GB scripts/gemm_shapes_kslope.txt > $out/kslope_$i.csv 2>> $out/kslope.err; done ;;
    kslopevar)   # the K-slope of v8 schedule variants (experiment build)
      for v in ${V8VARS:-1 2}; do
        LD_LIBRARY_PATH=scripts/probe/_probe_v8var MK_GEMM_V8_VAR=$v GB_ITERS=10 GB_ROUNDS=3 timeout 200 $GB scripts/gemm_shapes_kslope.txt > $out/kslope_var$v.csv 2>> $out/kslope.err
      done ;;
    tnew)    # the tests added in round 4
      timeout 1500 python -m pytest tests/test_hf_trainer_gpu.py tests/test_train_gpu.py "tests/test_fullsize_gpu.py::test_full_llama7b_fp32_engine_within_1e_3_of_the_fp32_oracle_on_gpu" "tests/test_fullsize_gpu.py::test_generate_at_7b_dimensions_fp32_ids_bit_exact_vs_the_restated_greedy_loop" "tests/test_fullsize_gpu.py::test_full_llama7b_against_fp32_oracle_on_gpu" -q -rf -s --timeout 600 --durations=8 -p no:cacheprovider > $out/t_new.log 2>&1
      echo "pytest rc=$?" >> $out/t_new.log ;;
    thf)
      timeout 600 python -m pytest tests/test_hf_trainer_gpu.py -q -rf --timeout 600 -p no:cacheprovider > $out/t_hf.log 2>&1
      echo "pytest rc=$?" >> $out/t_hf.log ;;
    rccl1)   # the collective path through a 1-rank RCCL group: bench line with `comm`, then a kernel trace of it
      MACAW_FORCE_COLLECTIVES=1 timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_cfg3_1rank_rccl.json 2> $out/bench_cfg3_1rank_rccl.err
      (cd /tmp && MACAW_FORCE_COLLECTIVES=1 timeout 500 rocprofv3 --kernel-trace -d /tmp/rccl1 -o t --output-format csv -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$out/rccl1_trace.log 2>&1)
      f=$(find /tmp/rccl1 -name '*kernel_trace.csv' | head -1)
      [ -n "$f" ] && python scripts/trace_last_step.py "$f" > $out/cfg3_1rank_rccl_last_step.txt 2>&1 ;;
    cuhold)  # GEMMs beside a resident kernel holding 8 / 16 / 32 CUs, planned for 256 or for the free CUs
      timeout 300 scripts/probe/_probe_cu_hold > $out/cu_hold.csv 2> $out/cu_hold.err ;;
    epi)     # timing-only epilogue ablations of v7 (scripts/probe/build_epi_variants.sh): what overlap could buy
      for i in 1 2; do
        GB_COLD=1 GB_ITERS=10 GB_ROUNDS=2 timeout 120 $GB scripts/gemm_shapes_epi.txt > $out/epi_base_$i.csv 2>> $out/epi.err
        for v in ${EPIVARS:-0 3 4}; do
          LD_LIBRARY_PATH=scripts/probe/_probe_epi$v GB_COLD=1 GB_ITERS=10 GB_ROUNDS=2 timeout 120 $GB scripts/gemm_shapes_epi.txt > $out/epi_v${v}_$i.csv 2>> $out/epi.err
        done
      done ;;
    walk)    # walking workgroups with the early prologue vs one workgroup per tile (MK_GEMM_NO_WALK), same library
      for i in 1 2 3; do
        GB_COLD=1 GB_ITERS=10 GB_ROUNDS=2 timeout 120 $GB scripts/gemm_shapes_walk.txt > $out/walk_on_$i.csv 2>> $out/walk.err
        MK_GEMM_NO_WALK=1 GB_COLD=1 GB_ITERS=10 GB_ROUNDS=2 timeout 120 $GB scripts/gemm_shapes_walk.txt > $out/walk_off_$i.csv 2>> $out/walk.err
      done ;;
    enc)
      GB_COLD=1 GB_ITERS=10 GB_ROUNDS=3 timeout 300 $GB scripts/gemm_shapes_enc.txt > $out/gemm_enc_cold.csv 2> $out/gemm_enc.err ;;
    attn)    # attention micro-benchmark (async staging vs the synchronous dq), then the attention / model tests
      timeout 200 python scripts/bench_attn.py > $out/attn_async.txt 2>&1
      MK_ATTN_DQ_ASYNC=1 timeout 200 python scripts/bench_attn.py > $out/attn_dqasync.txt 2>&1
      timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -k "attn or attention or flash or forward_backward" -q -rf --timeout 300 -p no:cacheprovider > $out/t_attn.log 2>&1
      echo "pytest rc=$?" >> $out/t_attn.log ;;
    adamw)
      for v in 0 2 1 0 2; do MK_ADAMW_VAR=$v timeout 120 python scripts/bench_adamw_multi.py 2>&1 | grep MK_ADAMW >> $out/adamw_ab.txt; done
      timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py -k "adamw or bucketed or optimizer or graphed" -q -rf --timeout 300 -p no:cacheprovider > $out/t_adamw.log 2>&1
      echo "pytest rc=$?" >> $out/t_adamw.log ;;
    attn2)   # XCD-grouped grids of the attention forward: non-causal (default on) and causal (A/B)
      timeout 200 python scripts/bench_attn.py > $out/attn_grouped.txt 2>&1
      MK_ATTN_NO_XCD_GROUP=1 timeout 200 python scripts/bench_attn.py > $out/attn_ungrouped.txt 2>&1
      MK_ATTN_XCD_GROUP_CAUSAL=1 timeout 200 python scripts/bench_attn.py > $out/attn_grouped_causal.txt 2>&1
      timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -k "attn or attention or flash or forward_backward" -q -rf --timeout 300 -p no:cacheprovider > $out/t_attn.log 2>&1
      echo "pytest rc=$?" >> $out/t_attn.log ;;
    shortbwd)  # one-kernel backward for Lq == Lk <= 160, hd 128 vs prep + dq + dkv (MK_ATTN_NO_SHORT_BWD), then the tests
      timeout 200 python scripts/bench_attn.py > $out/attn_short.txt 2>&1
      MK_ATTN_NO_SHORT_BWD=1 MK_ATTN_NO_SHORT_FWD=1 timeout 200 python scripts/bench_attn.py > $out/attn_3kernel.txt 2>&1
      timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -k "attn or attention or flash or forward_backward" -q -rf --timeout 300 -p no:cacheprovider > $out/t_attn.log 2>&1
      echo "pytest rc=$?" >> $out/t_attn.log ;;
    mix1)    # round-4 late changes: M-edge idle half, causal sub-block skip, RoPE index arithmetic
      timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -rf --timeout 300 -p no:cacheprovider > $out/t_mix.log 2>&1
      echo "pytest rc=$?" >> $out/t_mix.log
      timeout 200 python scripts/bench_attn.py > $out/attn.txt 2>&1
      GB_COLD=1 GB_ITERS=10 GB_ROUNDS=3 timeout 300 $GB scripts/gemm_shapes_enc.txt > $out/gemm_enc_cold.csv 2> $out/gemm_enc.err
      timeout 300 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_cfg2.err
      timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err ;;
    decode)  # decode attention for many (sample, head) pairs: four heads per workgroup vs one (A/B), tests
      timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py -k "decode or generate" -q -rf --timeout 300 -p no:cacheprovider > $out/t_decode.log 2>&1
      echo "pytest rc=$?" >> $out/t_decode.log
      timeout 300 python scripts/bench_generate.py 16 32 2>&1 | grep "B=" > $out/generate_attn4.txt
      MK_DECODE_ATTN_NO4=1 timeout 300 python scripts/bench_generate.py 16 32 2>&1 | grep "B=" > $out/generate_attn1.txt ;;
    tgemm)
      timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -k "gemm or fp8" -q -rf --timeout 240 -p no:cacheprovider > $out/t_gemm.log 2>&1
      echo "pytest rc=$?" >> $out/t_gemm.log ;;
    tests)
      timeout 1200 python -m pytest tests -m gpu -q -rf --timeout 300 --durations=12 -p no:cacheprovider > $out/tests.log 2>&1
      echo "pytest rc=$?" >> $out/tests.log ;;
    bench)
      timeout 600 python bench.py --steps 8 --warmup 3 > $out/bench_cfg3.json 2> $out/bench_cfg3.err ;;
    benchq)
      timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err ;;
    pcsamp)  # instruction-level stall evidence: stochastic PC sampling of the v7 / v8 main loops (beta feature: bounded)
      printf '4608 12288 4096 0 11 14\n4608 4096 12288 1 11 14\n' > /tmp/pcs_shapes.txt
      (cd /tmp && GB_ITERS=30 GB_ROUNDS=1 timeout 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic \
         --pc-sampling-unit cycles --pc-sampling-interval 65536 --kernel-trace -d /tmp/pcs -o pcs --output-format csv \
         -- $OLDPWD/$GB /tmp/pcs_shapes.txt > $OLDPWD/$out/pcsamp.log 2>&1; echo "rc=$?" >> $OLDPWD/$out/pcsamp.log)
      ls -la /tmp/pcs/* >> $out/pcsamp.log 2>&1
      for f in $(find /tmp/pcs -name '*.csv' -size -40M 2>/dev/null); do cp $f $out/ 2>/dev/null; done ;;
    *) echo "unknown step $step" ;;
  esac
  echo "$step: $(( $(date +%s) - t0 )) s" >> $out/timing.txt
done
tail -5 $out/t_*.log $out/tests*.log 2>/dev/null
cat $out/timing.txt
