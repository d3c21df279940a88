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
    v9)      # gemm_v9 in the harness: checked against the fp32 reference rows, K-slope and cold step shapes beside v7 / vendor
      GB_ITERS=3 GB_ROUNDS=1 timeout 300 $GB scripts/gemm_shapes_v9_check.txt > $out/v9_check.csv 2> $out/v9_check.err
      for i in 1 2; do GB_ITERS=10 GB_ROUNDS=3 timeout 200 $GB scripts/gemm_shapes_v9_kslope.txt > $out/v9_kslope_$i.csv 2>> $out/v9.err; done
      GB_COLD=1 GB_ITERS=10 GB_ROUNDS=3 timeout 400 $GB scripts/gemm_shapes_v9_step.txt > $out/v9_step_cold.csv 2>> $out/v9.err
      timeout 900 python -m pytest tests/test_kernels_gpu.py -k "v9" -q -rf -x --timeout 300 -p no:cacheprovider > $out/t_v9.log 2>&1
      echo "pytest rc=$?" >> $out/t_v9.log ;;
    v9pmc)   # cycles, not seconds: v7 / v8 / v9 / vendor and v9's timing-only ablations (scripts/probe/build_v9_variants.sh first)
      bash scripts/probe/v9_pmc_ablate.sh $out ${V9VARS:-nodma noread nodma_noread nobar} > $out/v9_pmc_summary.txt 2>&1 ;;
    v9mode)  # in-step A/B of the kernel policy: MK_GEMM_V9 = 0 (v7 only) / 1 (policy) / 2 (v9 wherever legal), alternated
      for i in 1 2; do for m in 0 1 2; do
        MK_GEMM_V9=$m MACAW_GEMM_REPORT=$out/shapes_v9mode${m}_$i.csv timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline \
          > $out/bench_v9mode${m}_$i.json 2> $out/bench_v9mode${m}_$i.err
      done; done ;;
    attn)    # attention micro-benchmark, the attention tests, PMC of the S = 2048 kernels
      timeout 300 python scripts/bench_attn.py > $out/attn.txt 2>&1
      MK_ATTN_DQ_ASYNC=1 timeout 300 python scripts/bench_attn.py > $out/attn_dqasync.txt 2>&1
      timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -k "attn or attention or flash or forward_backward" -q -x --timeout 300 -p no:cacheprovider > $out/t_attn.log 2>&1
      echo "pytest rc=$?" >> $out/t_attn.log
      bash scripts/probe/attn_pmc.sh ${tag}_pmc > $out/attn_pmc_summary.txt 2>&1 ;;
    localov) # one rank: per-bucket AdamW behind the backward (opt-in) against the one fused launch, alternated
      for i in 1 2; do
        timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_serial_$i.json 2> $out/bench_serial_$i.err
        MACAW_LOCAL_OVERLAP=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_overlap_$i.json 2> $out/bench_overlap_$i.err
      done ;;
    overlap2) # the collective path on one rank, default group vs the high-priority group, comm_cus 0 / 8 / 16 / 32
      bash scripts/probe/overlap_1rank_r05.sh > $out/overlap2.log 2>&1 ;;
    attn8)   # the 8-wave forward: its own tests first (a failure or a hang switches the kernel off for the rest of the call),
             # then the micro-benchmark with both forward kernels alternated in one process
      timeout 240 python -m pytest tests/test_kernels_gpu.py -k "eight_wave or lazy_rescale" -q -rf -x --timeout 200 -p no:cacheprovider > $out/t_attn8.log 2>&1
      rc=$?; echo "pytest rc=$rc" >> $out/t_attn8.log
      if [ $rc -ne 0 ]; then export MK_ATTN_FWD8_MIN=0; echo "fwd8 OFF for the rest of the call" >> $out/t_attn8.log; fi
      timeout 300 python scripts/bench_attn.py > $out/attn8.txt 2>&1 ;;
    bench2s) # cfg 2 with the per-shape GEMM report (the table was never committed)
      MACAW_GEMM_REPORT=$out/gemm_shapes_per_step_cfg2.csv timeout 600 python bench.py --config 2 --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_cfg2.err ;;
    gen)     # generate(): B = 1 / 8 / 16 / 32 from ONE run
      timeout 600 python scripts/bench_generate.py 1 8 16 32 > $out/generate.txt 2>&1 ;;
    tpins)   # the tightened bf16 generate() pin and the world-2 tests with the stage-digest diagnosis
      timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_train_gpu.py -k "generate_at_7b or two_ranks" -q -s -rf --timeout 600 -p no:cacheprovider > $out/t_pins.log 2>&1
      echo "pytest rc=$?" >> $out/t_pins.log ;;
    attnpmc) # PMC of the S = 2048 attention kernels (the 8-wave forward included)
      bash scripts/probe/attn_pmc.sh ${tag}_pmc > $out/attn_pmc_summary.txt 2>&1 ;;
    bench4ab) # cfg 4 in-step A/B of the forward attention kernels (4-wave vs 8-wave at S = 2048), alternated twice
      for i in 1 2; do
        MK_ATTN_FWD8_MIN=0 timeout 600 python bench.py --config 4 --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_cfg4_fwd4_$i.json 2> $out/bench_cfg4_fwd4_$i.err
        MK_ATTN_FWD8_MIN=1024 timeout 600 python bench.py --config 4 --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_cfg4_fwd8_$i.json 2> $out/bench_cfg4_fwd8_$i.err
      done
      python - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_cfg4_fwd*_?.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], r["attention_fwd"], r["attention_bwd"])
    except Exception as e:
        print(f, "unreadable", e)
PY
      ;;
    *) echo "unknown step $step" ;;
  esac
  echo "$step: $(( $(date +%s) - t0 )) s" >> $out/timing.txt
done
tail -5 $out/t_*.log $out/tests*.log 2>/dev/null
cat $out/timing.txt
