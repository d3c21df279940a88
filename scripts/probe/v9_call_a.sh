set -u
out=gpurun_out/r05/b1; mkdir -p $out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
GB=scripts/probe/_probe_gemm_bench
GB_ITERS=3 GB_ROUNDS=1 timeout 300 $GB scripts/gemm_shapes_v9_check.txt > $out/v9_check.csv 2> $out/v9_check.err
echo "check rc=$?" >> $out/v9_check.err
for i in 1 2; do GB_ITERS=10 GB_ROUNDS=3 timeout 200 $GB scripts/gemm_shapes_v9_kslope.txt > $out/v9_kslope_$i.csv 2>> $out/v9.err; done
GB_COLD=1 GB_ITERS=10 GB_ROUNDS=3 timeout 400 $GB scripts/gemm_shapes_v9_step.txt > $out/v9_step_cold.csv 2>> $out/v9.err
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "v9" -q -rf -x --timeout 300 -p no:cacheprovider > $out/t_v9.log 2>&1
echo "pytest rc=$?" >> $out/t_v9.log
cat $out/v9_check.csv; tail -5 $out/t_v9.log; cat $out/v9_kslope_1.csv $out/v9_step_cold.csv
