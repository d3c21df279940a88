out=gpurun_out/r05/b4; mkdir -p $out
GB=scripts/probe/_probe_gemm_bench
printf '4096 4096 1024 0 11 15 100\n4096 4096 4096 0 11 15 100\n4096 4096 16384 0 11 15 100\n' > /tmp/f.txt
for i in 1 2; do
  GB_ITERS=20 GB_ROUNDS=3 timeout 100 $GB /tmp/f.txt > $out/epi_reg_$i.csv 2>> $out/err.txt
  MK_V9_LDS_EPI=1 GB_ITERS=20 GB_ROUNDS=3 timeout 100 $GB /tmp/f.txt > $out/epi_lds_$i.csv 2>> $out/err.txt
done
GB_COLD=1 GB_ITERS=10 GB_ROUNDS=3 timeout 400 $GB scripts/gemm_shapes_v9_step.txt > $out/v9_step_cold.csv 2>> $out/err.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -k "v9" -q -rf -x --timeout 300 -p no:cacheprovider > $out/t_v9.log 2>&1
tail -3 $out/t_v9.log
tail -n +1 $out/epi_*.csv $out/v9_step_cold.csv
