out=gpurun_out/r05/b3; mkdir -p $out
GB=scripts/probe/_probe_gemm_bench
for i in 1 2; do
for v in base noepi nostore nodma_noread; do
  L=""; [ $v != base ] && L=$PWD/scripts/probe/_probe_v9_$v
  LD_LIBRARY_PATH=$L:${LD_LIBRARY_PATH:-} GB_ITERS=20 GB_ROUNDS=3 timeout 100 $GB scripts/gemm_shapes_v9_fixed.txt > $out/fixed_${v}_$i.csv 2>> $out/err.txt
done; done
printf '4096 4096 1024 0 11 14 100\n4096 4096 2048 0 11 14 100\n4096 4096 4096 0 11 14 100\n' > /tmp/f.txt
GB_ITERS=20 GB_ROUNDS=3 timeout 100 $GB /tmp/f.txt > $out/fixed_others.csv 2>> $out/err.txt
tail -n +1 $out/fixed_*.csv
