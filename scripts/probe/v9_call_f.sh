out=gpurun_out/r05/b6; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "v9" -q -x --timeout 300 -p no:cacheprovider > $out/t_v9.log 2>&1; grep -n "passed\|failed" $out/t_v9.log
GB=scripts/probe/_probe_gemm_bench
GB_COLD=1 GB_ITERS=10 GB_ROUNDS=3 timeout 400 $GB scripts/gemm_shapes_v9_step.txt > $out/v9_step_cold.csv 2>> $out/err.txt
for i in 1 2; do for m in 0 1 2; do
  MK_GEMM_V9=$m MACAW_GEMM_REPORT=$out/shapes_v9mode${m}_$i.csv timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_v9mode${m}_$i.json 2> $out/bench_v9mode${m}_$i.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05/b6/bench_*.json")):
    for line in open(f):
        if line.startswith("{"):
            d=json.loads(line); r=d["roofline"]
            print(f.split("/")[-1], d["value"], d["ms_per_step"], r["gemm_ms_per_step"], r["achieved"], r["frac"])
PY
cat $out/v9_step_cold.csv
