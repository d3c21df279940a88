out=gpurun_out/r05/b5; mkdir -p $out
for i in 1 2; do for m in 0 1 2; do
  MK_GEMM_V9=$m MACAW_GEMM_REPORT=$out/shapes_v9mode${m}_$i.csv timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_v9mode${m}_$i.json 2> $out/bench_v9mode${m}_$i.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05/b5/bench_*.json")):
    for line in open(f):
        if line.startswith("{"):
            d=json.loads(line); r=d["roofline"]
            print(f.split("/")[-1], d["value"], d["ms_per_step"], r["gemm_ms_per_step"], r["achieved"], r["frac"])
PY
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py -q -x --timeout 300 -p no:cacheprovider > $out/t.log 2>&1; tail -3 $out/t.log
