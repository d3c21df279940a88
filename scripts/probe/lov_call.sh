out=gpurun_out/r05/$1; mkdir -p $out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_hf_trainer_gpu.py -q -x --timeout 600 -p no:cacheprovider > $out/t_train.log 2>&1; grep -n "passed\|failed" $out/t_train.log; tail -5 $out/t_train.log | grep -v "^$" | head -5
for i in 1 2; do
  MACAW_NO_LOCAL_OVERLAP=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_serial_$i.json 2> $out/bench_serial_$i.err
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_overlap_$i.json 2> $out/bench_overlap_$i.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_*.json")):
    for line in open(f):
        if line.startswith("{"):
            d=json.loads(line); r=d["roofline"]
            print(f.split("/")[-1], d["value"], d["ms_per_step"], r["gemm_ms_per_step"], r["frac"], r["whole_step_frac"], d["comm"]["tail_after_backward_ms"])
PY
