out=gpurun_out/r05/$1; mkdir -p $out
timeout 900 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err
python - <<PY
import json
for line in open("$out/bench_cfg4.json"):
    if line.startswith("{"):
        d=json.loads(line); r=d["roofline"]
        print("cfg4", d["value"], d["ms_per_step"], r["frac"], r.get("attention_fwd"), r.get("attention_bwd"))
PY
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -q -x --timeout 600 -p no:cacheprovider -k "cfg4 or seq_2048 or whisper or clip or encoder" > $out/t_full.log 2>&1; grep -n "passed\|failed" $out/t_full.log
