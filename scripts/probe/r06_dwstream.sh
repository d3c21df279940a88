#!/bin/bash
# grad-weight GEMMs on a second stream beside the grad-input GEMMs (engine.DW_SIDE): model tests with it on, then the
# cfg-2 and cfg-3 steps alternated in fresh processes on this box
out=$1
MACAW_DW_STREAM=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|error" | tail -5
for i in 1 2; do
  for c in 2 3; do
    for m in side plain; do
      if [ $m = side ]; then export MACAW_DW_STREAM=1; else unset MACAW_DW_STREAM; fi
      timeout 300 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_cfg${c}_${m}_$i.json 2> $out/bench_cfg${c}_${m}_$i.err
      python3 -c "
import json
d=json.load(open('$out/bench_cfg${c}_${m}_$i.json'))
print('cfg$c $m', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('peak_mem_gib'))"
    done
  done
done
unset MACAW_DW_STREAM
