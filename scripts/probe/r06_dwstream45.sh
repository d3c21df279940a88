#!/bin/bash
# the grad-weight side stream forced on at cfg 4 (M = 8192) and cfg 5 (13B, D = 5120: 360 / 1080 / 972 tiles = fractional rounds)
# against "auto" (off there); plus the new world-2 test
out=$1
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "side_stream" -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|error" | tail -3
for c in 5 4; do
  for i in 1 2; do
    for m in side auto; do
      if [ $m = side ]; then export MACAW_DW_STREAM=1; else unset MACAW_DW_STREAM; fi
      timeout 500 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_cfg${c}_${m}_$i.json 2> $out/bench_cfg${c}_${m}_$i.err
      python3 -c "
import json
d=json.load(open('$out/bench_cfg${c}_${m}_$i.json'))
print('cfg$c $m', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('peak_mem_gib'))"
    done
  done
done
unset MACAW_DW_STREAM
