#!/bin/bash
# (1) decode weight layout probe; (2) frozen audio tower on a second stream beside the image tower (engine.ENC_SIDE):
# model tests with it on, then cfg 3 alternated in fresh processes
out=$1
scripts/probe/_probe_decode_weight_layout > $out/decode_weight_layout.csv 2>&1
cat $out/decode_weight_layout.csv
MACAW_ENC_STREAMS=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|error" | tail -5
for i in 1 2; do
  for m in side plain; do
    if [ $m = side ]; then export MACAW_ENC_STREAMS=1; else unset MACAW_ENC_STREAMS; fi
    timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_cfg3_${m}_$i.json 2> $out/bench_cfg3_${m}_$i.err
    python3 -c "
import json
d=json.load(open('$out/bench_cfg3_${m}_$i.json'))
print('cfg3 $m', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
unset MACAW_ENC_STREAMS
