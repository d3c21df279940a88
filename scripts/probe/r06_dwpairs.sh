#!/bin/bash
# cfg 3: which dx / dW pairs gain from two streams?  one pair at a time (MACAW_DW_STREAM=1 MACAW_DW_PAIRS=<pair>), alternated with off
out=$1
for i in 1 2; do
  for p in off o down qkv gu lm; do
    if [ $p = off ]; then export MACAW_DW_STREAM=0; unset MACAW_DW_PAIRS; else export MACAW_DW_STREAM=1 MACAW_DW_PAIRS=$p; fi
    timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_${p}_$i.json 2> $out/bench_${p}_$i.err
    python3 -c "
import json
d=json.load(open('$out/bench_${p}_$i.json'))
print('cfg3 pair=$p', d['value'], d['ms_per_step'])"
  done
done
unset MACAW_DW_STREAM MACAW_DW_PAIRS
