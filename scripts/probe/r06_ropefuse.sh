#!/bin/bash
# RoPE inside the short-sequence attention kernels vs the separate launches: the new parity tests, the model-level
# tests, then the cfg-3 step alternated in fresh processes on this box (MACAW_ROPE_FUSE = bwd | full | off)
out=$1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "rope or short" -p no:cacheprovider 2>&1 | tail -5
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|error" | tail -5
for i in 1 2; do
  for m in bwd full off; do
    export MACAW_ROPE_FUSE=$m
    MACAW_GEMM_REPORT=$out/shapes_${m}_$i.csv timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_${m}_$i.json 2> $out/bench_${m}_$i.err
    python3 -c "
import json,sys
d=json.load(open('$out/bench_${m}_$i.json'))
r=d['roofline']
print('$m', d['value'], d['ms_per_step'], r['frac'], r.get('attention_fwd',{}).get('ms_per_step'), r.get('attention_bwd',{}).get('ms_per_step'))"
  done
done
unset MACAW_ROPE_FUSE
