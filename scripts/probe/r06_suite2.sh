#!/bin/bash
# the full GPU suite on the tree with the rope entry points and the grad-weight side stream, then cfg 2 / cfg 3 bench lines
out=$1
timeout 2400 python -m pytest tests -m gpu -q -rf --timeout 300 --durations=8 -p no:cacheprovider > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
tail -15 $out/tests.log
timeout 400 python bench.py --config 2 --steps 10 --warmup 3 > $out/bench_cfg2.json 2> $out/bench_cfg2.err
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err
python3 - <<P
import json
for c in (2,3):
    d=json.load(open('$out/bench_cfg%d.json'%c))
    print('cfg',c,d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('whole_step_frac'))
P
