#!/bin/bash
# one rank: per-bucket AdamW behind the backward CONFINED to a few CUs (MACAW_LOCAL_CUS) against the one fused launch, alternated
out=$1
run() { # tag env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_$tag.json 2> $out/bench_$tag.err
  python3 - $out/bench_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:28s} {d['value']:8.2f} samples/s {d['ms_per_step']:8.2f} ms  gemm {d['roofline']['gemm_ms_per_step']:.1f} ms tail {d['comm']['tail_after_backward_ms']}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for i in 1 2; do
  run serial_$i X=1
  for c in ${LOCAL_CUS_SWEEP:-0 8 16 24 32 48}; do
    run ov_cus${c}_$i MACAW_LOCAL_OVERLAP=1 MACAW_LOCAL_CUS=$c
  done
done
