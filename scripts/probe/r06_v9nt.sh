#!/bin/bash
# gemm_v9's C stores with the non-temporal hint (experiment build -DMK_V9_NT_STORE) against the shipped build, cfg 3, one box:
# default build -> bench x2; rebuild with the flag -> bench x2; rebuild default -> bench x1
out=$1
run() { python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"; }
run default; run default
MK_EXTRA_FLAGS=-DMK_V9_NT_STORE python -m macaw_llm_amd.build --force 2>&1 | tail -1
export MK_EXTRA_FLAGS=-DMK_V9_NT_STORE      # (the stamp of the build covers the flags: keep them while this build is loaded)
run nt_store; run nt_store
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "v9" -p no:cacheprovider 2>&1 | grep -a "passed\|failed" | tail -1
unset MK_EXTRA_FLAGS
python -m macaw_llm_amd.build --force 2>&1 | tail -1
run default
