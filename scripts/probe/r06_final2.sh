#!/bin/bash
# end-of-round set, second edition (tree with the side stream, the fp8 fusions, the rope entry points): full GPU suite, then
# scripts/profile_round.sh r06 bench prof traffic decode parity
out=$1
timeout 2400 python -m pytest tests -m gpu -q -rf --timeout 300 --durations=8 -p no:cacheprovider > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
tail -4 $out/tests.log
cp $out/tests.log gpurun_out/r06_gpu_tests.log
bash scripts/profile_round.sh r06 bench prof traffic decode parity > $out/profile_round.log 2>&1
tail -45 $out/profile_round.log
