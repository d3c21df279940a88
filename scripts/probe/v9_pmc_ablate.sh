#!/bin/bash
# cycles (not seconds) of gemm_v9 and its timing-only ablations, v7 / v8 / vendor beside it: K-slope from two K
# usage (on the GPU box): scripts/probe/v9_pmc_ablate.sh <outdir> [variants...]
set -u
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
GB=$PWD/scripts/probe/_probe_gemm_bench
printf '4096 4096 4096 0 11 14 15 100\n4096 4096 16384 0 11 14 15 100\n' > /tmp/pmc_base.txt
printf '4096 4096 4096 0 15\n4096 4096 16384 0 15\n' > /tmp/pmc_var.txt
run() {  # tag, libdir-or-empty, shapes
  tag=$1; lib=$2; shapes=$3
  (cd /tmp && LD_LIBRARY_PATH=${lib:+$lib:}${LD_LIBRARY_PATH:-} GB_ITERS=3 GB_ROUNDS=1 timeout 200 rocprofv3 \
     --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
     -d /tmp/pmc_$tag -o p --output-format csv -- $GB $shapes > $OLDPWD/$out/pmc_$tag.log 2>&1)
  f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $out/pmc_$tag.csv
  (cd /tmp && LD_LIBRARY_PATH=${lib:+$lib:}${LD_LIBRARY_PATH:-} GB_ITERS=3 GB_ROUNDS=1 timeout 200 rocprofv3 --kernel-trace \
     -d /tmp/kt_$tag -o k --output-format csv -- $GB $shapes > $OLDPWD/$out/kt_$tag.log 2>&1)
  f=$(find /tmp/kt_$tag -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && cp $f $out/kt_$tag.csv
  rm -rf /tmp/pmc_$tag /tmp/kt_$tag
}
run base "" /tmp/pmc_base.txt
for v in "$@"; do run $v $PWD/scripts/probe/_probe_v9_$v /tmp/pmc_var.txt; done
python scripts/probe/v9_pmc_reduce.py $out
