#!/bin/bash
# PMC passes (rocprofv3 --pmc, one counter group per run) of the standalone GEMM harness.
# usage: scripts/pmc_gemm.sh <shapes-file> <out-prefix>   (run on the GPU box, from the repo root)
set -u
SHAPES=${1:-scripts/gemm_shapes_pmc.txt}
OUT=${2:-gpurun_out/pmc}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  GB_ITERS=2 GB_ROUNDS=1 timeout 120 rocprofv3 --pmc $group -d /tmp/pmc_$i -o p --output-format csv -- \
      "$ROOT/scripts/probe/_probe_gemm_bench" "$ROOT/$SHAPES" > "$ROOT/$OUT/pass$i.log" 2>&1
  f=$(find /tmp/pmc_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$ROOT/$OUT/pass$i.csv"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
GROUPS
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(out + '/pass*.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(mkg')[0][-60:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
with open(out + '/summary.txt', 'w') as fo:
    for k in agg:
        if 'gemm' not in k: continue
        fo.write(k + '\n')
        for c in sorted(agg[k]):
            fo.write(f'  {c:32s} {agg[k][c]/max(cnt[k][c],1):18.0f}  (avg of {cnt[k][c]})\n')
print(open(out + '/summary.txt').read())
PY
