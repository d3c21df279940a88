out=gpurun_out/r05/$1; mkdir -p $out
cat > /tmp/attn_one.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from macaw_llm_amd import ops
dev = torch.device("cuda:0")
B, H, S, hd, causal = 4, 32, 2048, 128, True
D = H * hd
g = torch.Generator(device="cpu").manual_seed(1)
mk = lambda: (torch.randn(B, S, D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
q, k, v, do = mk(), mk(), mk(), mk()
o = torch.empty_like(q); dq = torch.empty_like(q); dk = torch.empty_like(q); dv = torch.empty_like(q)
lse = torch.empty(B * H * S, dtype=torch.float32, device=dev)
a = (B, H, S, S, hd, D, S * D, D, S * D, D, S * D, D, S * D, hd ** -0.5)
for _ in range(3):
    ops.flash_attn_fwd(q, k, v, o, *a, causal=causal, lse=lse)
    ops.flash_attn_bwd(q, k, v, o, do, lse, dq, dk, dv, *a, causal=causal)
torch.cuda.synchronize()
PY
cd /tmp
for pass in 1 2; do
  if [ $pass = 1 ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE";
  else C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; fi
  timeout 200 rocprofv3 --pmc $C -d /tmp/apmc$pass -o p --output-format csv -- python /tmp/attn_one.py > $GRAFT_REPO_ROOT/$out/pmc$pass.log 2>&1
  f=$(find /tmp/apmc$pass -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$out/attn_pmc$pass.csv
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, sys, statistics, glob
for f in sorted(glob.glob(sys.argv[1] + "/attn_pmc*.csv") if len(sys.argv) > 1 else []):
    pass
PY
python scripts/probe/attn_pmc_reduce.py $out
