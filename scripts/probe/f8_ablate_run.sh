# usage (on the GPU box, from the repo root): bash scripts/probe/f8_ablate_run.sh <tag> <variant> ...
cd $GRAFT_REPO_ROOT; tag=$1; shift; mkdir -p gpurun_out/r05/$tag
for rep in 1 2; do for n in "$@"; do
  echo "== variant $n (rep $rep)"
  LD_LIBRARY_PATH=scripts/probe/_probe_f8_$n timeout 60 scripts/probe/_probe_attn_fwd 4 32 2048 128 1 | tail -1
  LD_LIBRARY_PATH=scripts/probe/_probe_f8_$n timeout 60 scripts/probe/_probe_attn_fwd 4 32 2048 128 0 | tail -1
done; done > gpurun_out/r05/$tag/ablate.txt 2>&1
cat gpurun_out/r05/$tag/ablate.txt
