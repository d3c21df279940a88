#!/bin/bash
# gemm_v9 NT on 16x16x32 MFMAs (experiment build, scripts/probe/build_v9_variants.sh mfma16 noepi mfma16_noepi)
GB=scripts/probe/_probe_gemm_bench
for i in 1 2; do
  echo "== shipped library (32x32x16), round $i"; GB_ITERS=10 GB_ROUNDS=3 $GB scripts/gemm_shapes_v9_mfma16.txt
  echo "== mfma16 variant, round $i"; LD_LIBRARY_PATH=$PWD/scripts/probe/_probe_v9_mfma16:${LD_LIBRARY_PATH:-} GB_ITERS=10 GB_ROUNDS=3 $GB scripts/gemm_shapes_v9_mfma16.txt
done
echo "== timing only, no epilogue: shipped loop"; LD_LIBRARY_PATH=$PWD/scripts/probe/_probe_v9_noepi:${LD_LIBRARY_PATH:-} GB_ITERS=10 GB_ROUNDS=3 $GB scripts/gemm_shapes_v9_mfma16.txt 2>/dev/null
echo "== timing only, no epilogue: mfma16 loop"; LD_LIBRARY_PATH=$PWD/scripts/probe/_probe_v9_mfma16_noepi:${LD_LIBRARY_PATH:-} GB_ITERS=10 GB_ROUNDS=3 $GB scripts/gemm_shapes_v9_mfma16.txt 2>/dev/null
