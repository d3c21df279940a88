#!/bin/bash
# A/B partner and timing-only ablations of the 256 x 256 kernel's epilogue (3 / 4: results WRONG by construction):
#   scripts/probe/_probe_epi0/libmacaw_hip.so   every tile through the LDS-staged epilogue (rounds 1-3 and early round 4)
#   scripts/probe/_probe_epi3/libmacaw_hip.so   the workgroup ends after its last K-tile (no staging, no stores)
#   scripts/probe/_probe_epi4/libmacaw_hip.so   staging + arithmetic, but no global stores
# They price what an epilogue that overlaps the next tile's main loop could buy at most.  Run the harness with
#   LD_LIBRARY_PATH=scripts/probe/_probe_epi3 scripts/probe/_probe_gemm_bench <shapes>
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result"
for v in 0 3 4; do
  tmp=$(mktemp -d)
  mkdir -p $tmp/macaw_llm_amd $tmp/include
  cp -r $root/macaw_llm_amd/csrc $tmp/macaw_llm_amd/csrc
  cp $root/include/*.h $tmp/include/
  h=$tmp/macaw_llm_amd/csrc/gemm_common.h
  sed -i 's|if (g.c_vec == 2 \&\& n0 + BN7 <= g.N) wave_epilogue_direct|if (g.alpha == 12345.f) wave_epilogue_direct|' $tmp/macaw_llm_amd/csrc/gemm_v7_impl.inc
  if cmp -s $tmp/macaw_llm_amd/csrc/gemm_v7_impl.inc $root/macaw_llm_amd/csrc/gemm_v7_impl.inc; then echo "direct-path pattern not found"; exit 1; fi
  if [ $v = 0 ]; then :
  elif [ $v = 3 ]; then
    sed -i 's|^  __syncthreads();                  // every wave is done with the operand tiles in LDS|  __syncthreads(); if (g.alpha != 12345.f) return;|' $h
  else
    sed -i 's|^          \*reinterpret_cast<e16x8\*>(C + (long)m \* g.ldc + ncol) = o;|          if (g.alpha == 12345.f) *reinterpret_cast<e16x8*>(C + (long)m * g.ldc + ncol) = o;|' $h
  fi
  if [ $v != 0 ] && cmp -s $h $root/macaw_llm_amd/csrc/gemm_common.h; then echo "variant $v: pattern not found"; exit 1; fi
  out=$root/scripts/probe/_probe_epi$v
  mkdir -p $out
  hipcc $F -c $tmp/macaw_llm_amd/csrc/gemm_v7.hip -o $out/gemm_v7.o
  objs=$(ls $root/macaw_llm_amd/csrc/_obj/*.o | grep -v "gemm_v7.o")
  hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libmacaw_hip.so $objs $out/gemm_v7.o
  rm -rf $tmp $out/*.o
done
ls -la $root/scripts/probe/_probe_epi0 $root/scripts/probe/_probe_epi3 $root/scripts/probe/_probe_epi4
