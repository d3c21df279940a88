#!/bin/bash
# experiment builds of the library with TIMING-ONLY ablations of gemm_v9's generated K loop (wrong results):
#   scripts/probe/_probe_v9_<name>/libmacaw_hip.so for name in nodma noread nodma_noread nobar
#   LD_LIBRARY_PATH=scripts/probe/_probe_v9_nodma scripts/probe/_probe_gemm_bench <shapes>
# read their numbers in CYCLES (rocprofv3 --pmc GRBM_GUI_ACTIVE): without the memory traffic the chip clocks higher.
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result -mllvm -pragma-unroll-threshold=1000000"
build() {   # name, generator options...
  name=$1; shift
  X=""
  if [ "$name" = noepi ] || [ "$name" = mfma16_noepi ]; then X="-DMK_V9_NOEPI"; fi
  if [ "$name" = nostore ]; then X="-DMK_EPI_NOSTORE"; fi
  out=$root/scripts/probe/_probe_v9_$name
  mkdir -p $out/src
  cp $root/macaw_llm_amd/csrc/*.h $root/macaw_llm_amd/csrc/*.inc $root/macaw_llm_amd/csrc/gemm_v9.hip $out/src/
  mkdir -p $out/include && cp $root/include/macaw_hip.h $out/include/
  sed -i 's#"../../include/macaw_hip.h"#"../include/macaw_hip.h"#' $out/src/gemm_common.h
  python $root/scripts/gen_v9_loop.py "$@" $out/src/gemm_v9_loop.inc
  hipcc $F $X -c $out/src/gemm_v9.hip -o $out/gemm_v9.o
  objs=$(ls $root/macaw_llm_amd/csrc/_obj/*.o | grep -v "gemm_v9.o")
  hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libmacaw_hip.so $objs $out/gemm_v9.o
  rm -rf $out/src $out/include $out/*.o
}
for v in "$@"; do
  case $v in
    nodma) build nodma --nodma & ;;
    noread) build noread --noread & ;;
    nodma_noread) build nodma_noread --nodma --noread & ;;
    nobar) build nobar --nobar & ;;
    noepi) build noepi & ;;
    nostore) build nostore & ;;
    mfma16) build mfma16 --mfma16 & ;;                  # round 6: layout NT on v_mfma_f32_16x16x32 (CORRECT results)
    mfma16_noepi) build mfma16_noepi --mfma16 & ;;      # ... timing-only, no epilogue (K-slope against `noepi`)
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done
wait
ls -la $root/scripts/probe/_probe_v9_*/
