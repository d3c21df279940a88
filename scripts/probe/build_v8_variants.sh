#!/bin/bash
# experiment build of the library with the schedule variants / ablations of gemm_v8 compiled in (-DMK_V8_VARIANTS;
# chosen at run time with MK_GEMM_V8_VAR, see csrc/gemm_v8_impl.inc).  Output: scripts/probe/_probe_v8var/libmacaw_hip.so
#   LD_LIBRARY_PATH=scripts/probe/_probe_v8var MK_GEMM_V8_VAR=3 scripts/probe/_probe_gemm_bench <shapes>
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/scripts/probe/_probe_v8var
mkdir -p $out
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result"
hipcc $F -DMK_V8_VARIANTS -mllvm -pragma-unroll-threshold=1000000 -c $root/macaw_llm_amd/csrc/gemm_v8.hip -o $out/gemm_v8.o
objs=$(ls $root/macaw_llm_amd/csrc/_obj/*.o | grep -v "gemm_v8.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libmacaw_hip.so $objs $out/gemm_v8.o
rm -f $out/*.o
ls -la $out
