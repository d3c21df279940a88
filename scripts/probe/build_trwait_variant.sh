#!/bin/bash
# A/B partner for the `tr_read_nw` finding (csrc/common.h): the same library with the transpose reads of the
# 256 x 256 tile kernels left to the compiler's LDS-DMA ordering (an `s_waitcnt vmcnt(0)` in front of the first
# ds_read_b64_tr_b16 of every K-tile).  Output: scripts/probe/_probe_trwait/libmacaw_hip.so -- run the harness with
#   LD_LIBRARY_PATH=scripts/probe/_probe_trwait scripts/probe/_probe_gemm_bench <shapes>
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/macaw_llm_amd $tmp/include
cp -r $root/macaw_llm_amd/csrc $tmp/macaw_llm_amd/csrc
cp $root/include/*.h $tmp/include/
sed -i 's/tr_read_nw(const char\* __restrict__ lds)/tr_read_nw(const char* lds)/' $tmp/macaw_llm_amd/csrc/common.h
out=$root/scripts/probe/_probe_trwait
mkdir -p $out
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result"
hipcc $F -c $tmp/macaw_llm_amd/csrc/gemm_v7.hip -o $out/gemm_v7.o &
hipcc $F -mllvm -pragma-unroll-threshold=1000000 -c $tmp/macaw_llm_amd/csrc/gemm_v8.hip -o $out/gemm_v8.o &
wait
objs=$(ls $root/macaw_llm_amd/csrc/_obj/*.o | grep -v "gemm_v7.o\|gemm_v8.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libmacaw_hip.so $objs $out/gemm_v7.o $out/gemm_v8.o
rm -rf $tmp $out/*.o
ls -la $out
