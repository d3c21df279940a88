#!/bin/bash
# A/B partner of the 16-bit staged epilogue of the 256 x 256 kernel: the same library with every tile through the
# fp32-staged form.  Output: scripts/probe/_probe_nolean/libmacaw_hip.so
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/macaw_llm_amd $tmp/include
cp -r $root/macaw_llm_amd/csrc $tmp/macaw_llm_amd/csrc
cp $root/include/*.h $tmp/include/
python3 - $tmp/macaw_llm_amd/csrc/gemm_common.h <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = "if constexpr (!SV && FN <= 2) {"
assert s.count(old) == 1
open(p, "w").write(s.replace(old, "if constexpr (false) {"))
PY
out=$root/scripts/probe/_probe_nolean
mkdir -p $out
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result"
hipcc $F -c $tmp/macaw_llm_amd/csrc/gemm_v7.hip -o $out/gemm_v7.o
objs=$(ls $root/macaw_llm_amd/csrc/_obj/*.o | grep -v "gemm_v7.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libmacaw_hip.so $objs $out/gemm_v7.o
rm -rf $tmp $out/*.o
ls -la $out
