#!/bin/bash
# experiment builds of the library with TIMING-ONLY ablations of flash_fwd8_kernel (wrong results):
#   scripts/probe/_probe_f8_<n>/libmacaw_hip.so, n = 0 (shipped), 1 no softmax segment, 2 no MFMAs, 3 no LDS fragment
#   reads, 4 no K / V staging, 5 no barriers;   LD_LIBRARY_PATH=scripts/probe/_probe_f8_<n> scripts/probe/_probe_attn_fwd
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result"
for n in "$@"; do
  ( out=$root/scripts/probe/_probe_f8_$n; mkdir -p $out
    X="-DMK_F8_ABLATE=$n"
    case $n in p0) X="-DMK_F8_PRIO=0" ;; p1) X="-DMK_F8_PRIO=1" ;; p2) X="-DMK_F8_PRIO=2" ;; l0) X="-DMK_F8_LATE=0" ;; l1) X="-DMK_F8_LATE=1" ;; l2) X="-DMK_F8_LATE=2" ;; m1) X="-DMK_F8_MIXED=1" ;; m0) X="-DMK_F8_MIXED=0" ;; m1a1) X="-DMK_F8_MIXED=1 -DMK_F8_ABLATE=1" ;; v5) X="-DMK_F8_VPG=5" ;; v7) X="-DMK_F8_VPG=7" ;; v9) X="-DMK_F8_VPG=9" ;; v12) X="-DMK_F8_VPG=12" ;; v7p0) X="-DMK_F8_VPG=7 -DMK_F8_PRIO=0" ;; esac     # priority variants (correct results)
    hipcc $F $X -c $root/macaw_llm_amd/csrc/attention.hip -o $out/attention.o
    objs=$(ls $root/macaw_llm_amd/csrc/_obj/*.o | grep -v "attention.o")
    hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libmacaw_hip.so $objs $out/attention.o
    rm -f $out/attention.o ) &
done
wait
hipcc -O2 --offload-arch=gfx950 $root/scripts/probe/attn_fwd_probe.cpp -o $root/scripts/probe/_probe_attn_fwd -L$root/macaw_llm_amd -lmacaw_hip
ls -la $root/scripts/probe/_probe_f8_*/
