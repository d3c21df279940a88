out=gpurun_out/r05/$1; mkdir -p $out
timeout 300 python scripts/bench_attn.py > $out/attn.txt 2>&1; cat $out/attn.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -k "attn or attention or flash or forward_backward" -q -x --timeout 300 -p no:cacheprovider > $out/t_attn.log 2>&1; grep -n "passed\|failed" $out/t_attn.log
