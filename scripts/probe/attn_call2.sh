out=gpurun_out/r05/$1; mkdir -p $out
timeout 300 python scripts/bench_attn.py > $out/attn.txt 2>&1; cat $out/attn.txt
MK_ATTN_DQ_ASYNC=1 timeout 300 python scripts/bench_attn.py > $out/attn_dqasync.txt 2>&1; grep bwd $out/attn_dqasync.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -k "attn or attention or flash or forward_backward" -q -x --timeout 300 -p no:cacheprovider > $out/t_attn.log 2>&1; grep -n "passed\|failed" $out/t_attn.log
MK_ATTN_DQ_ASYNC=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -k "flash and bwd" -q -x --timeout 300 -p no:cacheprovider > $out/t_attn_async.log 2>&1; grep -n "passed\|failed" $out/t_attn_async.log
