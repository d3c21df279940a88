#!/bin/bash
GB_COLD=1 GB_ITERS=20 GB_ROUNDS=3 scripts/probe/_probe_gemm_bench scripts/gemm_shapes_decode_ksplit.txt
