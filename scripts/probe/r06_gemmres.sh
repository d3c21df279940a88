#!/bin/bash
python scripts/bench_gemm_residual.py 2>&1 | grep -v "transformers\|^  - \|amdgpu.ids"
