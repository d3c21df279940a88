#!/bin/bash
python scripts/bench_align_attn.py 2>&1 | grep -v "transformers\|^  - \|amdgpu.ids"
