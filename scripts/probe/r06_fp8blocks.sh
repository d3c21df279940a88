#!/bin/bash
python scripts/fp8_block_scale_experiment.py real_13b 2>&1 | grep -v "transformers\|^  - \|amdgpu.ids"
