#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "weight_norm or golden or ragged" 2>&1 | tail -3
bash tools/gpu_jobs/attn_prof.sh r03s 2>&1 | grep -v "^\"\|amdgpu.ids" | tail -30
