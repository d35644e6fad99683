#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "live_k_tiles" --tb=short 2>&1 | grep -E "^E |passed|failed|test_kernels_gpu.py:[0-9]+" | head -12; done
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_decode_trainer_gpu.py tests/test_fullsize_gpu.py -m gpu -q 2>&1 | tail -6
