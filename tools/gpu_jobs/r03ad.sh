#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short 2>&1 | grep -E "^E |passed|failed|test_kernels_gpu.py:[0-9]+|parity_util|^tests/" | head -30
