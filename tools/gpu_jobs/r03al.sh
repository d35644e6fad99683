#!/bin/bash
cd $GRAFT_REPO_ROOT
CST_GEMM_EXPERIMENT=1 python tools/bench_gemm_cfg.py 2>&1 | tail -16
bash tools/gpu_jobs/full_tests.sh r03d 2>&1 | tail -12 | cut -c1-300
