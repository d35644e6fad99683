#!/bin/bash
cd $GRAFT_REPO_ROOT
CST_GEMM_EXPERIMENT=1 python tools/bench_gemm_cfg.py 2>&1 | tail -16
