#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y; mkdir -p $O
python tools/bench_gemm_splitk.py > $O/splitk.txt 2>&1; cat $O/splitk.txt
