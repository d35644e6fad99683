#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "attention" -q -x 2>&1 | tail -40 > $O/pytest_attn.txt
cat $O/pytest_attn.txt
python tools/bench_attn.py 3 > $O/bench_attn.txt 2>&1
cat $O/bench_attn.txt
