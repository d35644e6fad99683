#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
(python tools/bench_attn_packed.py 3; python tools/bench_attn.py 3) 2>&1 | grep -v amdgpu.ids > $O/bench_attn_both.txt
cat $O/bench_attn_both.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "attention" -q -x 2>&1 | tail -3
