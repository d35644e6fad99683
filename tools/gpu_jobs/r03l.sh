#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
python tools/bench_attn_packed.py 3 2>&1 | grep -v amdgpu.ids > $O/bench_attn_packed.txt
cat $O/bench_attn_packed.txt
