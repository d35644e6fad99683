#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
(echo "== B=32 T=1499"; python tools/bench_attn.py 3; echo "== B=8 T=5996"; ATT_B=8 ATT_T=5996 python tools/bench_attn.py 3; echo "== B=32 T=1536"; ATT_T=1536 python tools/bench_attn.py 3; echo "== zero data"; ATT_SCALE=0 python tools/bench_attn.py 3; echo "== small data (x0.1)"; ATT_SCALE=0.1 python tools/bench_attn.py 3; echo "== generic kernels"; CST_ATTN_GENERIC=1 python tools/bench_attn.py 3) 2>&1 | grep -v amdgpu.ids > $O/bench_variants.txt
cat $O/bench_variants.txt
