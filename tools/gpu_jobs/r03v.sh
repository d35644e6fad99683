#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03v; mkdir -p $O
timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -x -k "config1_wav" 2>&1 | tail -40
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_decode_engine_gpu.py -m gpu -q -k "transpose or repacks" 2>&1 | tail -15
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > $O/trace.log 2>&1
cd $R
TDB=$(find $O/trace -name '*.db' | head -1)
python tools/idle_gaps.py $TDB 3 10 4 45 > $O/idle_gaps.txt 2>&1
rm -rf $O/trace
head -45 $O/idle_gaps.txt
