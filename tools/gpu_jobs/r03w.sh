#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03w; mkdir -p $O
timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -x -k "config1_wav" --tb=short 2>&1 | grep -n "test_configs_gpu.py\|Error\|passed\|failed" | head
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_decode_engine_gpu.py -m gpu -q -k "transpose or repacks" 2>&1 | tail -5
python tools/host_time_step.py prof > $O/host_prof.txt 2>&1
head -80 $O/host_prof.txt
