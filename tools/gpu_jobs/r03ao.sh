#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ao; mkdir -p $O
python tools/gemm_shapes_in_step.py --model chimera > $O/chimera_shapes.txt 2>/dev/null
head -64 $O/chimera_shapes.txt
python tools/gemm_shapes_in_step.py > $O/s2t_shapes.txt 2>/dev/null
