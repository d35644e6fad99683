#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ae; mkdir -p $O
python tools/torch_prof_step.py > $O/torch_prof.txt 2>&1
sed -n '/==== ATen ops/,$p' $O/torch_prof.txt | head -60
