#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03u; mkdir -p $O
python tools/probes/hipblaslt_names.py > $O/hipblaslt_vs_ours.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > $O/trace.log 2>&1
cd $R
TDB=$(find $O/trace -name '*.db' | head -1)
python tools/idle_gaps.py $TDB 3 10 4 45 > $O/idle_gaps.txt 2>&1
rm -rf $O/trace
cat $O/hipblaslt_vs_ours.txt; head -30 $O/idle_gaps.txt
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
