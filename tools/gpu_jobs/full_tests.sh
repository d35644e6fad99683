#!/bin/bash
# full GPU test suite + default bench line.  usage: bash tools/gpu_jobs/full_tests.sh <tag>
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG}_full; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json; tail -3 $O/bench.err
