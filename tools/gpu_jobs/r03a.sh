#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
./tools/probes/attn_issue_probe.bin > $O/attn_issue_probe.txt 2>&1
python tools/bench_attn.py 3 > $O/bench_attn_base.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/blaslt -o t -- python3 $GRAFT_REPO_ROOT/tools/probes/hipblaslt_names.py > $GRAFT_REPO_ROOT/$O/hipblaslt_names.txt 2>&1
cd $GRAFT_REPO_ROOT
find $O/blaslt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/hipblaslt_kernel_stats.csv
rm -rf $O/blaslt
cat $O/attn_issue_probe.txt
