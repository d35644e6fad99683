#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
./tools/probes/attn_issue_probe.bin 2 > $O/attn_issue_probe_stg.txt 2>&1
cat $O/attn_issue_probe_stg.txt
