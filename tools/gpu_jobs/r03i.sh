#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
./tools/probes/attn_issue_probe.bin 4 > $O/probe_rounds.txt 2>&1
cat $O/probe_rounds.txt
