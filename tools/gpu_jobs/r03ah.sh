#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/bench_dec_splitk.py 2>&1 | tail -6
