#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/find_h2d.py 2>&1 | tail -30
