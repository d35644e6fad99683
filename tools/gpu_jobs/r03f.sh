#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
export CST_TRACE_LIB=$GRAFT_REPO_ROOT/tools/trace/libcst_trace.so
python - > $O/attn_trace.txt 2>&1 <<'PY'
import runpy, sys, os
for args in (["0.0"], ["0.1"]):
    print("== dropout", args[0]); sys.stdout.flush()
    sys.argv = ["tools/attn_trace.py"] + args
    try:
        runpy.run_path("tools/attn_trace.py", run_name="__main__")
    except SystemExit:
        pass
PY
cat $O/attn_trace.txt
