#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "attention" -q -x 2>&1 | tail -5 > $O/pytest_attn.txt
cat $O/pytest_attn.txt
python tools/bench_attn.py 3 2>&1 | grep -v amdgpu.ids > $O/bench_attn.txt
cat $O/bench_attn.txt
export CST_TRACE_LIB=$GRAFT_REPO_ROOT/tools/trace/libcst_trace.so
python - > $O/attn_trace.txt 2>&1 <<'PY'
import runpy, sys, os
for args in (["0.0"], ["0.1"]):
    print("== dropout", args[0]); sys.stdout.flush()
    sys.argv = ["tools/attn_trace.py"] + args
    runpy.run_path("tools/attn_trace.py", run_name="__main__")
PY
cat $O/attn_trace.txt
