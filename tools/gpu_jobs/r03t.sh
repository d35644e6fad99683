#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "linear or ffn or golden or transpose" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-extra > $O/bench_wt.json 2>/dev/null
CST_NO_WT=1 python bench.py --no-cpu-baseline --no-extra > $O/bench_nowt.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extra > $O/bench_wt2.json 2>/dev/null
python - <<'PY'
import json
for n in ("bench_wt", "bench_nowt", "bench_wt2"):
    d = json.loads([l for l in open("gpurun_out/r03t/%s.json" % n) if l.startswith("{")][-1])
    pc = d["roofline"]["per_class_ms"]
    print("%-12s %.1f utt/s %.2f ms  gemm %.2f elementwise %.2f sum %.2f" % (n, d["value"], d["ms_per_step"], pc["gemm"], pc["elementwise"], sum(pc.values())))
PY
