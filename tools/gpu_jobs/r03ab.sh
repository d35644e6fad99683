#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_decode_trainer_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-extra > $O/bench_a.json 2>/dev/null
CST_NO_WT=1 python bench.py --no-cpu-baseline --no-extra > $O/bench_nowt.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extra > $O/bench_b.json 2>/dev/null
python - <<'PY'
import json
for n in ("bench_a", "bench_nowt", "bench_b"):
    d = json.loads([l for l in open("gpurun_out/r03ab/%s.json" % n) if l.startswith("{")][-1])
    pc = d["roofline"]["per_class_ms"]
    print("%-12s %.1f utt/s %.2f ms  gemm %.2f elementwise %.2f sum %.2f" % (n, d["value"], d["ms_per_step"], pc["gemm"], pc["elementwise"], sum(pc.values())))
PY
