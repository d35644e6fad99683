#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03an; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py tests/test_decode_trainer_gpu.py -m gpu -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-extra > $O/new1.json 2>/dev/null
CST_WT_MIN_ROWS=4096 python bench.py --no-cpu-baseline --no-extra > $O/old.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extra > $O/new2.json 2>/dev/null
python bench.py --model chimera --no-cpu-baseline --no-extra > $O/chim_new.json 2>/dev/null
CST_WT_MIN_ROWS=4096 python bench.py --model chimera --no-cpu-baseline --no-extra > $O/chim_old.json 2>/dev/null
python - <<'PY'
import json
for n in ("new1", "old", "new2", "chim_new", "chim_old"):
    d = json.loads([l for l in open("gpurun_out/r03an/%s.json" % n) if l.startswith("{")][-1])
    pc = d["roofline"]["per_class_ms"]
    print("%-10s %.1f utt/s %.2f ms  gemm %.2f ew %.2f sum %.2f" % (n, d["value"], d["ms_per_step"], pc["gemm"], pc["elementwise"], sum(pc.values())))
PY
