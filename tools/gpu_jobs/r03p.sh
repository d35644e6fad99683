#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
python bench.py --no-cpu-baseline --no-extra > $O/bench_dma.json 2>/dev/null
CST_ATTN_GENERIC=1 python bench.py --no-cpu-baseline --no-extra > $O/bench_generic.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extra > $O/bench_dma2.json 2>/dev/null
python - <<'PY'
import json
for n in ("bench_dma", "bench_generic", "bench_dma2"):
    d = json.loads([l for l in open("gpurun_out/r03p/%s.json" % n) if l.startswith("{")][-1])
    pc = d["roofline"]["per_class_ms"]
    print("%-14s %.1f utt/s %.2f ms  classes sum %.2f  %s" % (n, d["value"], d["ms_per_step"], sum(pc.values()), pc))
PY
