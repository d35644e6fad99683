#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aa; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "pos_conv or golden or wav2vec or w2v" 2>&1 | tail -6
python bench.py --no-cpu-baseline --no-extra > $O/bench.json 2>/dev/null
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03aa/bench.json") if l.startswith("{")][-1])
pc = d["roofline"]["per_class_ms"]
print("train %.1f utt/s %.2f ms " % (d["value"], d["ms_per_step"]), pc, "sum %.2f" % sum(pc.values()))
PY
