#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ai; mkdir -p $O
timeout 1500 python -m pytest tests/test_decode_engine_gpu.py tests/test_decode_trainer_gpu.py tests/test_configs_gpu.py -m gpu -q -x -k "engine or decode or config5" 2>&1 | tail -3
python bench.py --mode decode > $O/dec.json 2>/dev/null
CST_DEC_NO_SPLITK=1 python bench.py --mode decode > $O/dec_nosplit.json 2>/dev/null
python - <<'PY'
import json
for n in ("dec", "dec_nosplit"):
    d = json.loads([x for x in open("gpurun_out/r03ai/%s.json" % n) if x.startswith("{")][-1])
    print("%-12s %.1f utt/s  %.3f ms/decode step  frac %.3f  %s" % (n, d["value"], d["config"]["ms_per_decode_step"], d["roofline"]["frac"], d["roofline"]["kernel"][:40]))
PY
