#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
timeout 1200 python -m pytest tests/test_decode_engine_gpu.py tests/test_decode_trainer_gpu.py tests/test_bench_gpu.py -m gpu -q -x 2>&1 | tail -8
for l in 1 2 3 4; do
  CST_DEC_LANES=$l timeout 600 python bench.py --mode decode > $O/dec_l$l.json 2>$O/dec_l$l.err
done
python - <<'PY'
import json
for l in (1,2,3,4):
    try:
        d = json.loads([x for x in open("gpurun_out/r03z/dec_l%d.json" % l) if x.startswith("{")][-1])
        print("lanes %d: %.1f utt/s  %.3f ms/decode step  frac %.3f  enc %.1f ms" % (l, d["value"], d["config"]["ms_per_decode_step"], d["roofline"]["frac"], d["config"]["encoder_ms"]))
    except Exception as e:
        print("lanes", l, "failed", e)
PY
python tools/bench_gemm_splitk.py 2>&1 | cut -c1-60 | head -12
python bench.py --no-cpu-baseline --no-extra > $O/bench.json 2>/dev/null
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03z/bench.json") if l.startswith("{")][-1])
pc = d["roofline"]["per_class_ms"]
print("train %.1f utt/s %.2f ms  gemm %.2f sum %.2f" % (d["value"], d["ms_per_step"], pc["gemm"], sum(pc.values())))
PY
