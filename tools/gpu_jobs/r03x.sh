#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03x; mkdir -p $O
timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -x -k "config1_wav" --tb=short 2>&1 | grep -n "test_configs_gpu.py\|Error\|passed\|failed" | head
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_decode_engine_gpu.py -m gpu -q -k "transpose or repacks or attn or attention" 2>&1 | tail -5
python tools/host_time_step.py > $O/host.txt 2>&1; tail -3 $O/host.txt
python bench.py --no-cpu-baseline --no-extra > $O/bench.json 2>/dev/null
python bench.py --model chimera --no-cpu-baseline --no-extra > $O/bench_chimera.json 2>/dev/null
python - <<'PY'
import json
for n in ("bench", "bench_chimera"):
    d = json.loads([l for l in open("gpurun_out/r03x/%s.json" % n) if l.startswith("{")][-1])
    pc = d["roofline"]["per_class_ms"]
    print("%-14s %.1f utt/s %.2f ms  gemm %.2f sum %.2f" % (n, d["value"], d["ms_per_step"], pc["gemm"], sum(pc.values())))
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > $O/trace.log 2>&1
cd $R
TDB=$(find $O/trace -name '*.db' | head -1)
python tools/idle_gaps.py $TDB 3 10 4 45 > $O/idle_gaps.txt 2>&1
rm -rf $O/trace
head -25 $O/idle_gaps.txt
