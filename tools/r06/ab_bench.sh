#!/bin/sh
# same-box A/B of two library builds inside the training update:  sh tools/r06/ab_bench.sh <other .so> [runs]
# prints value / ms per update / per-class ms of the default build and of the other one, alternating
O=$1; N=${2:-2}
for i in $(seq $N); do
  for which in new old; do
    if [ $which = new ]; then python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > /tmp/ab.json 2>/dev/null
    else CST_AB_LIB=$O python tools/ab_lib.py bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > /tmp/ab.json 2>/dev/null; fi
    python - $which <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 1), round(d["ms_per_step"], 2), d["config"]["loss"], d["roofline"]["per_class_ms"])
PY
  done
done
