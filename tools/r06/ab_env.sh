#!/bin/sh
# usage: sh ab_env.sh VAR=VAL [runs]
V=$1; N=${2:-2}
for i in $(seq $N); do
  for which in base alt; do
    if [ $which = base ]; then python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > /tmp/ab.json 2>/tmp/ab.err
    else env $V python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > /tmp/ab.json 2>/tmp/ab.err; fi
    python - $which <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["value"], 1), round(d["ms_per_step"], 2), d["config"]["loss"], d["roofline"]["per_class_ms"])
except Exception as e:
    print(sys.argv[1], "failed", e, open("/tmp/ab.err").read()[-800:])
PY
  done
done
