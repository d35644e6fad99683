R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/memcpy; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --output-format csv -d $O/t -o mc -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extra > $O/log.txt 2>&1
cd $R
F=$(find $O/t -name '*hip_api_trace.csv' | head -1)
echo $F; head -2 $F | cut -c1-400
python3 - "$F" <<'PY'
import csv, sys, collections
c = collections.Counter()
n = 0
for r in csv.DictReader(open(sys.argv[1])):
    f = r.get("Function", "")
    if "Memcpy" in f or "Memset" in f:
        c[f] += 1
    n += 1
print(n, "api calls")
for k, v in c.most_common(20):
    print(v, k)
PY
rm -rf $O/t
