R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > $O/trace.log 2>&1
cd $R
TDB=$(find $O/trace -name '*.db' | head -1)
python tools/kernel_stats.py $TDB 13 90 > $O/kernel_stats.txt 2>&1
rm -rf $O/trace
head -12 $O/kernel_stats.txt | cut -c1-150; grep -c . $O/kernel_stats.txt
python3 - <<'PY'
import re
tot=0;n=0
for line in open('gpurun_out/kstats/kernel_stats.txt'):
    m=re.match(r'\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)',line)
    if m and ('at::native' in m.group(4) or 'rocclr' in m.group(4)):
        tot+=float(m.group(2)); n+=float(m.group(1))
print("ATen/runtime kernels: %.3f ms per update over %.0f launches"%(tot,n))
PY
