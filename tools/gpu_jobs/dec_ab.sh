R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dec_ab; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for ck in flash shared flash shared; do
  export CST_DEC_CROSS_KERNEL=$ck
  python3 $R/bench.py --mode decode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$ck', round(d['value'],1), d['config'].get('ms_per_decode_step'), d.get('ms_per_step'))"
done
export CST_DEC_CROSS_KERNEL=shared
rocprofv3 --kernel-trace --stats -d $O/dec_trace -o trace -- python3 $R/bench.py --mode decode > $O/dec_trace.log 2>&1
cd $R
DDB=$(find $O/dec_trace -name '*.db' | head -1)
python tools/kernel_stats.py $DDB 1 14 > $O/decode_kernel_stats_shared.txt 2>&1
rm -rf $O/dec_trace
cut -c1-150 $O/decode_kernel_stats_shared.txt
