R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dec_ab; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/dec_trace -o trace -- python3 $R/bench.py --mode decode > $O/dec_trace.log 2>&1
cd $R
DDB=$(find $O/dec_trace -name '*.db' | head -1)
python tools/kernel_stats.py $DDB 1 16 > $O/decode_kernel_stats.txt 2>&1
rm -rf $O/dec_trace
cut -c1-150 $O/decode_kernel_stats.txt
