cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ddp1; mkdir -p $O
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 CST_DDP_FORCE=1
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > $O/trace.log 2>&1
cd $R
TDB=$(find $O/trace -name '*.db' | head -1)
python tools/kernel_stats.py $TDB 13 60 > $O/kernel_stats.txt 2>&1
python tools/idle_gaps.py $TDB 3 10 4 40 > $O/idle_gaps.txt 2>&1
rm -rf $O/trace

python tools/host_time_step.py prof > $O/host_time_ddp.txt 2>&1; unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT CST_DDP_FORCE; python tools/host_time_step.py prof > $O/host_time_plain.txt 2>&1
