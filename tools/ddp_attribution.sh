R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_ddp; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/a -o a -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/a.log 2>&1
export CST_DDP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
rocprofv3 --kernel-trace --stats -d $O/b -o b -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/b.log 2>&1
cd $R
python tools/ab_kernels.py $(find $O/a -name '*.db') $(find $O/b -name '*.db') 13 > $O/ab_ddp_kernels.txt 2>&1
python tools/kernel_stats.py $(find $O/b -name '*.db') 13 200 | grep -i "nccl\|rccl\|multi_tensor\|copyBuffer\|launches per" > $O/ddp_extra_kernels.txt
grep -h '^{' $O/a.log | cut -c1-250 > $O/bench_a.txt; grep -h '^{' $O/b.log | cut -c1-250 > $O/bench_b.txt
rm -rf $O/a $O/b
unset CST_DDP_FORCE RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT
for i in 1 2 3; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' >> $O/ab_ms.txt
CST_DDP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29518 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/ddp /' >> $O/ab_ms.txt
done
CST_DDP_FORCE=1 python tools/ddp_overlap_trace.py --gpus 1 --steps 4 --warmup 2 --out $O/trace1 > $O/ddp_overlap_trace_1gpu.txt 2>&1
rm -rf $O/trace1/rank0
cat $O/ab_ms.txt; tail -5 $O/ddp_overlap_trace_1gpu.txt
