#!/bin/bash
# Collects the per-round measurement artefacts on the GPU box (run through gpurun):  bash tools/collect_profiles.sh <round tag>
# rocprofv3 always gets the program itself behind `--`; counters are collected in their own passes without trace domains.
# The result databases are summarised ON THE BOX and deleted (gpurun merges at most 64 MiB back).
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_prof; mkdir -p $O
BUILD=${2:-unrecorded}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > $O/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE -d $O/pmc_hbm -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $O/pmc_hbm.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/dec_trace -o trace -- python3 $R/bench.py --mode decode > $O/dec_trace.log 2>&1
# phase ranges (round 6): the reference's record_function scopes as roctx markers, CSV so that the per-range statistics can be read as text
export CST_ROCTX=1
rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/roctx -o roctx -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extra > $O/roctx.log 2>&1
unset CST_ROCTX
(for f in $(find $O/roctx -name '*marker*stats*.csv' -o -name '*marker_api_stats*.csv' | head -2); do echo "== $f"; cat $f; done; for f in $(find $O/roctx -name '*marker*trace*.csv' | head -1); do echo "== $f (first 40 rows)"; head -40 $f; done) > $O/${TAG}_roctx_phases.txt 2>&1
rm -rf $O/roctx
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/pmc_one_ea -o pmc -- python3 $R/tools/gemm_one.py kk 47968 3072 768 3 fc1 > $O/pmc_one_ea.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_one_hm -o pmc -- python3 $R/tools/gemm_one.py kk 47968 3072 768 3 fc1 > $O/pmc_one_hm.log 2>&1
cd $R
TDB=$(find $O/trace -name '*.db' | head -1); DDB=$(find $O/dec_trace -name '*.db' | head -1)
python tools/pmc_gemm_one.py $O/pmc_one_ea $O/pmc_one_hm 47968 3072 768 $BUILD > $O/${TAG}_pmc_gemm.json 2> $O/pmc_gemm_one.err
python tools/kernel_stats.py $TDB 13 90 > $O/${TAG}_kernel_stats.txt 2>&1
python tools/kernel_stats.py $DDB 1 30 > $O/${TAG}_decode_kernel_stats.txt 2>&1
python tools/idle_gaps.py $TDB 3 10 4 40 > $O/${TAG}_idle_gaps.txt 2>&1
python tools/pmc_step_summary.py --trace-db $TDB $O/pmc_mfma $O/pmc_hbm > $O/${TAG}_pmc_step_summary.txt 2>&1
python tools/pmc_gemm_class.py $O/pmc_hbm 2 $BUILD > $O/${TAG}_pmc_gemm_class.json 2> $O/pmc_gemm_class.err
grep -h '^{' $O/trace.log > $O/${TAG}_bench_under_trace.json
# the same launch with the A-stationary walk (8 m-tile rows per XCD group, all N columns swept: CST_GEMM8P_GROUP_M=8)
export CST_GEMM8P_GROUP_M=8
cd /tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/pmc_g8_ea -o pmc -- python3 $R/tools/gemm_one.py kk 47968 3072 768 3 fc1 > $O/pmc_g8_ea.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_g8_hm -o pmc -- python3 $R/tools/gemm_one.py kk 47968 3072 768 3 fc1 > $O/pmc_g8_hm.log 2>&1
unset CST_GEMM8P_GROUP_M
cd $R
python tools/pmc_gemm_one.py $O/pmc_g8_ea $O/pmc_g8_hm 47968 3072 768 "$BUILD group_m=8" > $O/${TAG}_pmc_gemm_groupm8.json 2> $O/pmc_gemm_g8.err
rm -rf $O/trace $O/pmc_mfma $O/pmc_hbm $O/dec_trace $O/pmc_one_ea $O/pmc_one_hm $O/pmc_g8_ea $O/pmc_g8_hm
# the bench lines below quote the counter files of THIS build
cp $O/${TAG}_pmc_gemm_class.json $O/${TAG}_pmc_gemm.json $R/profiles/
python bench.py > $O/${TAG}_bench_s2t.json 2> $O/bench_s2t.err
python bench.py --model chimera --no-cpu-baseline --no-extra > $O/${TAG}_bench_chimera.json 2> $O/bench_chimera.err
python bench.py --dropout 0 --no-cpu-baseline --no-extra > $O/${TAG}_bench_dropout0.json 2> $O/bench_dropout0.err
python bench.py --mode decode > $O/${TAG}_bench_decode.json 2> $O/bench_decode.err
python bench.py --lengths max --no-cpu-baseline --no-extra > $O/${TAG}_bench_maxlen.json 2> $O/bench_maxlen.err
python tools/probes/hipblaslt_names.py 2>/dev/null | grep ratio > $O/${TAG}_hipblaslt_vs_this_library.txt
python tools/gemm_shapes_in_step.py > $O/${TAG}_gemm_shapes_in_step.txt 2>/dev/null
python tools/vendor_in_step.py > $O/${TAG}_vendor_in_step.txt 2>/dev/null
tools/probes/mfma_rate.bin 4000 > $O/${TAG}_mfma_shape_probe_raw.txt 2>&1
python tools/bench_conv_layout.py > $O/${TAG}_conv_layout.txt 2>/dev/null
python tools/r06/epi_decompose.py 31760 > $O/${TAG}_epilogue_decomposition.txt 2>/dev/null
(python tools/r06/bench_conv0_bwd.py; CST_CONV0_NO_MFMA=1 python tools/r06/bench_conv0_bwd.py) > $O/${TAG}_conv0_bwd_mfma_vs_register.txt 2>/dev/null
ls -la $O; du -sh $O
