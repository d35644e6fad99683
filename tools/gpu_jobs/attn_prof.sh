#!/bin/bash
# attention kernels alone: kernel-trace timings + matrix-pipe / VALU / wait counters.  usage: bash tools/gpu_jobs/attn_prof.sh <tag>
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_attn; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/tools/bench_attn.py 2 > $O/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc1 -o pmc -- python3 $R/tools/bench_attn.py 1 > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/pmc2 -o pmc -- python3 $R/tools/bench_attn.py 1 > $O/pmc2.log 2>&1
cd $R
python tools/pmc_step_summary.py $O/pmc1 > $O/${TAG}_attn_pmc.txt 2>&1
python - <<PY >> $O/${TAG}_attn_pmc.txt
import sys, glob, sqlite3, collections
sys.path.insert(0, "tools")
import pmc_step_summary as P
agg = P.load("$O/pmc2")
print("\n# second pass: wave-cycle breakdown (quad-cycles summed over waves; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)")
for k, c in sorted(agg.items()):
    if "attn" not in k and "fa_" not in k: continue
    wc = sum(c["SQ_WAVE_CYCLES"]) or 1
    f = lambda n: 100.0 * sum(c[n]) / wc
    lds = sum(c["SQ_LDS_IDX_ACTIVE"]) or 1
    print("%-70s n=%3d wait_any %5.1f%%  wait_inst %5.1f%%  active %5.1f%%  wait_inst_lds %5.1f%%  lds bank conflict %5.1f%% of lds cycles" % (P.short(k)[:70], len(c["SQ_WAVE_CYCLES"]), f("SQ_WAIT_ANY"), f("SQ_WAIT_INST_ANY"), f("SQ_ACTIVE_INST_ANY"), f("SQ_WAIT_INST_LDS"), 100.0 * sum(c["SQ_LDS_BANK_CONFLICT"]) / lds))
PY
grep -h "attn\|fa_" $(find $O/trace -name "*kernel_stats.csv") | cut -d, -f1-4 | sed 's/_ZN12_GLOBAL__N_1//' > $O/${TAG}_attn_kernel_stats.csv
python tools/bench_attn.py 3 > $O/${TAG}_bench_attn.txt 2>&1
rm -rf $O/trace $O/pmc1 $O/pmc2
cat $O/${TAG}_attn_pmc.txt $O/${TAG}_attn_kernel_stats.csv $O/${TAG}_bench_attn.txt
