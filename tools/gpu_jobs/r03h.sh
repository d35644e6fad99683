#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE -d $O/pmc -o pmc -- python3 $R/tools/bench_attn.py 1 > $O/pmc.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE -d $O/pmcp -o pmc -- $R/tools/probes/attn_issue_probe.bin 2 > $O/pmcp.log 2>&1
cd $R
python - > $O/l2.txt 2>&1 <<PY
import sys
sys.path.insert(0, "tools")
import pmc_step_summary as P
for d in ("$O/pmc", "$O/pmcp"):
    agg = P.load(d)
    print("#", d)
    for k, c in sorted(agg.items()):
        if "fa_" not in k and "probe_stg" not in k: continue
        n = len(c["TCC_HIT_sum"])
        hit, miss, rd, wr = (sum(c[x]) / n for x in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"))
        print("%-60s n=%3d  L2 hit %5.1f%%  requests %8.1f M  fabric read %8.1f MB (x2 corrected) write %7.1f MB" % (P.short(k)[:60], n, 100 * hit / max(hit + miss, 1), (hit + miss) / 1e6, 2 * rd * 64 / 1e6, wr * 64 / 1e6))
PY
cat $O/l2.txt
rm -rf $O/pmc $O/pmcp
