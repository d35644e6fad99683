#!/usr/bin/env python3
"""HBM-side traffic of the GEMM class per launch from a `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE` pass over
`bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline` (2 updates): the JSON bench.py quotes as `roofline.traffic`.
    python3 tools/pmc_gemm_class.py <dir-or-db> <updates in the pass> [<build id: git short hash of the tree the pass ran on>] > profiles/<round>_pmc_gemm_class.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_step_summary import load  # noqa: E402

agg = load(sys.argv[1])
updates = int(sys.argv[2]) if len(sys.argv) > 2 else 2
build = sys.argv[3] if len(sys.argv) > 3 else "unrecorded"
kern, tot_b, tot_n = [], 0.0, 0
for name, c in agg.items():
    if "gemm" not in name or "splitk_reduce" in name:
        continue
    rd, wr = c.get("TCC_EA0_RDREQ_sum", []), c.get("TCC_EA0_WRREQ_sum", [])
    n = min(len(rd), len(wr))
    if n == 0:
        continue
    b = sum(2.0 * rd[i] * 64 + wr[i] * 64 for i in range(n))
    kern.append({"kernel": name.replace("void (anonymous namespace)::", "")[:64], "launches_%d_updates" % updates: n, "fabric_MB_per_launch": round(b / n / 1e6, 1)})
    tot_b += b
    tot_n += n
kern.sort(key=lambda k: -k["fabric_MB_per_launch"] * k["launches_%d_updates" % updates])
print(json.dumps({
    "source": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE over %d updates of `bench.py --steps 1 --warmup 1`" % updates,
    "build": build,
    "kernels": kern,
    "gemm_class_launches_per_update": round(tot_n / updates),
    "traffic_bytes_per_launch": tot_b / max(tot_n, 1),
    "note": "HBM-side (fabric) bytes per GEMM-class launch, launch-weighted over the GEMM kernels of one update = (2 x TCC_EA0_RDREQ + TCC_EA0_WRREQ) x 64 B (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are included in the read count; split-K reduce launches excluded.",
}, indent=1))
