#!/usr/bin/env python3
"""Fabric-side traffic of ONE GEMM launch shape from two `rocprofv3 --pmc` passes over tools/gemm_one.py (the JSON bench.py quotes as
`roofline.dominant_launch.traffic`):
    python3 tools/pmc_gemm_one.py <dir/db: TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum pass> <dir/db: TCC_HIT_sum TCC_MISS_sum pass> M N K <build> > profiles/<round>_pmc_gemm.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_step_summary import load  # noqa: E402

ea, hm = load(sys.argv[1]), load(sys.argv[2])
M, N, K = (int(v) for v in sys.argv[3:6])
build = sys.argv[6] if len(sys.argv) > 6 else "unrecorded"


def mean(agg, counter):
    vals = [v for name, cs in agg.items() if "gemm8p" in name for v in cs.get(counter, [])]
    return sum(vals) / max(len(vals), 1), len(vals)


rd, n = mean(ea, "TCC_EA0_RDREQ_sum")
wr, _ = mean(ea, "TCC_EA0_WRREQ_sum")
hit, _ = mean(hm, "TCC_HIT_sum")
miss, _ = mean(hm, "TCC_MISS_sum")
alg = (M * K + N * K) * 2.0 + 2.0 * M * N * 2.0
print(json.dumps({
    "kernel": "gemm8p_kernel<true, true, 1>", "shape": [M, N, K], "epilogue": "bias+gelu+aux_out (wav2vec2 fc1 forward)", "build": build,
    "command": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -- python3 tools/gemm_one.py kk %d %d %d 3 fc1  (separate pass: --pmc TCC_HIT_sum TCC_MISS_sum)" % (M, N, K),
    "launches_averaged": n, "TCC_EA0_RDREQ_sum": rd, "TCC_EA0_WRREQ_sum": wr, "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
    "l2_hit_rate": hit / max(hit + miss, 1.0), "read_bytes_gfx950_corrected": 2.0 * rd * 64.0, "write_bytes": wr * 64.0,
    "traffic_bytes_per_launch": 2.0 * rd * 64.0 + wr * 64.0, "algorithmic_bytes_per_launch": alg,
    "note": "HBM-side (fabric) bytes per launch = 2 x TCC_EA0_RDREQ x 64 B (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section) + TCC_EA0_WRREQ x 64 B; Infinity-Cache hits are included in the read count.",
}, indent=1))
