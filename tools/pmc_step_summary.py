#!/usr/bin/env python3
"""Per-kernel PMC summary of one training update collected by
    rocprofv3 --pmc <counters> -d <dir> -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
(counters in their own passes, no trace domains: MI355X_MICROARCH.md §rocprofv3 PMC slots).  Usage:
    python3 tools/pmc_step_summary.py [--stats <rocpd_summary kernel stats .txt>] <dir-or-db> [<dir-or-db> ...] > profiles/<name>.txt
MFMA pass  (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE): MFMA utilisation = MFMA-busy cycles / (kernel cycles x 1024 SIMDs)
           — the fraction of SIMD-cycles in which the matrix pipe was executing.  rocprofv3 reports GRBM_GUI_ACTIVE summed over
           the 8 XCDs, so kernel cycles = GUI_ACTIVE / 8 (calibration: the fc1 GEMM is 226 GFLOP = 6.9 M v_mfma_32x32x16_bf16 x 32
           busy cycles = 216 k cycles per SIMD of a 0.275 ms launch = 39 % at the ~2.0 GHz the chip sustains under this load;
           the counter ratio gives 34-35 %);
HBM pass   (TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE): fabric-side bytes = 2 x RDREQ x 64 B (gfx950 wide-read correction,
           MI355X_MICROARCH.md §HBM) + WRREQ x 64 B; GB/s over GUI-active cycles at the effective clock reported alongside."""
import collections
import glob
import os
import re
import sqlite3
import sys


def load(path):
    if os.path.isdir(path):
        found = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
        if not found:
            sys.exit("no .db under " + path)
        path = found[0]
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tables else None
    if view is None:
        sys.exit("no counters_collection view in %s (tables: %s)" % (path, tables))
    cols = [c[1] for c in db.execute("pragma table_info('%s')" % view)]
    ix = {c: i for i, c in enumerate(cols)}
    kcol = "kernel_name" if "kernel_name" in ix else "name"
    gcol = next((c for c in ("grid_size", "grid_size_x", "grid_x") if c in ix), None)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in db.execute("select * from %s" % view):
        name = r[ix[kcol]]
        # the attention kernels serve three shape families in one update (wav2vec2 layers: ~1500 frames x 12 heads, packed; the
        # 512-wide encoder: ~375 frames; decoder cross-attention: <= 128 queries): one line per launch grid, so that the wav2vec2
        # launches — the "encoder attention" of the north-star target — have their own matrix-pipe figure
        if gcol is not None and ("fa_fwd" in name or "fa_dq" in name or "fa_dkv" in name):
            name = "%s [grid %d]" % (name, int(r[ix[gcol]]))
        agg[name][r[ix["counter_name"]]].append(r[ix["value"]])
    return agg


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n if len(n) <= 96 else n[:93] + "..."


XCDS = 8


def load_stats(path):
    """kernel name prefix -> average duration in us, from tools/rocpd_summary.py output (a separate --kernel-trace run)."""
    out = {}
    for line in open(path):
        m = re.match(r"^(.{112}) +(\d+) +([\d.]+) +([\d.]+) +[\d.]+%", line)
        if m:
            out[m.group(1).strip()[:60]] = float(m.group(4))
    return out


def main():
    argv, stats = sys.argv[1:], {}
    if argv and argv[0] == "--stats":
        stats, argv = load_stats(argv[1]), argv[2:]
    elif argv and argv[0] == "--trace-db":  # a `rocprofv3 --kernel-trace --stats` results.db of the same command: average durations
        c = sqlite3.connect(argv[1])
        stats = {short(r[0])[:60]: r[2] / max(r[1], 1) for r in c.execute("select name,total_calls,total_duration from top_kernels")}
        argv = argv[2:]
    for path in argv:
        agg = load(path)
        print("# %s" % path)
        rows = []
        for name, cs in agg.items():
            n = max(len(v) for v in cs.values())
            tot = {c: sum(v) for c, v in cs.items()}
            rows.append((tot.get("GRBM_GUI_ACTIVE", 0.0), name, n, tot))
        rows.sort(reverse=True)
        gui_all = sum(r[0] for r in rows) or 1.0
        for gui, name, n, tot in rows[:40]:
            line = "%-96s n=%5d gui_active %6.2f%%" % (short(name), n, 100.0 * gui / gui_all)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and gui > 0:
                line += "  MFMA busy %5.1f%% of SIMD-cycles" % (100.0 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / XCDS * 1024.0))
            if "SQ_ACTIVE_INST_VALU" in tot and gui > 0:  # quad-cycles (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count 4-cycle units)
                line += "  VALU active %5.1f%%" % (100.0 * 4.0 * tot["SQ_ACTIVE_INST_VALU"] / (gui / XCDS * 1024.0))
            if "TCC_EA0_RDREQ_sum" in tot and gui > 0:
                by = 2.0 * tot["TCC_EA0_RDREQ_sum"] * 64.0 + tot.get("TCC_EA0_WRREQ_sum", 0.0) * 64.0
                line += "  fabric %8.1f MB/launch  %7.1f B/cycle" % (by / n / 1e6, by / (gui / XCDS))
                us = stats.get(short(name)[:60])
                if us:
                    line += "  %6.0f GB/s at %.1f us/launch (kernel-trace run)" % (by / n / us / 1e3, us)
            print(line)
        print()


if __name__ == "__main__":
    main()
