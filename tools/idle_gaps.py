#!/usr/bin/env python3
"""Where the GPU sits idle inside an update: gaps between consecutive kernel dispatches of one `rocprofv3 --kernel-trace` result
database (rocpd sqlite), summed per (kernel before the gap -> kernel after the gap).

    python tools/idle_gaps.py <results.db> <warmup updates> <timed updates> [min gap us] [rows]

The window is cut with the optimizer kernel as the marker (one adam_kernel launch per update): from the end of the last warm-up
update's to the end of the last timed update's — whatever the program runs before or after (bench.py's other legs) is left out."""
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n[:70]


def main():
    path, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    min_gap = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
    nrows = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    s_col = "start" if "start" in cols else "start_timestamp"
    e_col = "end" if "end" in cols else "end_timestamp"
    rows = list(c.execute("select name, %s, %s from kernels order by %s" % (s_col, e_col, s_col)))
    marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    assert len(marks) >= warm + steps, "found %d optimizer launches, need %d" % (len(marks), warm + steps)
    rows = rows[marks[warm - 1] + 1:marks[warm + steps - 1] + 1]
    span = (rows[-1][2] - rows[0][1]) / 1e3
    busy = 0.0
    gaps = defaultdict(lambda: [0, 0.0])
    hist = defaultdict(lambda: [0, 0.0])
    end = rows[0][2]
    prev = rows[0][0]
    busy = (rows[0][2] - rows[0][1]) / 1e3
    for name, s, e in rows[1:]:
        g = (s - end) / 1e3
        if g >= min_gap:
            k = (short(prev), short(name))
            gaps[k][0] += 1
            gaps[k][1] += g
        if g > 0:
            b = "<2us" if g < 2 else "2-5us" if g < 5 else "5-20us" if g < 20 else "20-100us" if g < 100 else "100us-1ms" if g < 1000 else ">1ms"
            hist[b][0] += 1
            hist[b][1] += g
        if e > end:
            busy += (e - max(s, end)) / 1e3
            end = e
            prev = name
    upd = float(steps)
    print("window %.1f ms (%d updates): GPU busy %.2f ms/update, idle %.2f ms/update" % (span / 1e3, upd, busy / upd / 1e3, (span - busy) / upd / 1e3))
    print("gap histogram (count/update, ms/update):")
    for b in ("<2us", "2-5us", "5-20us", "20-100us", "100us-1ms", ">1ms"):
        if b in hist:
            print("  %-10s %8.1f %8.3f" % (b, hist[b][0] / upd, hist[b][1] / upd / 1e3))
    print("gaps >= %.0f us, by neighbours (count/update, ms/update, avg us):" % min_gap)
    for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:nrows]:
        print("  %6.1f %8.3f %8.1f  %s  ->  %s" % (n / upd, t / upd / 1e3, t / n, a, b))


if __name__ == "__main__":
    main()
