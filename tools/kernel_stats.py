#!/usr/bin/env python3
"""Per-kernel totals of one `rocprofv3 --kernel-trace --stats` result database (rocpd sqlite) per update.

    python tools/kernel_stats.py <results.db> <updates in the run> [rows]

Prints calls/update, us/update and the average duration of every kernel, the launch count per update and the summed kernel time.
A training run is cut with the optimizer kernel as the marker (one adam_kernel launch per update, as tools/idle_gaps.py does): the
window runs from the end of the FIRST update to the end of the last one, so the launches of the model build (≈ 1 000 parameter
uploads and initialisers) and of the first update's one-off work are not spread over the updates (until round 5 they were: the
ATen / copy lines of r01 .. r04 are inflated by them, the libcst_hip lines are not).  Without the marker (decode runs) the totals of
the whole process are divided by <updates in the run> as before."""
import sqlite3
import sys
from collections import defaultdict


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    rows_max = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    c = sqlite3.connect(path)
    rows = None
    try:
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        s_col = "start" if "start" in cols else "start_timestamp"
        e_col = "end" if "end" in cols else "end_timestamp"
        disp = list(c.execute("select name, %s, %s from kernels order by %s" % (s_col, e_col, s_col)))
        marks = [i for i, r in enumerate(disp) if "adam_kernel" in r[0]]
        if len(marks) >= 3:
            win = disp[marks[0] + 1:marks[-1] + 1]
            steps = float(len(marks) - 1)
            agg = defaultdict(lambda: [0, 0.0])
            for name, s, e in win:
                agg[name][0] += 1
                agg[name][1] += (e - s) / 1e3  # ns -> us
            rows = [(k, v[0], v[1]) for k, v in agg.items()]
            print("(window: the %d updates behind the first optimizer launch)" % int(steps))
    except sqlite3.Error:
        rows = None
    if rows is None:
        rows = list(c.execute("select name,total_calls,total_duration from top_kernels"))  # (total_duration is in microseconds)
    rows.sort(key=lambda r: -r[2])
    calls = sum(r[1] for r in rows)
    total = sum(r[2] for r in rows)
    print("%d kernels launched over %g updates = %.0f launches per update; kernel time %.2f ms per update" % (calls, steps, calls / steps, total / steps / 1e3))
    print("%10s %12s %10s  %s" % ("calls/upd", "ms/upd", "avg us", "kernel"))
    for name, n, dur in rows[:rows_max]:
        print("%10.1f %12.3f %10.2f  %s" % (n / steps, dur / steps / 1e3, dur / n, name[:150]))


if __name__ == "__main__":
    main()
