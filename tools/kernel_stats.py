#!/usr/bin/env python3
"""Per-kernel totals of one `rocprofv3 --kernel-trace --stats` result database (rocpd sqlite) per update.

    python tools/kernel_stats.py <results.db> <updates in the run> [rows]

Prints calls/update, us/update and the average duration of every kernel, the launch count per update and the summed kernel time."""
import sqlite3
import sys


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    rows_max = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    c = sqlite3.connect(path)
    rows = list(c.execute("select name,total_calls,total_duration from top_kernels"))
    rows.sort(key=lambda r: -r[2])
    calls = sum(r[1] for r in rows)
    total = sum(r[2] for r in rows)
    # (top_kernels.total_duration is in microseconds)
    print("%d kernels launched over %g updates = %.0f launches per update; kernel time %.2f ms per update" % (calls, steps, calls / steps, total / steps / 1e3))
    print("%10s %12s %10s  %s" % ("calls/upd", "ms/upd", "avg us", "kernel"))
    for name, n, dur in rows[:rows_max]:
        print("%10.1f %12.3f %10.2f  %s" % (n / steps, dur / steps / 1e3, dur / n, name[:150]))


if __name__ == "__main__":
    main()
