#!/usr/bin/env python3
"""Per-kernel totals of two rocprofv3 --kernel-trace --stats result databases side by side (A/B of two library builds)."""
import sqlite3, sys
def top(path):
    c = sqlite3.connect(path)
    return {r[0]: (r[1], r[2]) for r in c.execute("select name,total_calls,total_duration from top_kernels")}
a, b, steps = top(sys.argv[1]), top(sys.argv[2]), float(sys.argv[3])
rows = [(a[k][1] / steps, b.get(k, (0, 0))[1] / steps, a[k][0] / steps, k[:110]) for k in a]
for r in sorted(rows, key=lambda r: -r[0])[:24]:
    print("A %8.0f us/step  B %8.0f us/step  calls/step %6.1f  %s" % r)
print("all kernels A %.0f B %.0f us/step" % (sum(v[1] for v in a.values()) / steps, sum(v[1] for v in b.values()) / steps))
