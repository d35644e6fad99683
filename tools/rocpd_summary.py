#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (`rocprofv3 --kernel-trace --stats`) as the per-kernel stats table
(name, calls, total ms, average us, %) that gets committed under profiles/."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n if len(n) <= 110 else n[:107] + "..."


def main(path, out=None):
    import glob, os
    if os.path.isdir(path):  # rocprofv3 -d <dir>: take the (first) *.db below it
        found = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
        if not found:
            sys.exit("no .db under " + path)
        path = found[0]
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary of %s" % path, "# total kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)),
             "%-112s %8s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct")]
    for name, calls, tot, avg in rows:
        lines.append("%-112s %8d %12.3f %12.2f %6.2f%%" % (short(name), calls, tot / 1e6, avg / 1e3, 100.0 * tot / total))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
