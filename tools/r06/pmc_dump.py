#!/usr/bin/env python3
"""Per-kernel averages of every counter in a `rocprofv3 --pmc ... -d <dir>` result:  python3 tools/r06/pmc_dump.py <dir> [name filter]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pmc_step_summary import load, short
agg = load(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for name, cs in sorted(agg.items()):
    if flt and flt not in name:
        continue
    print(short(name))
    for c, v in sorted(cs.items()):
        print("    %-28s n=%-4d avg %.4g" % (c, len(v), sum(v) / len(v)))
