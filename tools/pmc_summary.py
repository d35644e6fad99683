#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc results (rocpd sqlite): per kernel name, mean of each counter per dispatch."""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [c[1] for c in db.execute("pragma table_info('counters_collection')")]
rows = db.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r[ix["kernel_name"]] if "kernel_name" in ix else r[ix["name"]]
    agg[name][r[ix["counter_name"]]].append(r[ix["value"]])
for name, cs in agg.items():
    short = re.sub(r"\(anonymous namespace\)::", "", name)[:100]
    print(short)
    for c, v in sorted(cs.items()):
        print("    %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
