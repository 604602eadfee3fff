#!/usr/bin/env python
"""Summarise rocprofv3 --pmc output (rocpd sqlite) per kernel: mean counter values per dispatch."""
import sqlite3
import sys

db = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
agg = {}
for r in rows:
    kn = r[ix["kernel_name"]] if "kernel_name" in ix else r[ix["name"]]
    if filt and filt not in kn:
        continue
    key = (kn[:60], r[ix["counter_name"]])
    a = agg.setdefault(key, [0.0, 0])
    a[0] += r[ix["value"]]
    a[1] += 1
for (kn, cn), (s, n) in sorted(agg.items()):
    print(f"{kn:60s} {cn:28s} mean={s / n:16.1f} n={n}")
