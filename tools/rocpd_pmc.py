#!/usr/bin/env python3
"""Per-kernel sums of the counters in a rocprofv3 --pmc (rocpd sqlite) result.
usage: python tools/rocpd_pmc.py results.db [more.db ...]   (one db per --pmc pass)"""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    rows = db.execute("select counter_name, kernel_name, count(*), sum(value), sum(duration) from counters_collection "
                      "group by counter_name, kernel_name order by counter_name, 4 desc").fetchall()
    cur = None
    for cname, kname, n, total, dur in rows:
        if cname != cur:
            print(cname)
            cur = cname
        print(f"  {kname[:60]:60s} calls {n:5d}  sum {total:16.0f}  per_launch {total/n:14.1f}  dur_ms {dur/1e6:9.3f}")
