#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max durations.
usage: python tools/rocpd_summary.py results.db [--grid-y N] > profiles/<name>.txt
--grid-y N keeps only dispatches whose grid has N rows of workgroups (the lock-step batch launches of bench.py's
timed region have grid_y = batch size; the parity gate and warm-up launches of small models do not)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
where = ""
if "--grid-y" in sys.argv:
    where = f" where grid_y = {int(sys.argv[sys.argv.index('--grid-y') + 1])}"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  f"from kernels{where} group by {namecol} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for n, c, s, a, mn, mx in rows:
    print(f"{n[:70]:70s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")

# all instantiations of the factorisation / prediction GEMM together (what bench.py's roofline.avg_launch_us averages)
g = [r for r in rows if "gemm_nt_kernel" in r[0]]
if g:
    c = sum(r[1] for r in g)
    t = sum(r[2] for r in g)
    print(f"{'gemm_nt_kernel (all tile shapes)':70s} {c:7d} {t/1e6:10.3f} {t/c/1e3:10.2f} {'':>10s} {'':>10s} {100*t/tot:6.2f}")
