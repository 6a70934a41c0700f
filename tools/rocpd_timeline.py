#!/usr/bin/env python3
"""Timeline view of a rocprofv3 (rocpd sqlite) kernel trace of concurrent streams: how much of the wall time each
kernel class is running, alone or overlapped.   usage: python tools/rocpd_timeline.py results.db [t0_frac t1_frac]"""
import sqlite3
import sys
import collections

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
T0, T1 = rows[0][1], max(r[2] for r in rows)
lo, hi = T0 + f0 * (T1 - T0), T0 + f1 * (T1 - T0)

def cls(n):
    for key in ("gemm_nt_kernel<128", "gemm_nt_kernel<64", "leaf_factor", "leaf_solve", "cov_fill"):
        if key in n:
            return key
    return "other"

ev = []
for n, s, e in rows:
    if e <= lo or s >= hi:
        continue
    s, e = max(s, lo), min(e, hi)
    c = cls(n)
    ev.append((s, 1, c))
    ev.append((e, -1, c))
ev.sort()
active = collections.Counter()
excl = collections.Counter()     # time with only this class running
share = collections.Counter()    # time this class runs at all
conc = collections.Counter()     # time by number of kernels in flight
busy = 0
prev = ev[0][0]
for t, d, c in ev:
    dt = t - prev
    if dt > 0:
        n = sum(active.values())
        conc[n] += dt
        if n:
            busy += dt
            live = [k for k, v in active.items() if v > 0]
            for k in live:
                share[k] += dt
            if len(live) == 1:
                excl[live[0]] += dt
    active[c] += d
    prev = t
span = hi - lo
print(f"span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms ({100*busy/span:.1f} %)")
print("kernels in flight: " + "  ".join(f"{n}:{100*v/span:.1f}%" for n, v in sorted(conc.items())))
print(f"{'class':24s} {'running_ms':>11s} {'pct_span':>9s} {'alone_ms':>10s}")
for k, v in sorted(share.items(), key=lambda x: -x[1]):
    print(f"{k:24s} {v/1e6:11.3f} {100*v/span:9.1f} {excl[k]/1e6:10.3f}")
