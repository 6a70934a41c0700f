#!/usr/bin/env python3
"""Merge the GPEMU_TRACE dumps of concurrent contexts (gpemu_trace_dump) into one timeline.
usage: python tools/trace_timeline.py trace_ctx0.txt trace_ctx1.txt ...
Prints, per context, where its wall time went (gemm K=512 / narrow gemm / leaf kernels / gaps) and, overall, how
much of the time at least one big GEMM was running."""
import re
import sys
import collections


def load(path):
    out = []
    for line in open(path):
        tag, _, times = line.rpartition("|")
        s, e, wsum, wn, wclk = (int(x) for x in times.split()[:5])
        tag = tag.strip()
        m = re.search(r"k=(\d+)", tag)
        if tag.startswith("gemm"):
            cls = "gemm_k512" if m and int(m.group(1)) >= 512 else "gemm_narrow"
            if m and int(m.group(1)) >= 1024:
                cls = "gemm_k1024"
        else:
            cls = tag.split()[0]
        out.append((s, e, cls, tag, wsum, wn, wclk))
    out.sort()
    return out


def main():
    traces = [load(p) for p in sys.argv[1:]]
    t0 = min(t[0][0] for t in traces)
    for i, tr in enumerate(traces):
        span = max(e for _, e, *_ in tr) - tr[0][0]
        by = collections.Counter()
        cnt = collections.Counter()
        gaps = 0
        prev_end = tr[0][0]
        wg = collections.Counter()
        wgn = collections.Counter()
        clk = collections.Counter()
        for s, e, cls, _, wsum, wn, wclk in tr:
            clk[cls] += wclk
            by[cls] += e - s
            cnt[cls] += 1
            wg[cls] += wsum
            wgn[cls] += wn
            if s > prev_end:
                gaps += s - prev_end
            prev_end = max(prev_end, e)
        print(f"ctx {i}: start +{(tr[0][0]-t0)/1e3:.0f} us, span {span/1e6:.3f} ms, gaps {gaps/1e6:.3f} ms; " +
              "; ".join(f"{k} {v/1e6:.3f} ms (n={cnt[k]}, avg {v/cnt[k]/1e3:.1f} us, wg life {wg[k]/max(wgn[k],1)/1e3:.1f} us, {clk[k]/max(wg[k],1):.2f} GHz)"
                        for k, v in sorted(by.items())))
    # union of big-GEMM time over all contexts inside the window where all contexts are active
    lo = max(t[0][0] for t in traces)
    hi = min(max(e for _, e, *_ in t) for t in traces)
    if hi > lo:
        ev = []
        for tr in traces:
            for s, e, cls, *_ in tr:
                if cls not in ("gemm_k512", "gemm_k1024") or e <= lo or s >= hi:
                    continue
                ev.append((max(s, lo), 1))
                ev.append((min(e, hi), -1))
        ev.sort()
        depth, prev, hist = 0, lo, collections.Counter()
        for t, d in ev:
            hist[depth] += t - prev
            prev = t
            depth += d
        hist[depth] += hi - prev
        tot = hi - lo
        print(f"window with all contexts active: {tot/1e6:.3f} ms; big GEMMs in flight: " +
              "  ".join(f"{k}:{100*v/tot:.1f}%" for k, v in sorted(hist.items())))


if __name__ == "__main__":
    main()
