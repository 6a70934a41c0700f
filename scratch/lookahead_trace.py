"""in-kernel timeline of ONE evaluation (N=8192) with the look-ahead schedule: do the bulk updates (stream 2) and the
panel chain (stream 1) really overlap?  usage: lookahead_trace.py [RESERVE_CUS] [NB_TOP]"""
import sys, os, time
os.environ['GPEMU_TRACE'] = '1'
os.environ['GPEMU_LOOKAHEAD'] = '1'
os.environ['GPEMU_NO_GRAPH'] = '1'
os.environ['GPEMU_RESERVE_CUS'] = sys.argv[1] if len(sys.argv) > 1 else '32'
os.environ['GPEMU_NB_TOP'] = sys.argv[2] if len(sys.argv) > 2 else '512'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi, synth
kind, N, order, d = 3, 8192, 1, 8
X, y = synth.design(N, d, 5); th = synth.default_thetas(kind, d)
c = abi.Context(0)
c.set_model(kind, order, X, y)
for i in range(3): c.loglik(th)
t = time.perf_counter(); c.loglik(th); print("ms/eval %.3f" % ((time.perf_counter() - t) * 1e3))
path = "gpurun_out/r02_la_trace_%s_%s.txt" % (os.environ['GPEMU_RESERVE_CUS'], os.environ['GPEMU_NB_TOP'])
c.trace_dump(path)
rows = []
for line in open(path):
    tag, _, times = line.rpartition("|")
    s, e = (int(x) for x in times.split()[:2])
    rows.append((s, e, tag.strip()))
rows.sort()
t0 = rows[0][0]
# the first 60 launches with their start / end relative to the first, to see the interleaving
for s, e, tag in rows[:70]:
    print("%9.1f %9.1f  %7.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, tag))
span = max(e for s, e, _ in rows) - t0
busy = sum(e - s for s, e, _ in rows)
print("span %.3f ms, sum of kernel times %.3f ms" % (span / 1e6, busy / 1e6))
