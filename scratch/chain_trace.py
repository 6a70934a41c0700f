"""in-kernel timeline of ONE evaluation (N=8192): the first 40 launches; usage: chain_trace.py FA SA"""
import sys, os, time
os.environ['GPEMU_TRACE'] = '1'
os.environ['GPEMU_NO_GRAPH'] = '1'
os.environ['GPEMU_FACTOR_AHEAD'] = sys.argv[1]
os.environ['GPEMU_SOLVE_AHEAD'] = sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi, synth
kind, N, order, d = 3, 8192, 1, 8
X, y = synth.design(N, d, 5); th = synth.default_thetas(kind, d)
c = abi.Context(0)
c.set_model(kind, order, X, y)
for i in range(3): c.loglik(th)
t = time.perf_counter(); c.loglik(th); print("ms/eval %.3f" % ((time.perf_counter() - t) * 1e3))
path = "/tmp/chain_trace.txt"
c.trace_dump(path)
rows = []
for line in open(path):
    tag, _, times = line.rpartition("|")
    q = [int(x) for x in times.split()]
    rows.append((q[0], q[1], tag.strip(), q[2] / max(q[3], 1)))
rows.sort()
t0 = rows[0][0]
for s, e, tag, wg in rows[:44]:
    print("%9.1f %9.1f  %7.1f us  (sampled wg life %6.1f us)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, wg / 1e3, tag))
span = max(e for s, e, _, _ in rows) - t0
print("launches", len(rows), "span %.3f ms, sum of kernel times %.3f ms" % (span / 1e6, sum(e - s for s, e, _, _ in rows) / 1e6))
