"""one lock-step likelihood batch of B (argv[1], default 16) at N=8192 under GPEMU_TRACE=1: device wall time by launch class (tag and, for the GEMM,
n x k), against the time the class's matrix instructions need at 78.6 TFLOP/s.  usage: python scratch/r04_batch_time_by_launch_class.py"""
import os, re, sys, tempfile, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPEMU_TRACE"] = "1"
import numpy as np
from madaiemulator_amd import abi, synth
N, d, B = 8192, 8, (int(sys.argv[1]) if len(sys.argv) > 1 else 16)
X, y = synth.design(N, d, 6)
c = abi.Context(0)
c.set_model(3, 1, X, y)
th8 = lambda j: np.array([synth.perturbed_thetas(3, d, 9, j * B + i) for i in range(B)])
c.loglik_batch(th8(0)); c.loglik_batch(th8(1))
c.loglik_batch(th8(2))
with tempfile.TemporaryDirectory() as t:
    p = os.path.join(t, "trace.txt")
    c.trace_dump(p)
    lines = open(p).read().splitlines()
agg = collections.OrderedDict()
first, last = None, 0
for ln in lines:
    tag, _, times = ln.rpartition("|")
    q = [int(v) for v in times.split()]
    s, e = q[0], q[1]
    if q[3] == 0: continue
    first = s if first is None else min(first, s); last = max(last, e)
    m = re.search(r"gemm m=(\d+) n=(\d+) k=(\d+)", tag)
    if m:
        mm, nn, kk = (int(v) for v in m.groups())
        key = "gemm n=%d k=%d%s" % (nn, kk, " (rhs rows)" if mm == 64 else "")
        fl = 2.0 * kk * B * sum(min(nn, i + 1 + 0) if False else min(nn, i + (mm - nn if mm >= nn else 0) + 1) for i in range(mm)) if mm != 64 else 2.0 * kk * B * 64 * nn
    else:
        key = tag.strip().split()[0]; fl = 0.0
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += fl
tot = (last - first) / 1e3
print("batch wall %.1f us (first workgroup start to last end)" % tot)
print("%-34s %6s %10s %6s %12s %8s" % ("class", "n", "wall_us", "pct", "mfma_us", "frac"))
for k, (n, w, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    ideal = fl / 78.6e12 * 1e6
    print("%-34s %6d %10.1f %6.1f %12.1f %8.2f" % (k, n, w, 100 * w / tot, ideal, ideal / w if w else 0))
print("sum of launch walls %.1f us" % sum(v[1] for v in agg.values()))
