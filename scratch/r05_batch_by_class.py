"""one lock-step batch under GPEMU_TRACE=1: device wall time by launch class (tag and, for the GEMM, m-class x n x k) against
the time the class's matrix instructions need at 78.6 TFLOP/s, plus the HIP-event totals per profiling class.
usage: python scratch/r05_batch_by_class.py N d B kind order inv     (inv = 1: a value+gradient batch with the inverse rows)"""
import os, re, sys, tempfile, collections, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPEMU_TRACE"] = "1"
import numpy as np
from madaiemulator_amd import abi, synth
N, d, B, kind, order, inv = (int(v) for v in sys.argv[1:7])
X, y = synth.design(N, d, 6)
c = abi.Context(0)
c.set_model(kind, order, X, y)
ths = lambda j: np.array([synth.perturbed_thetas(kind, d, 9, j * B + i) for i in range(B)])
run = (lambda t: c.loglik_grad_batch(t)) if inv else (lambda t: c.loglik_batch(t))
for j in range(3):
    run(ths(j))
with tempfile.TemporaryDirectory() as t:
    p = os.path.join(t, "trace.txt")
    c.trace_dump(p)
    lines = open(p).read().splitlines()
Np = (N + 63) // 64 * 64
agg = collections.OrderedDict()
first, last = None, 0
for ln in lines:
    tag, _, times = ln.rpartition("|")
    q = [int(v) for v in times.split()]
    s, e = q[0], q[1]
    if q[3] == 0:
        continue
    first = s if first is None else min(first, s); last = max(last, e)
    m = re.search(r"gemm m=(\d+) n=(\d+) k=(\d+)", tag)
    if m:
        mm, nn, kk = (int(v) for v in m.groups())
        cls = "rhs" if mm == 64 else ("tri" if mm >= nn else "rect")
        key = "gemm %s n=%d k=%d" % (cls, nn, kk)
        # flops of the launch: trapezoid when m >= n (triangular update: row i has min(n, i+1) columns), else full
        if mm == 64:
            fl = 2.0 * kk * B * 64 * nn
        else:
            fl = 2.0 * kk * B * (nn * (nn + 1) / 2.0 + (mm - nn) * nn)
    else:
        key = tag.strip().split()[0]; fl = 0.0
    a = agg.setdefault(key, [0, 0.0, 0.0, 0, 0, 0, 0, 0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += fl; a[3] += q[2]; a[4] += q[3]; a[5] += q[4]; a[6] += q[5]; a[7] += q[7]
tot = (last - first) / 1e3
print("N=%d d=%d B=%d kind=%d order=%d inv=%d : traced window %.1f us (first workgroup start to last end of the potrf launches)" % (N, d, B, kind, order, inv, tot))
print("%-34s %6s %10s %6s %12s %8s %8s %9s %9s %9s" % ("class", "n", "wall_us", "pct", "mfma_us", "frac", "wg_us", "wg_clk", "prolog_clk", "land_clk"))
for k, (n, w, fl, lsum, lcnt, csum, psum, wsum) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    ideal = fl / 78.6e12 * 1e6
    print("%-34s %6d %10.1f %6.1f %12.1f %8.2f %8.2f %9.0f %9.0f %9.0f" % (k, n, w, 100 * w / tot, ideal, ideal / w if w else 0, lsum / max(lcnt, 1) / 1e3,
                                                                        csum / max(lcnt, 1), psum / max(lcnt, 1), wsum / max(lcnt, 1)))
print("sum of launch walls %.1f us; N^3/3 x B at peak %.1f us -> potrf fraction %.3f" % (
    sum(v[1] for v in agg.values()), B * Np ** 3 / 3.0 / 78.6e12 * 1e6, B * Np ** 3 / 3.0 / 78.6e12 * 1e6 / tot))
t0 = time.perf_counter()
for j in range(5):
    run(ths(10 + j))
w = (time.perf_counter() - t0) / 5
print("host wall per batch %.3f ms -> %.1f evaluations/s on one context" % (w * 1e3, B / w))
