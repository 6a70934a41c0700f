"""four host threads, one context each, different models and sizes, running likelihood batches, gradients and
predictions at the same time (ctypes releases the GIL): every result must equal, bit for bit, the one the same
context produced alone beforehand"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
specs = [(1, 1, 3000, 4, 5), (3, 1, 4500, 8, 3), (1, 0, 1500, 8, 16), (2, 2, 777, 3, 7)]
work = []
for kind, order, N, d, nb in specs:
    X, y = synth.design(N, d, N)
    ths = np.array([synth.perturbed_thetas(kind, d, 5, i) for i in range(nb)])
    c = abi.Context(0)
    c.set_model(kind, order, X, y)
    ref_b = c.loglik_batch(ths)
    ref_g = c.loglik_grad(np.concatenate([[0.0], ths[0][1:]])) if kind == 1 else None
    c.predict_setup(ths[0])
    Q = synth.queries(257, d, 3)
    ref_p = c.predict(Q)
    work.append((c, kind, ths, Q, ref_b, ref_g, ref_p))
errors, counts = [], [0] * len(work)
stop = time.time() + SECONDS
def run(i):
    c, kind, ths, Q, ref_b, ref_g, ref_p = work[i]
    while time.time() < stop and not errors:
        b = c.loglik_batch(ths)
        if not (np.array_equal(b["value"], ref_b["value"]) and np.array_equal(b["beta"], ref_b["beta"])):
            errors.append(("batch", i)); return
        if ref_g is not None:
            g = c.loglik_grad(np.concatenate([[0.0], ths[0][1:]]))
            if not np.array_equal(g["grad"], ref_g["grad"]):
                errors.append(("grad", i)); return
        c.predict_setup(ths[0])
        m, v = c.predict(Q)
        if not (np.array_equal(m, ref_p[0]) and np.array_equal(v, ref_p[1])):
            errors.append(("predict", i)); return
        counts[i] += 1
ts = [threading.Thread(target=run, args=(i,)) for i in range(len(work))]
for t in ts: t.start()
while any(t.is_alive() for t in ts):
    time.sleep(20)
    print("rounds so far", counts, flush=True)
for t in ts: t.join()
print("rounds", counts, "errors", errors)
assert not errors
