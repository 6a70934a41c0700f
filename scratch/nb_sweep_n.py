"""ms per evaluation for one (N, batch) at the outer panel width of the environment (GPEMU_NB_TOP), two contexts"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi, synth
N, B, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 3
order, d = (1 if kind == 3 else 0), 8
X, y = synth.design(N, d, 5)
ths = np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(B)])
cs = [abi.Context(0) for _ in range(2)]
for c in cs: c.set_model(kind, order, X, y); c.loglik_batch(ths); c.loglik_batch(ths)
K = max(3, int(3e11 / (N ** 3 / 3 * B)))
best = 1e9
for rep in range(3):
    t = time.perf_counter()
    for i in range(K):
        for c in cs: c.loglik_batch_enqueue(ths)
    for c in cs: c.loglik_batch_collect()
    best = min(best, (time.perf_counter() - t) / K / 2 / B)
print("N %d B %d NB_TOP %s ms/eval %.4f" % (N, B, os.environ.get("GPEMU_NB_TOP", "auto"), best * 1e3), flush=True)
