"""parity at the sizes where the schedule changes (leaf, outer-panel and tile-table boundaries): single and batched
likelihood, gradient and predictions against the oracle"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
from oracle import oracle as O
ctx = abi.Context(0)
worst = 0.0
t0 = time.time()
for N in (63, 64, 65, 127, 128, 129, 511, 512, 513, 1023, 1024, 1025, 1088, 1536, 2047, 2048, 2049, 3000, 4097):
    for kind, order, d in ((1, 1, 3), (3, 0, 8)):
        X, y = synth.design(N, d, 100 + N)
        ctx.set_model(kind, order, X, y)
        th = synth.default_thetas(kind, d)
        e = O.Emulator(kind, order, X, y, th)
        r = y - e.H @ e.beta
        ref = -(-0.5 * e.logdet - N / 2.0 * 1.83788 - 0.5 * (r @ e.cinverse @ r))
        errs = []
        for nb in (1, 2, 16):
            got = ctx.loglik_batch(np.array([th] * nb))
            assert np.all(got["status"] == 0)
            errs.append(float(np.max(np.abs(got["value"] - ref)) / abs(ref)))
            errs.append(float(np.max(np.abs(got["sigma2"] - y @ e.cinverse @ r / N)) / abs(got["sigma2"][0])))
        if kind == 1 and N <= 2049:
            thg = th.copy(); thg[0] = 0.0
            gg = ctx.loglik_grad(thg)
            gref, st = O.grad_fn_multi(kind, order, X, y, thg[1:])
            errs.append(float(np.max(np.abs(gg["grad"] - gref)) / np.max(np.abs(gref))))
            gb = ctx.loglik_grad_batch(np.array([thg] * 3))
            errs.append(float(np.max(np.abs(gb["grad"] - gref)) / np.max(np.abs(gref))))
        ctx.predict_setup(th)
        Q = synth.queries(33, d, 7)
        m, v = ctx.predict(Q)
        mo, vo, _ = e.emulate(Q)
        errs.append(float(np.max(np.abs(m - mo)) / max(1.0, np.max(np.abs(mo)))))
        errs.append(float(np.max(np.abs(v - vo)) / O.cov(kind, Q[0], Q[0], th)))
        worst = max(worst, max(errs))
        print("N %5d kind %d  max rel err %.2e  (%.0fs)" % (N, kind, max(errs), time.time() - t0), flush=True)
        assert max(errs) < 1e-8, errs
print("worst", worst)
