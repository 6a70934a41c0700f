"""golden_n16384_c5.npz: BASELINE.json configs[4]'s size (N = 16384, d = 8, pow-exp, regression order 0; the design and
thetas of tests/test_gpu_parity.py::test_config5_n16384_powexp) evaluated by the independent numpy / LAPACK chain of
tests/gradref.py -- value, sigma^2, beta, log det, quadratic form, the literal gradient (maxmultimin.c:416-550, 571-608 with
emulator.c:173-209) and the exact one, and 64 posterior means / variances (emulator.c:578-593, 672-785): one explicit
N = 16384 inverse and two more factorisations, about 4 minutes of 8 host cores -- too long for a test with a 240 s limit,
hence a fixture.  Nothing in the chain comes from the device library or from oracle/gp_oracle.c.
    python tests/golden/make_golden_n16384_c5.py"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gradref
from madaiemulator_amd import synth

N, d, order, seed = 16384, 8, 0, 20261003 + 4
X, y = synth.design(N, d, seed)
th = synth.default_thetas(1, d)
assert th[0] == 0.0                                    # value / gradient take theta0 = 0 (maxmultimin.c:311): one matrix serves both
t0 = time.time()
ref = gradref.value_and_gradients(X, y, order, th)
print("value + gradients: %.0f s" % (time.time() - t0), ref["value"], ref["literal"], ref["exact"])
Xq = np.vstack([synth.queries(62, d, 91), X[5:6], np.full((1, d), 40.0)])      # 62 random queries, a training point, a far point
t0 = time.time()
mean, var = gradref.predict(X, y, order, th, Xq)
print("64 predictions: %.0f s" % (time.time() - t0))
np.savez(os.path.join(HERE, "golden_n16384_c5.npz"), meta=np.array([1, order, N, d, seed]), thetas=th, value=ref["value"],
         sigma2=ref["sigma2"], beta=ref["beta"], logdet=ref["logdet"], quad=ref["quad"], literal=ref["literal"], exact=ref["exact"],
         Xq=Xq, mean=mean, var=var)
print("written")
