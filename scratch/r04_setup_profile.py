"""gpemu_predict_setup (= alloc_emulator_struct) at N=8192: a few calls, for rocprofv3 --kernel-trace --stats"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
N, d = 8192, 8
X, y = synth.design(N, d, 5)
c = abi.Context(0)
c.set_model(3, 1, X, y)
for i in range(3): c.predict_setup(synth.perturbed_thetas(3, d, 1, i))
t0 = time.perf_counter()
for i in range(5): c.predict_setup(synth.perturbed_thetas(3, d, 1, 10 + i))
print("predict_setup %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
