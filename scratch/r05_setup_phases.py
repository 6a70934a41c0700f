"""where the start-up of an emulator goes (verdict r04 'weak' 5: alloc_multi_emulator 0.20-0.25 s against 10-22 ms of kernels):
host timestamps around every C-ABI call of the first emulator of a process, then of a second one, c3 model and one pca8 component.
usage: python scratch/r05_setup_phases.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t_import = time.perf_counter()
from madaiemulator_amd import abi, synth
abi.load()
T = time.perf_counter
print("load library %.1f ms" % ((T() - t_import) * 1e3))
for label, N, d, kind, order in (("c3 N=8192 d=8 matern52 order1", 8192, 8, 3, 1), ("pca8 component N=4096 d=16 powexp order0", 4096, 16, 1, 0)):
    X, y = synth.design(N, d, 5)
    th = synth.default_thetas(kind, d)
    for rep in range(3):
        t0 = T(); c = abi.Context(0); t1 = T()
        c.set_model(kind, order, X, y); t2 = T()
        c.predict_setup(th); t3 = T()
        c.predict(synth.queries(1, d, 3)); t4 = T()
        c.predict(synth.queries(1, d, 4)); t5 = T()
        c.predict_setup(th); t6 = T()
        c.predict_setup(th); t7 = T()
        print("%s  emulator %d of the process: ctx_create %.1f  set_model %.1f  predict_setup(1st: plain launches) %.1f  first point %.2f  second point %.2f  "
              "predict_setup(2nd: graph capture) %.1f  predict_setup(3rd: replay) %.1f ms" % (
                  label, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3, (t7 - t6) * 1e3))
        t0 = T(); c.close(); print("   ctx_destroy %.1f ms" % ((T() - t0) * 1e3))
