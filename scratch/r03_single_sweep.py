"""one evaluation at a time (N=8192, Matern 5/2, order 1) under the remaining schedule switches, one context per variant in
ONE process (the switches are per context since round 3): outer panel width, big-tile threshold, factor-ahead, graph"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi, synth
N, d = 8192, 8
X, y = synth.design(N, d, 20261005)
variants = [{}, {"GPEMU_NB_TOP": "256"}, {"GPEMU_NB_TOP": "384"}, {"GPEMU_NB_TOP": "768"}, {"GPEMU_NB_TOP": "1024"},
            {"GPEMU_GEMM_BIG_TILES": "512"}, {"GPEMU_GEMM_BIG_TILES": "768"}, {"GPEMU_GEMM_BIG_TILES": "1000000"},
            {"GPEMU_NB_TOP": "1024", "GPEMU_GEMM_BIG_TILES": "512"}, {"GPEMU_FACTOR_AHEAD": "0"}, {"GPEMU_NO_GRAPH": "1"}]
ctxs = []
for env in variants:
    for k, v in env.items(): os.environ[k] = v
    c = abi.Context(0)
    for k in env: del os.environ[k]
    c.set_model(3, 1, X, y)
    for i in range(3): c.loglik(synth.perturbed_thetas(3, d, 1, i))
    ctxs.append(c)
for rnd in range(2):
    for env, c in zip(variants, ctxs):
        t0 = time.perf_counter()
        for i in range(10): r = c.loglik(synth.perturbed_thetas(3, d, 2 + rnd, i))
        print("round", rnd, env or "default", "%.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
