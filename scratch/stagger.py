"""two contexts x lock-step batches of 16 (N=8192): does a phase offset between the contexts' batch sequences pay?
the second context starts with a batch of `first` evaluations (and ends with 16 - first), total work unchanged"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
N, d, B, K = 8192, 8, 16, 24
X, y = synth.design(N, d, 5)
theta = lambda i: synth.perturbed_thetas(3, d, 7, i)
ctxs = [abi.Context(0), abi.Context(0)]
for c in ctxs:
    c.set_model(3, 1, X, y)
def run(first):
    # per-context sequences of batch sizes
    seqs = [[B] * (K // 2), ([first] if first else []) + [B] * (K // 2 - (1 if first else 0)) + ([B - first] if first else [])]
    for rep in range(2):                       # first pass warms the graphs of the odd batch sizes
        for c in ctxs: c.sync()
        t0 = time.perf_counter()
        pos = [0, 0]; n = 0; pend = [0, 0]
        while pos[0] < len(seqs[0]) or pos[1] < len(seqs[1]):
            for s in (0, 1):
                if pos[s] < len(seqs[s]):
                    nb = seqs[s][pos[s]]
                    if pend[s] == abi.RESULT_RING - 1:
                        ctxs[s].loglik_batch_collect_back(abi.RESULT_RING - 2, pend_nb[s].pop(0)); pend[s] -= 1
                    ctxs[s].loglik_batch_enqueue(np.array([theta(n + i) for i in range(nb)]))
                    pend_nb[s].append(nb); pend[s] += 1
                    n += nb; pos[s] += 1
        for s in (0, 1):
            while pend[s]:
                ctxs[s].loglik_batch_collect_back(pend[s] - 1, pend_nb[s].pop(0)); pend[s] -= 1
        for c in ctxs: c.sync()
        dt = time.perf_counter() - t0
    return n / dt, dt
pend_nb = [[], []]
for rnd in range(2):
    for first in (0, 4, 8, 12, 0, 8):
        v, dt = run(first)
        print("round", rnd, "first batch of context 2:", first or 16, " %.1f evaluations/s  (%.1f ms)" % (v, dt * 1e3), flush=True)
