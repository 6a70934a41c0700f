"""in-kernel timestamps of leaf_factor_kernel (GPEMU_TRACE=1): shader clocks from the workgroup's start to the end of the first
16-column panel, to the end of the first rank-16 update, to the end -- what the 64 sequential pivots cost a single matrix.
usage: python scratch/r05_leaf_factor_phases.py [N]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPEMU_TRACE"] = "1"
os.environ["GPEMU_FACTOR_AHEAD"] = "0"       # every diagonal block through leaf_factor_kernel
import numpy as np
from madaiemulator_amd import abi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
X, y = synth.design(N, 8, 6)
c = abi.Context(0)
c.set_model(1, 0, X, y)
for j in range(3): c.loglik(synth.perturbed_thetas(1, 8, 9, j))
with tempfile.TemporaryDirectory() as t:
    p = os.path.join(t, "trace.txt"); c.trace_dump(p); lines = open(p).read().splitlines()
rows = []
for ln in lines:
    tag, _, times = ln.rpartition("|")
    if not tag.strip().startswith("leaf_factor"): continue
    q = [int(v) for v in times.split()]
    if q[3]: rows.append(q)
a = np.array(rows, float)
print("leaf_factor_kernel, %d launches with a reporting workgroup: wall %.2f us; shader clocks: whole workgroup %.0f, first panel done at %.0f, first update done at %.0f"
      % (len(rows), np.mean(a[:, 1] - a[:, 0]) / 1e3, np.mean(a[:, 4]), np.mean(a[:, 6]), np.mean(a[:, 7])))
