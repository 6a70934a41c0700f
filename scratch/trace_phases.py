"""per-launch phases of the big GEMM launches of one lock-step batch from the in-kernel stamps (GPEMU_TRACE):
workgroup lifetime, prologue (C tile + first operand chunk), main loop, epilogue (stores) in shader clocks"""
import sys, os, re
os.environ['GPEMU_TRACE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
kind, N, order, d, B = 3, 8192, 1, 8, 16
X, y = synth.design(N, d, 5)
ths = np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(B)])
c = abi.Context(0)
c.set_model(kind, order, X, y)
for _ in range(3): c.loglik_batch(ths)
os.makedirs("gpurun_out", exist_ok=True)
c.trace_dump("gpurun_out/trace_phases.txt")
tot = [0, 0, 0, 0, 0]
for line in open("gpurun_out/trace_phases.txt"):
    tag, _, times = line.rpartition("|")
    if not tag.strip().startswith("gemm"): continue
    v = [int(x) for x in times.split()]
    s, e, wsum, wn, wclk, pro, epi = v[:7]
    if wn == 0: continue
    m = re.search(r"k=(\d+)", tag)
    k = int(m.group(1)) if m else 0
    life, p, ep = wclk / wn, pro / wn, epi / wn
    if k >= 512:
        print("%-44s wall %8.1f us  sampled wgs %5d  life %8.0f clk  prologue %6.0f (%.1f%%)  epilogue %6.0f (%.1f%%)  loop %.1f%%" %
              (tag.strip()[:44], (e - s) / 1e3, wn, life, p, 100 * p / life, ep, 100 * ep / life, 100 * (life - p - ep) / life))
        tot[0] += wclk; tot[1] += pro; tot[2] += epi
print("big launches: prologue %.2f%%  epilogue %.2f%% of workgroup lifetime" % (100 * tot[1] / tot[0], 100 * tot[2] / tot[0]))
