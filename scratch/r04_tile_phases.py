"""where a 128x128 tile's lifetime goes: GPEMU_TRACE=1, one lock-step likelihood batch of 16 at N=8192; per big GEMM launch
the mean workgroup lifetime, prologue (C tile + first chunk) and epilogue (stores) in us, the shader clock, and the time the
tile's matrix instructions need on half a CU.   usage: GPEMU_STAGGER_US=.. python scratch/r04_tile_phases.py"""
import os, re, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPEMU_TRACE"] = "1"
import numpy as np
from madaiemulator_amd import abi, synth
N, d, B = 8192, 8, 16
X, y = synth.design(N, d, 6)
c = abi.Context(0)
c.set_model(3, 1, X, y)
th8 = lambda j: np.array([synth.perturbed_thetas(3, d, 9, j * B + i) for i in range(B)])
c.loglik_batch(th8(0)); c.loglik_batch(th8(1))
c.loglik_batch(th8(2))
with tempfile.TemporaryDirectory() as t:
    p = os.path.join(t, "trace.txt")
    c.trace_dump(p)
    lines = open(p).read().splitlines()
print("launch                          wall_us  wg_life_us  prologue_us  (landing_us)  epilogue_us  GHz   mfma_us(half CU)  rounds")
for ln in lines:
    tag, _, times = ln.rpartition("|")
    m = re.search(r"gemm m=(\d+) n=(\d+) k=(\d+)", tag)
    if not m: continue
    mm, nn, kk = (int(v) for v in m.groups())
    q = [int(v) for v in times.split()]
    s, e, wsum, wn, wclk, pro, epi = q[:7]
    land = q[7] if len(q) > 7 else 0
    if kk < 512 or wn == 0 or nn < 1024: continue
    ghz = wclk / max(wsum, 1)
    ideal = 2.0 * 128 * 128 * kk / (78.6e12 / 512) * 1e6
    tiles = sum(1 for i in range((mm + 127) // 128) for j in range((nn + 127) // 128) if j * 128 <= i * 128 + 127 + (mm - nn if mm > nn else 0)) * B
    print("%-30s %8.1f %10.1f %11.1f %13.1f %11.1f  %.2f %12.1f %10.2f" % (tag.strip()[:30], (e - s) / 1e3, wsum / wn / 1e3, pro / wn / ghz / 1e3, land / wn / ghz / 1e3, epi / wn / ghz / 1e3, ghz, ideal, tiles / 512.0))
