"""narrow panel updates (n = K = 64 .. 512) of a batch of 16, emulated as one launch with 16 x the rows, row stride 8192
as in the workspace: TFLOP/s per tile configuration (2: 64x64 4 waves PF2, 3: 128x128 8 waves, 0: 128x128 4 waves, 1: 128x64)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi
c = abi.Context(0)
for m in (16 * 6144, 16 * 2048):
    for nk in (64, 128, 256, 512):
        row = []
        for cfg in (2, 3, 0, 1):
            ms, fl = c.gemm_bench(m=m, n=nk, k=nk, ld=8192, cfg=cfg, tri=0, beta=1, reps=5)
            ms, fl = c.gemm_bench(m=m, n=nk, k=nk, ld=8192, cfg=cfg, tri=0, beta=1, reps=20)
            row.append("cfg%d %6.1f us %5.1f TF/s" % (cfg, ms * 1e3, fl / ms / 1e9))
        print("m %6d n=k %4d  " % (m, nk) + "  ".join(row), flush=True)
