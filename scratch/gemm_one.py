import sys; sys.path.insert(0,'.')
from madaiemulator_amd import abi
ctx = abi.Context(0)
for cfg in (2,0,3):
    ms,fl = ctx.gemm_bench(7680,7680,1024,ld=8192,cfg=cfg,tri=0,beta=1,reps=3)
    print(cfg, ms, fl/ms/1e9)
