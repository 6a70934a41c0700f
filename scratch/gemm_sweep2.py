import sys; sys.path.insert(0,'.')
from madaiemulator_amd import abi
ctx = abi.Context(0)
for (m,n,k) in [(7680,7680,256),(7680,7680,512),(7680,7680,1024),(7680,7680,2048),(4096,4096,512),(2048,2048,512)]:
    for tri in (0,1):
        row=[]
        for cfg in (0,2,3):
            ms,fl = ctx.gemm_bench(m,n,k,ld=8192,cfg=cfg,tri=tri,beta=1,reps=5)
            row.append((ms*1e3, fl/ms/1e9))
        print(f"m={m:6d} n={n:5d} k={k:5d} tri={tri} | " + " | ".join(f"{us:9.1f}us {tf:5.1f}TF" for us,tf in row), flush=True)
