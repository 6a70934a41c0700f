import sys; sys.path.insert(0,'.')
from madaiemulator_amd import abi
ctx = abi.Context(0)
cfgs=(2,4,5,3,0)
print("cfgs",cfgs)
for (m,n,k,tri) in [(7680,7680,512,0),(7680,7680,512,1),(7680,7680,1024,1),(4096,4096,512,1),(8128,128,128,1),(8000,256,256,1),(7744,512,512,1),(4160,256,256,1)]:
    row=[]
    for cfg in cfgs:
        ms,fl = ctx.gemm_bench(m,n,k,ld=8192,cfg=cfg,tri=tri,beta=1,reps=5)
        row.append((ms*1e3, fl/ms/1e9))
    print(f"m={m:6d} n={n:5d} k={k:5d} tri={tri} | " + " | ".join(f"{us:8.1f}us {tf:5.1f}TF" for us,tf in row), flush=True)
