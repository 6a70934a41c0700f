import sys; sys.path.insert(0,'.')
from madaiemulator_amd import abi
ctx = abi.Context(0)
shapes = [ # (m, n, k, tri) as in potrf levels of N=8192 (+64 rhs rows)
 (4160,4096,4096,1),(6208,2048,2048,1),(2112,2048,2048,1),(7232,1024,1024,1),(3136,1024,1024,1),(1088,1024,1024,1),
 (7744,512,512,1),(4160,512,512,1),(576,512,512,1),(8000,256,256,1),(4160,256,256,1),(320,256,256,1),
 (8128,128,128,1),(4160,128,128,1),(192,128,128,1),(8192,64,64,1),(4160,64,64,1),(128,64,64,1),
 (16384,8256,8192,0),
]
for (m,n,k,tri) in shapes:
    row=[]
    for cfg in (0,1,2):
        reps = 3 if k>=2048 else 10
        ms,fl = ctx.gemm_bench(m,n,k,ld=8192,cfg=cfg,tri=tri,beta=1,reps=reps)
        row.append((ms*1e3, fl/ms/1e9))
    print(f"m={m:6d} n={n:5d} k={k:5d} tri={tri} | " + " | ".join(f"{us:9.1f}us {tf:5.1f}TF" for us,tf in row), flush=True)
