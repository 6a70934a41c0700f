import sys, time, threading
sys.path.insert(0,'.')
from madaiemulator_amd import abi
S=4
ctxs=[abi.Context(0) for _ in range(S)]
shape=dict(m=7744,n=7680,k=512,ld=8192,cfg=-1,tri=1,beta=1,reps=60)
ms,fl=ctxs[0].gemm_bench(**shape); ms,fl=ctxs[0].gemm_bench(**shape)
print("alone: %.3f ms  %.1f TF/s"%(ms, fl/ms/1e9))
for S2 in (2,4):
    out=[None]*S2
    def work(i): out[i]=ctxs[i].gemm_bench(**shape)
    t=time.perf_counter()
    thr=[threading.Thread(target=work,args=(i,)) for i in range(S2)]
    [x.start() for x in thr]; [x.join() for x in thr]
    print(S2,"concurrent: per-launch ms", ["%.3f"%o[0] for o in out], "aggregate TF/s %.1f"%(sum(o[1]/o[0]/1e9 for o in out)))
