import sys, time, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
kind,N,order=3,8192,1
d=8; X,y = synth.design(N,d,5)
ths=np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(16)])
nctx=int(sys.argv[1]) if len(sys.argv)>1 else 2
cs=[abi.Context(0) for _ in range(nctx)]
for c in cs: c.set_model(kind,order,X,y)
for B in (8,16):
    for c in cs: c.loglik_batch(ths[:B]); c.loglik_batch(ths[:B])
    K=max(2,48//B)
    t=time.perf_counter()
    for i in range(K):
        for c in cs: c.loglik_batch_enqueue(ths[:B])
    for c in cs: c.loglik_batch_collect()
    dt2=(time.perf_counter()-t)/K/nctx
    print("  %d ctx B %d ms/eval %.3f"%(nctx,B,dt2*1e3/B), flush=True)
