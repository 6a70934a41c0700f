import sys, time, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
kind,N,order=3,8192,1
d=8; X,y = synth.design(N,d,5)
ths=np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(16)])
c=abi.Context(0); c.set_model(kind,order,X,y)
c2=abi.Context(0); c2.set_model(kind,order,X,y)
for B in (4,8,16):
    c.loglik_batch(ths[:B]); c2.loglik_batch(ths[:B])
    K=max(2,32//B)
    t=time.perf_counter()
    for i in range(K): c.loglik_batch_enqueue(ths[:B])
    c.loglik_batch_collect()
    dt1=(time.perf_counter()-t)/K
    t=time.perf_counter()
    for i in range(K): c.loglik_batch_enqueue(ths[:B]); c2.loglik_batch_enqueue(ths[:B])
    c.loglik_batch_collect(); c2.loglik_batch_collect()
    dt2=(time.perf_counter()-t)/K/2
    print("B",B,"1 ctx ms/eval %.3f   2 ctx ms/eval %.3f"%(dt1*1e3/B, dt2*1e3/B), flush=True)
