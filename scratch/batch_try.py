import sys, time, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
kind,N,order=3,8192,1
d=8; X,y = synth.design(N,d,5); th0 = synth.default_thetas(kind,d)
c=abi.Context(0); c.set_model(kind,order,X,y)
ths=np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(8)])
print(ths[:3])
single=[c.loglik(t)['value'] for t in ths]
for B in (1,2,4,8):
    r=c.loglik_batch(ths[:B])
    ok = all(r['value'][i]==single[i] for i in range(B))
    K=max(2,16//B)
    t=time.perf_counter()
    for i in range(K): c.loglik_batch_enqueue(ths[:B])
    r2=c.loglik_batch_collect()
    dt=(time.perf_counter()-t)/K
    print("B",B,"bit-identical to single:",ok,"status",r['status'],"ms/batch %.3f ms/eval %.3f"%(dt*1e3,dt*1e3/B), flush=True)
# two contexts, batches interleaved
c2=abi.Context(0); c2.set_model(kind,order,X,y)
for B in (2,4,8):
    c2.loglik_batch(ths[:B])
    K=max(2,16//B)
    t=time.perf_counter()
    for i in range(K): c.loglik_batch_enqueue(ths[:B]); c2.loglik_batch_enqueue(ths[:B])
    c.loglik_batch_collect(); c2.loglik_batch_collect()
    dt=(time.perf_counter()-t)/K/2
    print("2 ctx x B",B,"ms/eval %.3f"%(dt*1e3/B), flush=True)
