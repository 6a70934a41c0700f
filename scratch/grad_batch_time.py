import sys, time, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
kind,order,d,N=1,1,8,8192
X,y = synth.design(N,d,5)
c=abi.Context(0); c.set_model(kind,order,X,y)
ths=np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(16)])
c.loglik_grad(ths[0]); c.loglik_grad(ths[0])    # plain launches, then the graph is recorded
t=time.perf_counter()
for i in range(4): c.loglik_grad(ths[i])
print("single value+grad %.2f ms"%((time.perf_counter()-t)/4*1e3), flush=True)
for B in (4,8,16):
    c.loglik_grad_batch(ths[:B]); c.loglik_grad_batch(ths[:B])
    t=time.perf_counter()
    for _ in range(2): r=c.loglik_grad_batch(ths[:B])
    dt=(time.perf_counter()-t)/2
    print("B",B,"value+grad %.2f ms per evaluation"%(dt/B*1e3), r['status'][:3], flush=True)
