import sys, time, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
for N in (4096, 8192):
    kind,order,d=1,1,8
    X,y = synth.design(N,d,5); th = synth.default_thetas(kind,d)
    c=abi.Context(0); c.set_model(kind,order,X,y)
    c.loglik(th); c.loglik_grad(th)
    t=time.perf_counter()
    for i in range(5): c.loglik(th)
    t1=(time.perf_counter()-t)/5
    t=time.perf_counter()
    for i in range(5): r=c.loglik_grad(th)
    t2=(time.perf_counter()-t)/5
    print("N",N,"loglik %.2f ms   loglik_grad %.2f ms"%(t1*1e3,t2*1e3), flush=True)
    c.close()
