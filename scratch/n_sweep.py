"""likelihood evaluations per second over N (Matern 5/2, d=8, order 1): two contexts, lock-step batches"""
import sys, time, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
kind,order,d=3,1,8
for N in (64,128,256,512,1024,2048,4096,8192,12288,16384):
    X,y=synth.design(N,d,5)
    B=int(min(64,max(16,16*(8192/N)**2)))
    ths=np.array([synth.perturbed_thetas(kind,d,7,i) for i in range(B)])
    cs=[abi.Context(0),abi.Context(0)]
    for c in cs: c.set_model(kind,order,X,y); c.loglik_batch(ths); c.loglik_batch(ths)   # plain launches, then the graph
    K=max(2,int(2e12/(B*N**3/3*2)))
    K=min(K,200)
    t=time.perf_counter()
    for i in range(K):
        for c in cs: c.loglik_batch_enqueue(ths)
    for c in cs: c.loglik_batch_collect()
    dt=(time.perf_counter()-t)/(K*2*B)
    c=cs[0]
    t1=time.perf_counter()
    c.loglik(ths[0]); c.loglik(ths[0])
    t1=time.perf_counter()
    for i in range(5): c.loglik(ths[0])
    single=(time.perf_counter()-t1)/5
    print("N %5d  batch %2d  %9.1f evals/s  (%.3f ms each, %.1f TFLOP/s of N^3/3)   one at a time %.3f ms"%(N,B,1/dt,dt*1e3,N**3/3/dt/1e12,single*1e3),flush=True)
    for c in cs: c.close()
