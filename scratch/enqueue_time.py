import sys, time, threading
sys.path.insert(0,'.')
import torch; torch.cuda.is_available()
from madaiemulator_amd import abi, synth
kind,N,order=3,8192,1
d=8; X,y = synth.design(N,d,5); th = synth.default_thetas(kind,d)
S=4
ctxs=[abi.Context(0) for _ in range(S)]
for c in ctxs: c.set_model(kind,order,X,y); c.loglik(th); c.loglik(th)
c=ctxs[0]
ts=[]
for i in range(6):
    t=time.perf_counter(); c.loglik_enqueue(th); ts.append(time.perf_counter()-t)
c.loglik_collect()
print("host ms per enqueue (1 ctx, queue filling):", ["%.2f"%(x*1e3) for x in ts])
# single thread round robin
K=10
t=time.perf_counter()
for i in range(K):
    for c in ctxs: c.loglik_enqueue(th)
te=time.perf_counter()-t
for c in ctxs: c.loglik_collect()
tt=time.perf_counter()-t
print("1 thread: enqueue %.2f ms/eval, total %.2f ms/eval"%(te/K/S*1e3, tt/K/S*1e3))
# one host thread per ctx
def work(c):
    for i in range(K): c.loglik_enqueue(th)
    c.loglik_collect()
t=time.perf_counter()
thr=[threading.Thread(target=work,args=(c,)) for c in ctxs]
[x.start() for x in thr]; [x.join() for x in thr]
tt=time.perf_counter()-t
print("%d threads: total %.2f ms/eval"%(S, tt/K/S*1e3))
