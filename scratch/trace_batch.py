import sys, os, time, numpy as np
os.environ['GPEMU_TRACE']='1'
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
kind,N,order=3,8192,1
d=8; X,y = synth.design(N,d,5)
S=int(sys.argv[1]) if len(sys.argv)>1 else 2
B=int(sys.argv[2]) if len(sys.argv)>2 else 16
ths=np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(B)])
ctxs=[abi.Context(0) for _ in range(S)]
for c in ctxs: c.set_model(kind,order,X,y); c.loglik_batch(ths)
K=3
t=time.perf_counter()
for i in range(K):
    for c in ctxs: c.loglik_batch_enqueue(ths)
for c in ctxs: c.loglik_batch_collect()
print("ms/eval %.3f"%((time.perf_counter()-t)/K/S/B*1e3))
for i,c in enumerate(ctxs): c.trace_dump("gpurun_out/trace_b%d_s%d_ctx%d.txt"%(B,S,i))
