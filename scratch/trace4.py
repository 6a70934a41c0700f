import sys, os, time
os.environ['GPEMU_TRACE']='1'
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
kind,N,order=3,8192,1
d=8; X,y = synth.design(N,d,5); th = synth.default_thetas(kind,d)
S=int(sys.argv[1]) if len(sys.argv)>1 else 4
ctxs=[abi.Context(0) for _ in range(S)]
for c in ctxs: c.set_model(kind,order,X,y); c.loglik(th); c.loglik(th)
K=8
t=time.perf_counter()
for i in range(K):
    for c in ctxs: c.loglik_enqueue(th)
for c in ctxs: c.loglik_collect()
print("ms/eval %.3f"%((time.perf_counter()-t)/K/S*1e3))
for i,c in enumerate(ctxs): c.trace_dump("gpurun_out/trace_s%d_ctx%d.txt"%(S,i))
