import sys, time, os, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
ctx = abi.Context(0)
for (kind,N,order) in [(3,8192,1),(1,4096,0)]:
    d=8; X,y = synth.design(N,d,5); th = synth.default_thetas(kind,d)
    ctx.set_model(kind,order,X,y)
    r = ctx.loglik(th); print(N, r['value'], r['info'])
    t=time.time(); K=10
    for i in range(K): ctx.loglik_enqueue(th)
    r = ctx.loglik_collect(); dt=(time.time()-t)/K
    print("N",N,"eval ms %.3f"%(dt*1e3),"TF %.1f"%(N**3/3/dt/1e12), flush=True)
    for cls,name in ((abi.PROF_GEMM,'gemm'),(abi.PROF_LEAF,'leaf'),(abi.PROF_FILL,'fill')):
        ctx.prof_begin(cls); ctx.loglik_enqueue(th); p=ctx.prof_end(); print("  ",name,"n",p['n'],"ms %.3f"%p['ms'], "TF/s %.1f"%(p['flops']/p['ms']/1e9 if p['ms'] else 0))
