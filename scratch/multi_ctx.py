import sys, time, numpy as np
sys.path.insert(0,'.')
import torch; torch.cuda.is_available()
from madaiemulator_amd import abi, synth
N,d=8192,8; X,y = synth.design(N,d,5)
for nctx in (1,2,3,4):
    ctxs=[abi.Context(0) for _ in range(nctx)]
    for c in ctxs: c.set_model(3,1,X,y); c.loglik(synth.default_thetas(3,d))
    K=24
    torch.cuda.synchronize(); t=time.time()
    for i in range(K): ctxs[i%nctx].loglik_enqueue(synth.perturbed_thetas(3,d,1,i))
    vals=[c.loglik_collect()['value'] for c in ctxs]
    torch.cuda.synchronize(); dt=time.time()-t
    print("nctx",nctx,"evals/s %.1f"%(K/dt),"ms/eval %.2f"%(dt/K*1e3), flush=True)
    for c in ctxs: c.close()
