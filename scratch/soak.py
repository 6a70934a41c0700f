"""soak: random sequence of model changes / evaluations / batches / gradients / predictions on two contexts;
device memory must return to its starting level and results must stay bit-identical for repeated inputs"""
import sys, time, numpy as np
sys.path.insert(0,'.')
import torch
torch.cuda.init()
from madaiemulator_amd import abi, synth
free0,_=torch.cuda.mem_get_info()
rng=np.random.default_rng(1)
ctxs=[abi.Context(0), abi.Context(0)]
ref={}
t0=time.time()
NIT=int(sys.argv[1]) if len(sys.argv)>1 else 400
for it in range(NIT):
    c=ctxs[it%2]
    N=int(rng.choice([100,257,640,1500,3000])); d=int(rng.choice([1,3,8])); kind=int(rng.choice([1,1,3])); order=int(rng.choice([0,1]))
    X,y=synth.design(N,d,N+d)
    c.set_model(kind,order,X,y)
    th=synth.default_thetas(kind,d)
    key=(N,d,kind,order)
    v=c.loglik(th)['value']
    B=int(rng.integers(1,7))
    ths=np.array([synth.perturbed_thetas(kind,d,3,i) for i in range(B)]); ths[0]=th
    vb=c.loglik_batch(ths)['value']
    assert vb[0]==v or abs(vb[0]-v)<=1e-11*abs(v)
    if kind==1:
        g=c.loglik_grad(np.concatenate([[0.0],th[1:]]))
        gb=c.loglik_grad_batch(np.array([np.concatenate([[0.0],th[1:]])]*2))
        assert np.allclose(gb['grad'][0],g['grad'],rtol=1e-9,atol=1e-12)
    c.predict_setup(th); m,var=c.predict(synth.queries(int(rng.integers(1,300)),d,7))
    sig=(v,float(m[0]))
    # the likelihood value bit for bit; the first query's mean to rounding (the number of queries of the call picks the path:
    # one query = matrix-vector kernel, 2-16 = skinny product, more = GEMM with Gram-form k-vectors; each is deterministic)
    if key in ref: assert ref[key][0]==sig[0] and abs(ref[key][1]-sig[1])<=1e-12*max(1.0,abs(sig[1])),(key,ref[key],sig)
    ref[key]=sig
    if it%100==99: print("it",it,"elapsed %.1fs"%(time.time()-t0),"free MB",torch.cuda.mem_get_info()[0]>>20,flush=True)
for c in ctxs: c.close()
free1,_=torch.cuda.mem_get_info()
print("free before %d MB after %d MB"%(free0>>20,free1>>20))
print("delta MB", (free0-free1)>>20)
print("soak ok")
