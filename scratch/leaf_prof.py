import sys, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
kind,N,order=3,8192,1
d=8; X,y = synth.design(N,d,5)
c=abi.Context(0); c.set_model(kind,order,X,y)
for B in (1,2,4,16):
    ths=np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(B)])
    c.loglik_batch(ths)
    for cls,name in ((abi.PROF_LEAF,'leaf(factor+solve)'),(abi.PROF_GEMM,'gemm'),(abi.PROF_FILL,'fill')):
        c.prof_begin(cls); c.loglik_batch_enqueue(ths); p=c.prof_end(); c.loglik_batch_collect()
        print("B",B,name,"n",p['n'],"total ms %.3f avg us %.1f"%(p['ms'],p['ms']*1e3/p['n']),flush=True)
