import sys, time, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
kind,N,order,d=3,8192,1,8
X,y = synth.design(N,d,5); th = synth.default_thetas(kind,d)
c=abi.Context(0); c.set_model(kind,order,X,y); c.predict_setup(th)
for M in (1,64):
    Xq=synth.queries(M,d,3)
    c.predict(Xq)
    for cls,name in ((abi.PROF_GEMM,'gemm'),(abi.PROF_FILL,'fill')):
        c.prof_begin(cls); c.predict(Xq); p=c.prof_end(); print("M",M,name,"ms %.4f"%p['ms'],flush=True)
    dq=c.dev_alloc(Xq.nbytes); dm=c.dev_alloc(M*8); dv=c.dev_alloc(M*8); c.upload(dq,Xq)
    c.predict_dev(M,dq,dm,dv); c.sync()
    t=time.perf_counter()
    for i in range(50): c.predict_dev(M,dq,dm,dv)
    c.sync(); print("M",M,"device-resident call %.4f ms"%((time.perf_counter()-t)/50*1e3))
    t=time.perf_counter()
    for i in range(50): c.predict(Xq)
    print("M",M,"host call %.4f ms"%((time.perf_counter()-t)/50*1e3))
