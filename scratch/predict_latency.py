import sys, time, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
kind,N,order,d=3,8192,1,8
X,y = synth.design(N,d,5); th = synth.default_thetas(kind,d)
c=abi.Context(0); c.set_model(kind,order,X,y); c.predict_setup(th)
for M in (1,4,16,17,32,33,64,256,1024,4096,50000):
    Xq=synth.queries(M,d,3)
    c.predict(Xq)
    K=20 if M<5000 else 3
    t=time.perf_counter()
    for i in range(K): c.predict(Xq)
    dt=(time.perf_counter()-t)/K
    c.prof_begin(abi.PROF_GEMM); c.predict(Xq); p=c.prof_end()
    print("M",M,"%.3f ms per call  %.1f us per point   (product kernel %.1f us)"%(dt*1e3, dt/M*1e6, p['ms']*1e3), flush=True)
