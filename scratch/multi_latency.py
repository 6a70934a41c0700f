import sys, time, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
N,d,nr=4096,16,8
X,y=synth.design(N,d,3); Y=synth.multi_outputs(X,y,nr)
th=synth.default_thetas(1,d); th[2:]=np.log(1.5)
cs=[]
for c in range(nr):
    k=abi.Context(0); k.set_model(1,1,X,Y[:,c]); k.predict_setup(th); cs.append(k)
q=synth.queries(1,d,5)
for k in cs: k.predict(q)
K=50
t=time.perf_counter()
for i in range(K):
    r=[k.predict(q) for k in cs]
t1=(time.perf_counter()-t)/K
t=time.perf_counter()
for i in range(K):
    for k in cs: k.predict_enqueue(q)
    r2=[k.predict_collect() for k in cs]
t2=(time.perf_counter()-t)/K
assert all(a[0][0]==b[0][0] for a,b in zip(r,r2))
print("one point, %d components: sequential %.3f ms, all enqueued then collected %.3f ms"%(nr,t1*1e3,t2*1e3))
