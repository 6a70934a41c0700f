import sys, time, numpy as np
sys.path.insert(0, '.')
from madaiemulator_amd import abi
from oracle import oracle as O
import scipy.linalg as sl
ctx = abi.Context(0)
rng = np.random.default_rng(0)
# 1. GEMM layout check with asymmetric data
for (m,n,k) in [(16,16,16),(128,128,16),(200,72,64),(384,256,128)]:
    A = rng.standard_normal((m,k)); B = rng.standard_normal((n,k)); C0 = rng.standard_normal((m,n))
    C = ctx.test_gemm_nt(A,B,C0,alpha=-1.0,beta=1)
    ref = C0 - A@B.T
    print("gemm",m,n,k,"maxerr",abs(C-ref).max())
# 2. potrf
for n in [64,100,128,200,512,1000]:
    M = rng.standard_normal((n,n)); S = M@M.T + n*np.eye(n)
    L,info = ctx.test_potrf(S)
    Lr = np.linalg.cholesky(S)
    print("potrf",n,"info",info,"maxerr",abs(L-Lr).max())
# non-PD
S = np.eye(100); S[57,57] = -1.0
L,info = ctx.test_potrf(S); print("nonpd info", info)
# 3. cov fill + loglik vs oracle
for kind in (1,2,3):
  for (N,d,order) in [(34,1,1),(200,2,0),(300,8,1)]:
    X = rng.random((N,d)); y = np.sin(X.sum(1)*3)+0.01*rng.standard_normal(N)
    nth = abi.nthetas_for(kind,d)
    th = np.array([0.0,-4.0]+[np.log(0.6)]*d) if kind==1 else np.array([1.0,0.01,np.log(0.6)])
    ctx.set_model(kind,order,X,y)
    Cg = ctx.cov_matrix(th); Co = O.cov_matrix(kind,X,th)
    print("cov kind",kind,N,d,"rel",abs(Cg-Co).max()/abs(Co).max())
    r = ctx.loglik(th)
    if kind==1:
        o = O.eval_fn_multi(kind,order,X,y,th[1:])
        print("  loglik",r['value'],o['value'],"rel",abs(r['value']-o['value'])/abs(o['value']),"s2",r['sigma2'],o['sigma2'],"beta",abs(r['beta']-o['beta']).max())
    e = O.Emulator(kind,order,X,y,th)
    beta,rc = ctx.predict_setup(th)
    Xq = np.vstack([rng.random((20,d)), X[:3]])
    m,v = ctx.predict(Xq); mo,vo,_ = e.emulate(Xq)
    print("  pred mean err",abs(m-mo).max(),"var err",abs(v-vo).max(), "beta err",abs(beta-e.beta).max())
    Ci = ctx.cinverse(); print("  cinv rel",abs(Ci-e.cinverse).max()/abs(e.cinverse).max())
    if kind==1:
        g,rc = ctx.grad(th); go,_ = O.grad_fn_multi(kind,order,X,y,th[1:]); print("  grad",g,go)
# 4. timing N=4096, 8192
for N in (4096,8192):
    d=8; X = rng.random((N,d)); y = np.sin(X.sum(1)*3)+0.01*rng.standard_normal(N)
    th = np.array([0.0,-4.0]+[np.log(0.6)]*d)
    ctx.set_model(1,1,X,y)
    r = ctx.loglik(th); print(N, r['value'], r['info'])
    t=time.time(); K=5
    for i in range(K): ctx.loglik_enqueue(th)
    r = ctx.loglik_collect(); dt=(time.time()-t)/K
    print("N",N,"eval ms",dt*1e3,"TF",N**3/3/dt/1e12)
    for cls,name in ((abi.PROF_GEMM,'gemm'),(abi.PROF_LEAF,'leaf'),(abi.PROF_FILL,'fill')):
        ctx.prof_begin(cls); ctx.loglik_enqueue(th); p=ctx.prof_end(); print("  ",name,p, (p['flops']/p['ms']/1e9 if p['ms'] else 0),"TF/s")
    t=time.time(); beta,rc = ctx.predict_setup(th); print(" setup s",time.time()-t)
    M=32768; Xq=rng.random((M,d)); t=time.time(); m,v=ctx.predict(Xq); dt=time.time()-t; print(" predict/s",M/dt)
    t=time.time(); m,v=ctx.predict(Xq); dt=time.time()-t; print(" predict/s (2nd)",M/dt)
    ctx.prof_begin(abi.PROF_GEMM); m,v=ctx.predict(Xq); p=ctx.prof_end(); print("   pred gemm",p,p['flops']/p['ms']/1e9,"TF/s")
