import sys, os
os.environ['GPEMU_PROF_DUMP']='1'
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
ctx = abi.Context(0)
kind,N,order=3,8192,1
d=8; X,y = synth.design(N,d,5); th = synth.default_thetas(kind,d)
ctx.set_model(kind,order,X,y)
r = ctx.loglik(th); r=ctx.loglik(th)
ctx.prof_begin(abi.PROF_GEMM); ctx.loglik_enqueue(th); p=ctx.prof_end(); print(p)
