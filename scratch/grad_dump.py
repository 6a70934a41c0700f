import sys, os
os.environ['GPEMU_PROF_DUMP']='1'
sys.path.insert(0,'.')
import numpy as np, time
from madaiemulator_amd import abi, synth
kind,order,d,N=1,1,8,8192
X,y = synth.design(N,d,5)
c=abi.Context(0); c.set_model(kind,order,X,y)
th=synth.default_thetas(kind,d)
c.loglik_grad(th)
c.prof_begin(abi.PROF_GEMM); c.loglik_grad(th); p=c.prof_end(); print(p)
