import sys, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import abi, synth
ctx = abi.Context(0)
N,d=8192,8; X,y = synth.design(N,d,5); th = synth.default_thetas(3,d)
ctx.set_model(3,1,X,y)
for i in range(3): r = ctx.loglik(th)
print(r['value'])
