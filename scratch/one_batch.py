"""two lock-step batches of 16 likelihood evaluations (N=8192 d=8 Matern52 order 1) -- the unit bench.py times;
used under rocprofv3 --pmc for the HBM traffic of one batched factorisation"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi, synth
ctx = abi.Context(0)
N,d,B=8192,8,16; X,y = synth.design(N,d,5)
ctx.set_model(3,1,X,y)
ths=np.array([synth.perturbed_thetas(3, d, 7, i) for i in range(B)])
for i in range(2): r = ctx.loglik_batch(ths)
print(r['value'][:3])
