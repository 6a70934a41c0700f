import os, sys, subprocess, numpy as np
sys.path.insert(0,'.')
exe="/tmp/host_api_driver"; f="/tmp/train_model.dat"
th=[-3.5]+[-2.5]*8
out=subprocess.run([exe,"eval",f,"1","1"]+[repr(float(t)) for t in th],capture_output=True,text=True,timeout=300)
print(out.stdout[-800:]); print(out.stderr[-500:])
