import os, sys, subprocess, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import synth
N,d=1024,8
X,y=synth.design(N,d,777); y=y+0.1*synth.normal(5,N)
f="/tmp/train_model2.dat"
open(f,"w").write(f"1\n{d}\n{N}\n"+"\n".join(" ".join(repr(float(v)) for v in r) for r in X)+"\n"+"\n".join(repr(float(v)) for v in y)+"\n")
e=dict(os.environ,GPEMU_SEED="99",GPEMU_RESTARTS="2",GPEMU_LOCKSTEP="1",GPEMU_OPT_DEBUG="1")
out=subprocess.run(["/tmp/host_api_driver","train",f,"1","1"],env=e,capture_output=True,text=True,timeout=600)
print(out.stderr[-3000:]); print(out.stdout[-300:])
