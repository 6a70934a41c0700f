import os, sys, subprocess, time, numpy as np
sys.path.insert(0,'.')
from madaiemulator_amd import build, synth
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N=int(sys.argv[1]) if len(sys.argv)>1 else 4096; d=int(sys.argv[2]) if len(sys.argv)>2 else 8
X,y=synth.design(N,d,777); y=y+0.1*synth.normal(5,N)
f="/tmp/train_model.dat"
open(f,"w").write(f"1\n{d}\n{N}\n"+"\n".join(" ".join(repr(float(v)) for v in r) for r in X)+"\n"+"\n".join(repr(float(v)) for v in y)+"\n")
exe="/tmp/host_api_driver"
subprocess.check_call(["gcc","-std=gnu99","-O1","-I",os.path.join(ROOT,"include"),"-I",build.HOST_SRC,"-o",exe,os.path.join(ROOT,"tests","c","host_api_driver.c"),"-L",build.LIBDIR,"-lEmuMI","-lgpemu_hip",f"-Wl,-rpath,{build.LIBDIR}","-lm"])
for name,env in (("lockstep16",dict(GPEMU_LOCKSTEP="16")),("1 thread sequential",dict(GPEMU_LOCKSTEP="1",GPEMU_NTHREADS="1")),("4 threads x own ctx",dict(GPEMU_LOCKSTEP="1",GPEMU_NTHREADS="4",GPEMU_JOBS="4",GPEMU_RESTARTS="4"))):
    e=dict(os.environ,GPEMU_SEED="99",GPEMU_RESTARTS="16"); e.update(env)
    t=time.perf_counter()
    out=subprocess.run([exe,"train",f,"1","1"],env=e,capture_output=True,text=True,timeout=1000)
    dt=time.perf_counter()-t
    last=[l for l in out.stdout.splitlines() if l.startswith("neglogl")]
    print(name,"%.1f s"%dt,last, flush=True)
