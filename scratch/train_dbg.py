"""GPEMU_OPT_DEBUG trace of a training run through the C host layer (tests/c/host_api_driver train)
   usage: train_dbg.py COV [extra env K=V ...]"""
import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madaiemulator_amd import synth, build
cov = sys.argv[1] if len(sys.argv) > 1 else "1"
N, d = 1024, 8
X, y = synth.design(N, d, 4242); y = y + 0.2 * synth.normal(17, N)
f = "/tmp/train_model2.dat"
open(f, "w").write(f"1\n{d}\n{N}\n" + "\n".join(" ".join(repr(float(v)) for v in r) for r in X) + "\n" + "\n".join(repr(float(v)) for v in y) + "\n")
exe = "/tmp/host_api_driver"
subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-I", os.path.join(ROOT, "include"), "-I", build.HOST_SRC, "-o", exe,
                       os.path.join(ROOT, "tests", "c", "host_api_driver.c"), "-L", build.LIBDIR, "-lEmuMI", "-lgpemu_hip",
                       f"-Wl,-rpath,{build.LIBDIR}", "-lm"])
e = dict(os.environ, GPEMU_SEED="99", GPEMU_JOBS="4", GPEMU_RESTARTS="1", GPEMU_LOCKSTEP="4", GPEMU_OPT_DEBUG="1", GPEMU_EXACT_GRAD="1", GPEMU_DEVICES="0")
if cov != "1": e["GPEMU_MATERN_FIXED"] = "1"
for kv in sys.argv[2:]:
    k, v = kv.split("="); e[k] = v
out = subprocess.run([exe, "train", f, cov, "0"], env=e, capture_output=True, text=True, timeout=600)
print(out.stderr[-30000:]); print(out.stdout[-1500:])
