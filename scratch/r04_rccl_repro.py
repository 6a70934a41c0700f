"""does gpemu_rccl_allgather work after the process has already used HIP through libgpemu_hip?  (it failed in the full suite)
   python scratch/r04_rccl_repro.py [first]   first = call the gather before any other device work"""
import ctypes, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth

def gather(tag):
    L = abi.load()
    n = 5
    send = (ctypes.c_double * n)(*[float(i) for i in range(n)])
    recv = (ctypes.c_double * n)()
    err = ctypes.create_string_buffer(256)
    with tempfile.TemporaryDirectory() as t:
        rc = L.gpemu_rccl_allgather(0, 0, 1, os.path.join(t, "id").encode(), send, n, recv, err, 256)
    print(tag, "rc", rc, err.value, list(recv), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "first":
    gather("before any device work:")
X, y = synth.design(256, 3, 1)
ctx = abi.Context(0)
ctx.set_model(1, 1, X, y)
print("loglik", ctx.loglik(synth.default_thetas(1, 3)), flush=True)
gather("after a likelihood:")
ctx.close()
gather("after closing the context:")
if "torch" in sys.argv:
    import torch                      # torch brings its own librccl + HIP runtime into the process
    torch.zeros(1)
    import torch.distributed          # noqa
    gather("after importing torch:")
